"""Seeded synthetic read generator for the tests (numpy RandomState is stable across numpy versions).
Same spirit as the reference's util/gen_fastq_noRC (uniform read starts on an i.i.d. genome; -e: 1 % substitutions of which
a quarter become N, gen_fastq_noRC.cpp:67-71,119-130) plus reverse-complemented odd reads as util/gen_fastq."""
import numpy as np

_COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    _COMP[a] = b


def reads_array(seed, n, L, genome_len, err=0.0, rc=True, n_frac=0.25):
    """-> uint8 array [n, L] of ASCII bases"""
    rs = np.random.RandomState(seed)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=genome_len)]
    starts = rs.randint(0, genome_len - L, size=n)
    idx = starts[:, None] + np.arange(L)[None, :]
    r = genome[idx].copy()
    if err > 0:
        e = rs.random_sample((n, L)) < err
        isN = e & (rs.random_sample((n, L)) < n_frac)
        sub = e & ~isN
        code = np.searchsorted(np.frombuffer(b"ACGT", dtype=np.uint8), r)
        newcode = (code + rs.randint(1, 4, size=(n, L))) % 4
        r[sub] = np.frombuffer(b"ACGT", dtype=np.uint8)[newcode[sub]]
        r[isN] = ord("N")
    if rc:
        odd = np.arange(n) % 2 == 1
        r[odd] = _COMP[r[odd][:, ::-1]]
    return r


def reads_array_big(seed, n, L, genome_len, err=0.0, n_frac=0.25, chunk=250_000):
    """reads_array's distribution for CONFIG-size inputs (millions of reads), made chunk by chunk so that no (n, L) array of doubles
    ever exists; its own stream of random numbers (not reads_array's).  Odd reads reverse-complemented."""
    rs = np.random.RandomState(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rs.randint(0, 4, size=genome_len, dtype=np.uint8)]
    out = np.empty((n, L), dtype=np.uint8)
    ar = np.arange(L)
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        starts = rs.randint(0, genome_len - L, size=m)
        r = genome[starts[:, None] + ar[None, :]]
        if err > 0:
            e = rs.random_sample((m, L)) < err
            ne = int(e.sum())
            isn = rs.random_sample(ne) < n_frac
            code = np.searchsorted(acgt, r[e])
            newb = acgt[(code + rs.randint(1, 4, size=ne)) % 4]
            newb[isn] = ord("N")
            r[e] = newb
        odd = (np.arange(s, s + m) % 2) == 1
        r[odd] = _COMP[r[odd][:, ::-1]]
        out[s:s + m] = r
    return out


def lines_of(r):
    """[n, L] uint8 -> bytes, one read per line"""
    n, L = r.shape
    out = np.empty((n, L + 1), dtype=np.uint8)
    out[:, :L] = r
    out[:, L] = 10
    return out.tobytes()


def reads_text(*a, **kw):
    """-> bytes: one read per line"""
    r = reads_array(*a, **kw)
    n, L = r.shape
    out = np.empty((n, L + 1), dtype=np.uint8)
    out[:, :L] = r
    out[:, L] = 10
    return out.tobytes()


def reads_text_lowcomplexity(seed, n, L, genome_len, n_repeat=60, n_polya=12, err=0.0):
    """genome with `n_repeat` copies of one 300-bp element and `n_polya` poly-A runs of 150 bp: dictionary bins with
    hundreds to thousands of reads (the maxsearch window and the wave-cooperative bin scan of k_steps)"""
    rs = np.random.RandomState(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rs.randint(0, 4, size=genome_len)].copy()
    rep = acgt[rs.randint(0, 4, size=300)]
    for p in rs.randint(0, genome_len - 400, size=n_repeat):
        genome[p:p + 300] = rep
    for p in rs.randint(0, genome_len - 400, size=n_polya):
        genome[p:p + 150] = ord("A")
    starts = rs.randint(0, genome_len - L, size=n)
    r = genome[starts[:, None] + np.arange(L)[None, :]].copy()
    if err > 0:
        e = rs.random_sample((n, L)) < err
        r[e] = acgt[rs.randint(0, 4, size=int(e.sum()))]
    odd = np.arange(n) % 2 == 1
    r[odd] = _COMP[r[odd][:, ::-1]]
    out = np.empty((n, L + 1), dtype=np.uint8)
    out[:, :L] = r
    out[:, L] = 10
    return out.tobytes()


def reads_text_bigbin_stage2(seed, n_clean=3000, n_dupN=2500, L=100, genome_len=5000):
    """stage-II bins above maxsearch: n_dupN reads with N that all share their first 50 bases (copies of one genome window
    with an N and a substitution in the second half), next to n_clean ordinary reads that build the contig they realign to"""
    rs = np.random.RandomState(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rs.randint(0, 4, size=genome_len)]
    starts = rs.randint(0, genome_len - L, size=n_clean)
    clean = genome[starts[:, None] + np.arange(L)[None, :]].copy()
    p = genome_len // 2
    dup = np.tile(genome[p:p + L], (n_dupN, 1)).copy()
    dup[np.arange(n_dupN), rs.randint(50, L, size=n_dupN)] = ord("N")
    sub = rs.randint(50, L, size=n_dupN)
    dup[np.arange(n_dupN), sub] = np.where(dup[np.arange(n_dupN), sub] == ord("N"), ord("N"), acgt[rs.randint(0, 4, size=n_dupN)])
    allr = np.concatenate([clean, dup])
    rs.shuffle(allr)
    out = np.empty((allr.shape[0], L + 1), dtype=np.uint8)
    out[:, :L] = allr
    out[:, L] = 10
    return out.tobytes()


def reads_text_bigbin_stage2_mixed(seed, n_clean=4000, n_dupN=3000, L=100, genome_len=6000, copies=3, fail_frac=0.6, nsub=32):
    """as reads_text_bigbin_stage2, but the shared 100-base window occurs `copies` times in the genome (several probes per bin, in
    tuple order) and a random fail_frac of the N reads carry nsub substitutions in their second half: they sit in the same two bins
    but fail the Hamming test, so they stay unclaimed inside every probe's maxsearch window and push it on unevenly"""
    rs = np.random.RandomState(seed)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = acgt[rs.randint(0, 4, size=genome_len)]
    win = genome[200:200 + L].copy()
    for k in range(1, copies):
        at = 200 + k * (genome_len - 400) // copies
        genome[at:at + L] = win
    starts = rs.randint(0, genome_len - L, size=n_clean)
    clean = genome[starts[:, None] + np.arange(L)[None, :]].copy()
    dup = np.tile(win, (n_dupN, 1)).copy()
    bad = rs.rand(n_dupN) < fail_frac
    for i in np.nonzero(bad)[0]:
        cols = rs.choice(np.arange(50, L), size=nsub, replace=False)
        dup[i, cols] = acgt[(np.searchsorted(acgt, dup[i, cols]) + 1 + rs.randint(0, 3, size=nsub)) % 4]
    dup[np.arange(n_dupN), rs.randint(50, L, size=n_dupN)] = ord("N")
    allr = np.concatenate([clean, dup])
    rs.shuffle(allr)
    out = np.empty((allr.shape[0], L + 1), dtype=np.uint8)
    out[:, :L] = allr
    out[:, L] = 10
    return out.tobytes()


def reads_text_hugebin_stage1(seed, n_core=9000, n_norm=6000, L=100, genome_len=40000):
    """stage-I bins far above maxsearch AND above the 4096 entries k_compact_huge looks at per pass: n_core reads that share bases [15, 55) (the first
    dictionary's whole window) and are random elsewhere -- one bin of n_core reads none of which overlaps another within the Hamming threshold, thinned
    from the top as the chains take their seeds from the descending cursor -- shuffled among ordinary reads of a small genome"""
    rs = np.random.RandomState(seed)
    core = rs.randint(0, 4, 40)
    a = rs.randint(0, 4, (n_core, L))
    a[:, 15:55] = core
    g = rs.randint(0, 4, genome_len)
    st = rs.randint(0, genome_len - L, n_norm)
    b = g[st[:, None] + np.arange(L)[None, :]]
    allr = np.concatenate([a, b])
    allr = allr[rs.permutation(allr.shape[0])]
    return lines_of(np.frombuffer(b"ACGT", dtype=np.uint8)[allr])


def auto_chains(n_clean, reads_per_chain=2048, clean=None):
    """auto_chains() of harc_amd/csrc/stage1.hip: K when harc_amd_params.num_chains = 0.  clean: the clean reads ([n, L] uint8 array or
    the lines of input_clean.dna) for the low-coverage rule of stage1_run_w -- more than 98 % distinct first-dictionary k-mers: up to
    4096 chains of at least 256 reads"""
    k = n_clean // reads_per_chain
    k = max(k, min(2048, n_clean // 1024))
    k = max(1, min(k, 65536))
    if clean is not None and n_clean > 0:
        if isinstance(clean, (bytes, bytearray)):
            rows = [l for l in bytes(clean).split(b"\n") if l and b"N" not in l]
            clean = np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), -1)
        L = clean.shape[1]
        ds = L // 2 - 32 if L > 100 else L // 2 - L * 32 // 100      # harc:57-58, dict1_start .. dict1_end
        de = L // 2 - 1
        nbins = np.unique(np.ascontiguousarray(clean[:, ds:de + 1]), axis=0).shape[0]
        if nbins > 0.98 * n_clean:
            k = max(k, min(4096, n_clean // 256))
    return k


def _clean_array(clean):
    if isinstance(clean, (bytes, bytearray)):
        rows = [l for l in bytes(clean).split(b"\n") if l and b"N" not in l]
        return np.frombuffer(b"".join(rows), dtype=np.uint8).reshape(len(rows), -1)
    return clean


def auto_steps(clean, K):
    """steps per super-round when harc_amd_params.num_steps = 0 (stage1_run_w): 64 for one chain; 32 with more than 16 384 chains, or from 2048 chains on an input
    that is not a low-coverage one (at most 98 % distinct first-dictionary k-mers) -- in both cases only where the bins of more than 16 reads of the two
    dictionaries hold at most 2 % of N entries; 16 otherwise.  clean: the clean reads ([n, L] uint8 array or the lines of input_clean.dna)"""
    if K == 1:
        return 64
    a = _clean_array(clean)
    n, L = a.shape
    if n == 0:
        return 16
    w = 32 if L >= 100 else L * 32 // 100
    d1 = (L // 2 - w, L // 2 - 1)                                  # harc:57-60
    d2 = (L // 2, L // 2 + w - 1)
    large = 0
    nb1 = 0
    for k, (ds, de) in enumerate((d1, d2)):
        _, cnt = np.unique(np.ascontiguousarray(a[:, ds:de + 1]), axis=0, return_counts=True)
        if k == 0:
            nb1 = cnt.shape[0]
        large += int(cnt[cnt > 16].sum())
    lowcov = nb1 > 0.98 * n
    return 32 if (K > 16384 or (K >= 2048 and not lowcov)) and large * 50 <= n else 16
