#!/usr/bin/env python3
"""Generate golden vectors from the REAL reference (oracle/_ref, built by oracle/build_ref.sh from
/root/reference) at num_thr=1 -- the only setting at which the reference is deterministic
(reorder.cpp:545-552 races otherwise).  Runs only in the build container; the GPU box sees only the
committed fixtures tests/golden/<case>.tar.xz, each holding

    reads.txt              the FASTQ sequence lines fed to preprocess.out (inputs)
    stage1/<file>          output/ after reorder.out   (reorder.cpp:722-830 file family)
    stage2/<file>          output/ after encoder.out   (encoder.cpp:457-505, packbits :512-616)
    packed/<file>          read_order.bin(+.tail) after pack_order.out (pack_order.cpp:20-77)
    decoded.txt            output.dna of the reference decoder.out run on stage2 (decoder.cpp:65-172)
    meta.json              L, counts, stdout counters of the reference run

This file is test infrastructure (oracle side); nothing in the product imports it.
"""
import json, os, shutil, subprocess, sys, tarfile, tempfile, io
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.path.join(HERE, "_ref")
GOLD = os.path.join(os.path.dirname(HERE), "tests", "golden")

COMP = {"A": "T", "C": "G", "G": "C", "T": "A", "N": "N"}


def revcomp(s):
    return "".join(COMP[c] for c in reversed(s))


def gen_reads(seed, n, L, genome_len, err=0.0, rc=True, repeat=0, dup=0, n_frac=0.25):
    """i.i.d. genome (optionally with one `repeat`-bp segment copied elsewhere), uniform read starts,
    per-base substitution probability `err` of which n_frac become 'N' (gen_fastq_noRC.cpp:67-71 spirit),
    odd reads reverse-complemented, `dup` extra identical copies of read 0."""
    rs = np.random.RandomState(seed)
    g = rs.randint(0, 4, size=genome_len)
    if repeat:
        g[genome_len // 2: genome_len // 2 + repeat] = g[100:100 + repeat]
    genome = "".join("ACGT"[x] for x in g)
    reads = []
    for i in range(n):
        p = rs.randint(0, genome_len - L)
        r = list(genome[p:p + L])
        if err > 0:
            for j in range(L):
                if rs.random_sample() < err:
                    if rs.random_sample() < n_frac:
                        r[j] = "N"
                    else:
                        r[j] = "ACGT"[(("ACGT".index(r[j]) if r[j] != "N" else 0) + rs.randint(1, 4)) % 4]
        r = "".join(r)
        if rc and (i & 1):
            r = revcomp(r)
        reads.append(r)
    for _ in range(dup):
        reads.insert(rs.randint(0, len(reads)), reads[0])
    return reads


def gen_repfam(seed, n, L, genome_len, ncopy, div, npolya, nstr, err):
    """a genome with a diverged repeat family (ncopy copies of one 300-bp element, every copy with `div` of its bases substituted),
    poly-A runs and (CA)n runs of 150 bp: dictionary bins of tens of reads, most of which fail the Hamming test"""
    rs = np.random.RandomState(seed)
    g = rs.randint(0, 4, size=genome_len)
    rep = rs.randint(0, 4, size=300)
    slots = rs.permutation(genome_len // 400)[:ncopy] * 400
    for p in slots:
        cp = rep.copy()
        mut = rs.random_sample(300) < div
        cp[mut] = (cp[mut] + rs.randint(1, 4, size=int(mut.sum()))) % 4
        g[p:p + 300] = cp
    for p in rs.randint(0, genome_len - 400, size=npolya):
        g[p:p + 150] = 0
    for p in rs.randint(0, genome_len - 400, size=nstr):
        g[p:p + 150] = np.tile([1, 0], 75)
    genome = "".join("ACGT"[x] for x in g)
    reads = []
    for i in range(n):
        p = rs.randint(0, genome_len - L)
        r = list(genome[p:p + L])
        for j in range(L):
            if rs.random_sample() < err:
                r[j] = "N" if rs.random_sample() < 0.25 else "ACGT"[("ACGT".index(r[j]) + rs.randint(1, 4)) % 4]
        r = "".join(r)
        reads.append(revcomp(r) if (i & 1) else r)
    return reads


def gen_noRC(seed, n, L, genome_len, width, errors=True):
    """the reference's own simulator (util/gen_fastq_noRC, built into oracle/_ref by build_ref.sh) on a small FASTA written here:
    std::mt19937 with its default seed, read = ref[pos, pos+L) for a 32-bit pos below ref_len - L, -e: one base in a hundred replaced,
    a quarter of those by N (gen_fastq_noRC.cpp:90-131).  Its FASTA reader takes the last line twice when the file ends with a newline
    (the feof loop of :19-32): the genome the reads come from is 'width' characters longer than the file says -- kept, it is what a user
    of the tool gets."""
    rs = np.random.RandomState(seed)
    genome = "".join("ACGT"[x] for x in rs.randint(0, 4, size=genome_len))
    wd = tempfile.mkdtemp(prefix="harc_gen_")
    try:
        fa, fq = os.path.join(wd, "g.fa"), os.path.join(wd, "out.fastq")
        with open(fa, "w") as f:
            f.write(">chrT synthetic\n")
            for i in range(0, genome_len, width):
                f.write(genome[i:i + width] + "\n")
        run([os.path.join(REF, "gen_fastq_noRC"), str(n), str(L), fa, fq] + (["-e"] if errors else []), wd)
        lines = open(fq).read().split("\n")
        reads = [lines[4 * i + 1] for i in range(n)]
        assert all(len(r) == L for r in reads) and lines[0] == "@T.0" and lines[3] == "H" * L
        return reads
    finally:
        shutil.rmtree(wd)


CASES = {
    # name: dict(kwargs for gen_reads)
    "L100_clean_5k": dict(seed=1, n=5000, L=100, genome_len=25000),
    "L100_err_5k": dict(seed=2, n=5000, L=100, genome_len=25000, err=0.01),
    "L100_lowcov_4k": dict(seed=3, n=4000, L=100, genome_len=130000, err=0.005),
    "L150_err_3k": dict(seed=4, n=3000, L=150, genome_len=20000, err=0.01),
    "L101_err_3k": dict(seed=5, n=3000, L=101, genome_len=15000, err=0.01),
    "L63_err_3k": dict(seed=6, n=3000, L=63, genome_len=10000, err=0.01),
    "L40_err_3k": dict(seed=7, n=3000, L=40, genome_len=6000, err=0.01),
    "L255_err_1k": dict(seed=8, n=1000, L=255, genome_len=12000, err=0.01),
    "L100_repeat_dup_4k": dict(seed=9, n=2500, L=100, genome_len=12000, err=0.005, repeat=2000, dup=1500),
    "L100_one": dict(seed=10, n=1, L=100, genome_len=1000),
    "L100_allN_20": dict(seed=11, n=20, L=100, genome_len=1000, err=0.2, n_frac=1.0),
    "L100_three": dict(seed=12, n=3, L=100, genome_len=120),
    # stage-II dictionary bins above maxsearch (2500 N reads sharing their first 50 bases): the sliding window of encoder.cpp:293
    "L100_bigbin2_5k": dict(custom="bigbin_stage2", seed=77, L=100),
    # the reference's own generator (BASELINE.json configs[0] is "util/gen_fastq_noRC on chrom22"), with -e
    "L100_gen_noRC_e_3k": dict(custom="gen_noRC", seed=31, n=3000, L=100, genome_len=20030, width=70),
    # repeat families: diverged copies of a 300-bp element, poly-A and (CA)n runs -> stage-I bins of more than 16 reads (the bin-ordered
    # mirror, the COOP kernel and the scan budget of the GPU build all sit on this path)
    "L100_repfam_5k": dict(custom="repfam", seed=41, n=5000, L=100, genome_len=40000, ncopy=40, div=0.10, npolya=8, nstr=4, err=0.004),
    "L101_repfam_4k": dict(custom="repfam", seed=42, n=4000, L=101, genome_len=30000, ncopy=30, div=0.13, npolya=6, nstr=4, err=0.006),
}


def run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if r.returncode not in (0,):
        raise RuntimeError(f"{cmd} failed rc={r.returncode}:\n{r.stdout}")
    return r.stdout


def snapshot(outdir):
    d = {}
    for f in sorted(os.listdir(outdir)):
        p = os.path.join(outdir, f)
        if os.path.isfile(p):
            with open(p, "rb") as fh:
                d[f] = fh.read()
    return d


def make_case(name, kw):
    L = kw["L"]
    subprocess.check_call([os.path.join(HERE, "build_ref.sh")])
    subprocess.check_call([os.path.join(HERE, "build_ref.sh"), str(L), "1"])
    if kw.get("custom") == "bigbin_stage2":
        sys.path.insert(0, os.path.dirname(HERE))
        from tests import gen as tgen
        reads = tgen.reads_text_bigbin_stage2(kw["seed"]).decode().split()
    elif kw.get("custom") == "gen_noRC":
        reads = gen_noRC(kw["seed"], kw["n"], L, kw["genome_len"], kw["width"])
    elif kw.get("custom") == "repfam":
        reads = gen_repfam(**{k: v for k, v in kw.items() if k != "custom"})
    else:
        reads = gen_reads(**kw)
    wd = tempfile.mkdtemp(prefix="harc_gold_")
    try:
        os.makedirs(os.path.join(wd, "output"))
        fq = os.path.join(wd, "in.fastq")
        with open(fq, "w") as f:
            for i, r in enumerate(reads):
                f.write(f"@T.{i}\n{r}\n+\n{'H' * L}\n")
        log = run([os.path.join(REF, "preprocess.out"), fq, wd, "False", "False", str(L)], wd)
        log += run([os.path.join(REF, f"reorder_L{L}_t1.out"), wd], wd)   # cwd=wd: BBHash temp files land in CWD
        stage1 = snapshot(os.path.join(wd, "output"))
        log += run([os.path.join(REF, f"encoder_L{L}_t1.out"), wd], wd)
        stage2 = snapshot(os.path.join(wd, "output"))
        # -p path: pack_order on a copy of read_order.bin (harc:112)
        pd = tempfile.mkdtemp(prefix="harc_gold_p_")
        os.makedirs(os.path.join(pd, "output"))
        shutil.copy(os.path.join(wd, "output", "read_order.bin"), os.path.join(pd, "output", "read_order.bin"))
        packed = {}
        if len(stage2["read_order.bin"]) > 0:      # pack_order.cpp:36 log2(0) is UB on empty input
            run([os.path.join(REF, "pack_order.out"), pd], pd)
            packed = snapshot(os.path.join(pd, "output"))
        shutil.rmtree(pd)
        # decode with the reference decoder (harc:188): decoder.out <dir> <num_thr> <num_thr_e>
        log += run([os.path.join(REF, "decoder.out"), wd, "1", "1"], wd)
        with open(os.path.join(wd, "output", "output.dna"), "rb") as fh:
            decoded = fh.read()
        assert sorted(decoded.decode().split()) == sorted(reads), "reference round trip failed?!"
        meta = dict(name=name, L=L, n_reads=len(reads), gen=kw, log=log.splitlines())
        os.makedirs(GOLD, exist_ok=True)
        tarpath = os.path.join(GOLD, name + ".tar.xz")
        with tarfile.open(tarpath, "w:xz", preset=9) as tf:
            def add(arc, data):
                ti = tarfile.TarInfo(arc)
                ti.size = len(data)
                ti.mtime = 0
                tf.addfile(ti, io.BytesIO(data))
            add("reads.txt", ("\n".join(reads) + "\n").encode())
            for k, v in stage1.items():
                add("stage1/" + k, v)
            for k, v in stage2.items():
                add("stage2/" + k, v)
            for k, v in packed.items():
                add("packed/" + k, v)
            add("decoded.txt", decoded)
            add("meta.json", json.dumps(meta, indent=1).encode())
        print(name, os.path.getsize(tarpath), "bytes;", [l for l in meta["log"] if "unmatched" in l or "aligned" in l])
    finally:
        shutil.rmtree(wd)


# -q cases: ids and quality values (preprocess.cpp:61-118, reorder_quality.cpp).  Each fixture holds in.fastq, p/output.{quality,id}
# (preprocess.out <fastq> <dir> True True L) and np/output.{quality,id} + the two order files reorder_quality.out consumed
# (preprocess False True -> reorder -> encoder at num_thr=1 -> reorder_quality.out).
QCASES = {
    "q_L100_lastN_1500": dict(seed=21, n=1500, L=100, genome_len=9000, err=0.004, last="N"),
    "q_L100_lastclean_1500": dict(seed=22, n=1500, L=100, genome_len=9000, err=0.004, last="clean"),
    "q_L63_1200": dict(seed=23, n=1200, L=63, genome_len=5000, err=0.006, last="clean"),
}


def make_qcase(name, kw):
    L = kw["L"]
    last = kw["last"]
    kw = {k: v for k, v in kw.items() if k != "last"}
    subprocess.check_call([os.path.join(HERE, "build_ref.sh")])
    subprocess.check_call([os.path.join(HERE, "build_ref.sh"), str(L), "1"])
    subprocess.check_call([os.path.join(HERE, "build_ref.sh"), "quality", str(L)])
    reads = gen_reads(**kw)
    rs = np.random.RandomState(kw["seed"] + 1000)
    if last == "N":
        reads[-1] = reads[-1][:10] + "N" + reads[-1][11:]
    else:
        reads[-1] = reads[-1].replace("N", "A")
    recs = []
    for i, r in enumerate(reads):
        q = "".join("#" if c == "N" else "FHJ5"[rs.randint(0, 4)] for c in r)
        rid = "@SRR%d.%d %s/%d" % (kw["seed"], i, "x" * rs.randint(0, 9), 1 + (i & 1))
        recs.append(f"{rid}\n{r}\n+\n{q}\n")
    fastq = "".join(recs).encode()
    out = {}
    for mode, po in (("p", "True"), ("np", "False")):
        wd = tempfile.mkdtemp(prefix="harc_goldq_")
        try:
            os.makedirs(os.path.join(wd, "output"))
            fq = os.path.join(wd, "in.fastq")
            with open(fq, "wb") as f:
                f.write(fastq)
            run([os.path.join(REF, "preprocess.out"), fq, wd, po, "True", str(L)], wd)
            if mode == "np":
                run([os.path.join(REF, f"reorder_L{L}_t1.out"), wd], wd)
                run([os.path.join(REF, f"encoder_L{L}_t1.out"), wd], wd)
                run([os.path.join(REF, f"reorder_quality_L{L}.out"), wd], wd)
                for f in ("read_order.bin", "read_order_N_pe.bin"):
                    out[mode + "/" + f] = open(os.path.join(wd, "output", f), "rb").read()
            for f in ("output.quality", "output.id"):
                out[mode + "/" + f] = open(os.path.join(wd, "output", f), "rb").read()
        finally:
            shutil.rmtree(wd)
    tarpath = os.path.join(GOLD, name + ".tar.xz")
    with tarfile.open(tarpath, "w:xz", preset=9) as tf:
        def add(arc, data):
            ti = tarfile.TarInfo(arc)
            ti.size = len(data)
            ti.mtime = 0
            tf.addfile(ti, io.BytesIO(data))
        add("in.fastq", fastq)
        for k, v in out.items():
            add(k, v)
        add("meta.json", json.dumps(dict(name=name, L=L, n_reads=len(reads), gen=kw, last=last), indent=1).encode())
    print(name, os.path.getsize(tarpath), "bytes")


# Exact mode at CONFIG size (BASELINE.json configs[0] / configs[1]: "bit-exact vs CPU"): the inputs are too large to commit, so the fixture is
# the generator call (tests/gen.reads_array_big, numpy RandomState: stable across versions), the md5 of the reads it makes, and the md5 + size
# of EVERY file the reference leaves at -t 1 after reorder.out and after encoder.out.  tests/test_gpu_full_size.py regenerates the reads on the
# GPU box, checks their md5, runs num_chains = 1 and compares file by file.
MD5CASES = {
    "configs0_1M": dict(seed=20260, n=1_000_000, L=100, genome_len=35_000_000, err=0.0),
    "configs1_3p3M": dict(seed=20261, n=3_300_000, L=100, genome_len=6_300_000, err=0.005),
}


def make_md5case(name, kw):
    import hashlib
    sys.path.insert(0, os.path.dirname(HERE))
    from tests import gen as tgen
    L = kw["L"]
    subprocess.check_call([os.path.join(HERE, "build_ref.sh")])
    subprocess.check_call([os.path.join(HERE, "build_ref.sh"), str(L), "1"])
    arr = tgen.reads_array_big(**kw)
    wd = tempfile.mkdtemp(prefix="harc_goldmd5_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
    try:
        os.makedirs(os.path.join(wd, "output"))
        n = arr.shape[0]
        # FASTQ records "@T.<i>\n<read>\n+\n<H x L>\n" without a Python loop over millions of reads
        fq = os.path.join(wd, "in.fastq")
        with open(fq, "wb") as f:
            q = ("+\n" + "H" * L + "\n").encode()
            for s0 in range(0, n, 100000):
                blk = arr[s0:s0 + 100000]
                f.write(b"".join(b"@T.%d\n" % (s0 + i) + blk[i].tobytes() + b"\n" + q for i in range(blk.shape[0])))
        log = run([os.path.join(REF, "preprocess.out"), fq, wd, "False", "False", str(L)], wd)
        os.remove(fq)
        import time
        t0 = time.time()
        log += run([os.path.join(REF, f"reorder_L{L}_t1.out"), wd], wd)
        t1 = time.time()
        s1 = {k: [hashlib.md5(v).hexdigest(), len(v)] for k, v in snapshot(os.path.join(wd, "output")).items()}
        t1b = time.time()
        log += run([os.path.join(REF, f"encoder_L{L}_t1.out"), wd], wd)
        t2 = time.time()
        s2 = {k: [hashlib.md5(v).hexdigest(), len(v)] for k, v in snapshot(os.path.join(wd, "output")).items()}
        meta = dict(name=name, gen=dict(function="tests.gen.reads_array_big", **kw), reads_md5=hashlib.md5(arr.tobytes()).hexdigest(),
                    reference="reorder.out + encoder.out at num_thr = 1 (oracle/_ref, built by oracle/build_ref.sh from /root/reference)",
                    reference_seconds=dict(reorder=round(t1 - t0, 1), encoder=round(t2 - t1b, 1)),
                    stage1=s1, stage2=s2, log=log.splitlines())
        with open(os.path.join(GOLD, "md5_" + name + ".json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        print(name, meta["reference_seconds"], [l for l in meta["log"] if "unmatched" in l or "aligned" in l])
    finally:
        shutil.rmtree(wd)


if __name__ == "__main__":
    names = sys.argv[1:] or (list(CASES) + list(QCASES) + list(MD5CASES))
    for n in names:
        if n in MD5CASES:
            make_md5case(n, MD5CASES[n])
        elif n in QCASES:
            make_qcase(n, QCASES[n])
        else:
            make_case(n, CASES[n])
