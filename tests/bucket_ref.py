"""numpy restatement of the multi-GPU shard key (harc_amd/csrc/stage1.hip k_bucket) -- test infrastructure."""
import numpy as np

_PC = np.zeros(256, dtype=np.uint64)
for ch, v in zip(b"AGCT", range(4)):
    _PC[ch] = v
M1, M2 = np.uint64(0xff51afd7ed558ccd), np.uint64(0xc4ceb9fe1a85ec53)


def pack2(reads):
    """[n, L] ASCII -> [n, W] int64 in std::bitset<2L> layout (A0 G1 C2 T3 at bits 2i)"""
    n, L = reads.shape
    W = (2 * L + 63) // 64
    out = np.zeros((n, W), dtype=np.uint64)
    code = _PC[reads]
    for i in range(L):
        out[:, i // 32] |= code[:, i] << np.uint64(2 * (i % 32))
    return out.view(np.int64)


def _mix64(x):
    with np.errstate(over="ignore"):
        x = x ^ (x >> np.uint64(33)); x = x * M1; x = x ^ (x >> np.uint64(33)); x = x * M2; x = x ^ (x >> np.uint64(33))
    return x


def bucket_ref(packed, L, nb):
    p = packed.view(np.uint64)
    n = p.shape[0]
    K = min(L, 15)
    kmask = np.uint64((1 << (2 * K)) - 1)
    fw = np.zeros(n, dtype=np.uint64); rv = np.zeros(n, dtype=np.uint64)
    best = np.full(n, np.uint64(0xFFFFFFFFFFFFFFFF))
    for b in range(L):
        pc = (p[:, b // 32] >> np.uint64(2 * (b % 32))) & np.uint64(3)
        fw = ((fw << np.uint64(2)) | pc) & kmask
        rv = (rv >> np.uint64(2)) | ((np.uint64(3) - pc) << np.uint64(2 * (K - 1)))
        if b >= K - 1:
            h = _mix64(np.minimum(fw, rv))
            best = np.minimum(best, h)
    return (best % np.uint64(nb)).astype(np.int64)


def reads_signature(lines):
    """numpy restatement of the read-multiset signature of harc_amd/csrc/verify.hip: (count, sum, xor) of
    mix64(FNV-1a over base codes A0 C1 G2 T3 N4).  lines: iterable of equal-length byte strings."""
    arr = np.frombuffer(b"".join(lines), dtype=np.uint8).reshape(len(lines), -1) if len(lines) else np.zeros((0, 1), dtype=np.uint8)
    code = np.full(256, 4, dtype=np.uint64)
    for ch, v in zip(b"ACGT", range(4)):
        code[ch] = v
    h = np.full(arr.shape[0], np.uint64(1469598103934665603))
    prime = np.uint64(1099511628211)
    with np.errstate(over="ignore"):
        for j in range(arr.shape[1] if len(lines) else 0):
            h = (h ^ code[arr[:, j]]) * prime
    h = _mix64(h)
    s = int(h.sum(dtype=np.uint64)) if len(lines) else 0
    x = int(np.bitwise_xor.reduce(h)) if len(lines) else 0
    return (len(lines), s, x)


def bucket3_ref(reads_ascii, nb, gid0=0):
    """numpy restatement of harc_amd/csrc/shard.hip k_bucket3: the shard key of a read WITH N -- canonical minimizer (k = 15) over the
    windows that hold no N, packed code A0 G1 C2 T3 as for clean reads; no such window: hash of the read's global id.
    reads_ascii: [n, L] uint8"""
    n, L = reads_ascii.shape
    K = min(L, 15)
    kmask = np.uint64((1 << (2 * K)) - 1)
    fw = np.zeros(n, dtype=np.uint64); rv = np.zeros(n, dtype=np.uint64)
    best = np.full(n, np.uint64(0xFFFFFFFFFFFFFFFF))
    valid = np.zeros(n, dtype=np.int64)
    anyw = np.zeros(n, dtype=bool)
    for b in range(L):
        ch = reads_ascii[:, b]
        isn = ch == ord("N")
        pc = _PC[ch]
        fw = ((fw << np.uint64(2)) | pc) & kmask
        rv = (rv >> np.uint64(2)) | ((np.uint64(3) - pc) << np.uint64(2 * (K - 1)))
        fw[isn] = 0; rv[isn] = 0
        valid = np.where(isn, 0, valid + 1)
        ok = valid >= K
        h = _mix64(np.minimum(fw, rv))
        best = np.where(ok, np.minimum(best, h), best)
        anyw |= ok
    fallback = _mix64((np.uint64(gid0) + np.arange(n, dtype=np.uint64)))
    best = np.where(anyw, best, fallback)
    return (best % np.uint64(nb)).astype(np.int64)


def shard_plan(slices, L, world):
    """What harc_amd_shard_exchange must deliver.  slices: list (one per rank) of [n_r, L] uint8 reads in file order.
    -> per destination rank r: dict(clean=[m, L] reads, gid=global clean ids, withN=[k, L], ngid=global N ids) in the order of
    arrival: source-rank-major, original order inside a source."""
    cl, nn = [], []
    for s in slices:
        hasN = (s == ord("N")).any(1)
        cl.append(s[~hasN]); nn.append(s[hasN])
    coff = np.concatenate([[0], np.cumsum([c.shape[0] for c in cl])]).astype(np.int64)
    noff = np.concatenate([[0], np.cumsum([c.shape[0] for c in nn])]).astype(np.int64)
    out = []
    for r in range(world):
        parts, gids, nparts, ngids = [], [], [], []
        for s in range(len(slices)):
            b = bucket_ref(pack2(cl[s]), L, world) if cl[s].shape[0] else np.zeros(0, dtype=np.int64)
            sel = np.nonzero(b == r)[0]
            parts.append(cl[s][sel]); gids.append(coff[s] + sel)
            b3 = bucket3_ref(nn[s], world, int(noff[s])) if nn[s].shape[0] else np.zeros(0, dtype=np.int64)
            sel3 = np.nonzero(b3 == r)[0]
            nparts.append(nn[s][sel3]); ngids.append(noff[s] + sel3)
        out.append(dict(clean=np.concatenate(parts), gid=np.concatenate(gids).astype(np.uint32),
                        withN=np.concatenate(nparts), ngid=np.concatenate(ngids).astype(np.uint32)))
    return out
