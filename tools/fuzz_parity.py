#!/usr/bin/env python3
"""Parity fuzz on the GPU box: random small inputs rich in what the schedule is sensitive to (exact repeats, diverged repeat families,
poly-A / (CA)n runs, duplicates, N reads, short and long reads), random (K, S, E); libharc_amd.so through its file contract against the
CPU oracle, every stage-I and stage-II file byte for byte, then the decoder round trip.   [FUZZ_K=k] [FUZZ_S=s] python tools/fuzz_parity.py [iterations] [seed]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import harc_amd                                                    # noqa: E402
from tests import oracle_lib as ol                                 # noqa: E402

COMP = np.zeros(256, dtype=np.uint8)
for a, b in zip(b"ACGTN", b"TGCAN"):
    COMP[a] = b
ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)


def make_reads(rs, L):
    glen = int(rs.choice([3000, 8000, 20000, 60000]))
    n = int(rs.choice([1500, 4000, 9000, 16000]))
    g = ACGT[rs.randint(0, 4, size=glen)].copy()
    kind = rs.randint(0, 5)
    if kind >= 1:                                                  # exact or diverged copies of one element
        el = ACGT[rs.randint(0, 4, size=int(rs.choice([120, 300, 700])))]
        div = float(rs.choice([0.0, 0.0, 0.05, 0.12]))
        for p in rs.randint(0, max(1, glen - len(el) - 1), size=int(rs.choice([5, 30, 120]))):
            cp = el.copy()
            m = rs.random_sample(len(el)) < div
            cp[m] = ACGT[rs.randint(0, 4, size=int(m.sum()))]
            g[p:p + len(el)] = cp[:max(0, min(len(el), glen - p))]
    if kind >= 2:
        for p in rs.randint(0, max(1, glen - 400), size=int(rs.choice([1, 4, 12]))):
            g[p:p + int(rs.choice([60, 150, 300]))] = ord("A")
    if kind >= 3:
        for p in rs.randint(0, max(1, glen - 400), size=int(rs.choice([1, 3]))):
            run = np.tile(np.frombuffer(b"CA", dtype=np.uint8), 150)[:int(rs.choice([80, 200]))]
            g[p:p + len(run)] = run[:max(0, min(len(run), glen - p))]
    st = rs.randint(0, glen - L, size=n)
    r = g[st[:, None] + np.arange(L)[None, :]].copy()
    err = float(rs.choice([0.0, 0.003, 0.01]))
    if err > 0:
        e = rs.random_sample((n, L)) < err
        isN = e & (rs.random_sample((n, L)) < 0.25)
        r[e & ~isN] = ACGT[rs.randint(0, 4, size=int((e & ~isN).sum()))]
        r[isN] = ord("N")
    odd = np.arange(n) % 2 == 1
    r[odd] = COMP[r[odd][:, ::-1]]
    ndup = int(rs.choice([0, 0, 50, 1500]))
    if ndup:
        r = np.concatenate([r, np.tile(r[rs.randint(0, n)], (ndup, 1))])
        rs.shuffle(r)
    nbig = int(rs.choice([0, 0, 0, 1300, 2600]))                  # stage-II bins above maxsearch, only partly matching (k_realign_big)
    if nbig:
        src = r[rs.randint(0, r.shape[0])].copy()
        src[src == ord("N")] = ord("A")
        dupn = np.tile(src, (nbig, 1))
        h = L // 2
        for i in np.nonzero(rs.random_sample(nbig) < float(rs.choice([0.0, 0.4, 0.8])))[0]:
            cols = h + rs.choice(L - h, size=min(L - h, 32 * L // 100 + 1), replace=False)
            dupn[i, cols] = ACGT[rs.randint(0, 4, size=len(cols))]
        dupn[np.arange(nbig), h + rs.randint(0, L - h, size=nbig)] = ord("N")
        r = np.concatenate([r, dupn])
        rs.shuffle(r)
    out = np.empty((r.shape[0], L + 1), dtype=np.uint8)
    out[:, :L] = r
    out[:, L] = 10
    return out.tobytes()


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 50
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle = ol.load()
    bad = 0
    for it in range(iters):
        rs = np.random.RandomState(seed0 * 1000 + it)
        L = int(rs.choice([40, 63, 100, 100, 100, 101, 150]))
        txt = make_reads(rs, L)
        nreads = len(txt) // (L + 1)
        K = int(rs.choice([1, 2, 7, 33, 0, nreads // 64 + 1]))
        S = int(rs.choice([1, 4, 16, 16, 32, 64]))
        E = int(rs.choice([1, 2, 5]))
        if os.environ.get("FUZZ_K"): K = int(os.environ["FUZZ_K"])          # e.g. FUZZ_K=1 FUZZ_S=64: exact mode only
        if os.environ.get("FUZZ_S"): S = int(os.environ["FUZZ_S"])
        with tempfile.TemporaryDirectory() as d:
            od, gd = os.path.join(d, "o"), os.path.join(d, "g")
            os.makedirs(od); os.makedirs(gd)
            bo = ol.stage_dir(od, {})
            assert oracle.harc_oracle_preprocess(txt, len(txt), L, bo.encode()) == 0
            inputs = ol.read_dir(bo)
            nclean = len(inputs["input_clean.dna"]) // (L + 1)
            from tests import gen
            Ko = K if K else gen.auto_chains(nclean, clean=inputs["input_clean.dna"])
            assert oracle.harc_oracle_reorder(bo.encode(), L, Ko, S, None, None) == 0
            s1 = ol.read_dir(bo)
            assert oracle.harc_oracle_encoder(bo.encode(), L, E, None, None) == 0
            s2 = ol.read_dir(bo)
            bg = ol.stage_dir(gd, {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
            harc_amd.reorder(bg, L, num_chains=K, num_steps=S)
            g1 = ol.read_dir(bg)
            diff = [f for f in ol.STAGE1_FILES if g1.get(f) != s1[f]]
            if not diff:
                harc_amd.encoder(bg, L, num_thr=E)
                g2 = ol.read_dir(bg)
                diff = [f for f in ol.stage2_files(E) if g2.get(f) != s2[f]]
                if not diff:
                    harc_amd.decoder(bg, E)
                    if sorted(ol.read_dir(bg)["output.dna"].split()) != sorted(txt.split()):
                        diff = ["round trip"]
            print(f"iter {it}: L={L} reads={nreads} K={K} S={S} E={E} -> {'OK' if not diff else 'DIFF ' + ','.join(diff)}", flush=True)
            bad += bool(diff)
    print(f"{iters - bad} / {iters} identical to the oracle")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
