#!/usr/bin/env python3
"""Fuzz of stage II's window passes on the GPU box: random inputs with bins above maxsearch whose reads only partly pass the Hamming test and are probed
from several places of the consensus (tests/gen.py reads_text_bigbin_stage2_mixed, random depth / copies / failing share / read length), every form the
passes can take (an event per lane, a wave per event, that in two kernels; narrow rank ranges, flat passes, no chaser; with and without the two rules
about who looks again) against the CPU oracle, every stage-II file byte for byte.      python tools/fuzz_s2.py [iterations] [seed]"""
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import harc_amd                                                    # noqa: E402
from tests import gen, oracle_lib as ol                            # noqa: E402

FORMS = [{"HARC_AMD_S2_BLOCK": "1"}, {"HARC_AMD_S2_BLOCK": "0"}, {"HARC_AMD_S2_TWOKERNELS": "1"}, {}]
SCHED = [{}, {"HARC_AMD_S2_RANK0": "1"}, {"HARC_AMD_S2_RANK0": "3"}, {"HARC_AMD_S2_FLATPASSES": "1"}, {"HARC_AMD_S2_NOCHASE": "1"}, {"HARC_AMD_S2_RANK0": "2", "HARC_AMD_S2_NOCHASE": "1"}]
RULES = [{}, {}, {"HARC_AMD_S2_RANGE": "0"}, {"HARC_AMD_S2_EBOT": "0"}, {"HARC_AMD_S2_RANGE": "0", "HARC_AMD_S2_EBOT": "0"}, {"HARC_AMD_S2_COMPACT": "0"}, {"HARC_AMD_S2_COMPACT": "0", "HARC_AMD_S2_EBOT": "0"}]
KEYS = sorted({k for group in (FORMS, SCHED, RULES) for e in group for k in e})


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    oracle = ol.load()
    bad = 0
    for it in range(iters):
        rs = np.random.RandomState(seed0 * 1000 + it)
        L = int(rs.choice([100, 100, 100, 101, 150, 63]))
        ndup = int(rs.choice([1300, 2200, 3000, 4300, 6100]))
        copies = int(rs.choice([1, 2, 3, 5]))
        fail = float(rs.choice([0.0, 0.3, 0.5, 0.8]))
        nsub = max(4, min(L - 52, int(rs.choice([20, 32, 40]))))
        txt = gen.reads_text_bigbin_stage2_mixed(int(rs.randint(1, 1 << 30)), n_clean=int(rs.choice([2000, 4000])), n_dupN=ndup, L=L, genome_len=int(rs.choice([5000, 9000])),
                                                 copies=copies, fail_frac=fail, nsub=nsub)
        K = int(rs.choice([1, 2, 4, 8])); S = 16; E = int(rs.choice([1, 2, 3]))
        with tempfile.TemporaryDirectory() as d:
            od = os.path.join(d, "o"); os.makedirs(od)
            bo = ol.stage_dir(od, {})
            assert oracle.harc_oracle_preprocess(txt, len(txt), L, bo.encode()) == 0
            inputs = ol.read_dir(bo)
            assert oracle.harc_oracle_reorder(bo.encode(), L, K, S, None, None) == 0
            assert oracle.harc_oracle_encoder(bo.encode(), L, E, None, None) == 0
            s2 = ol.read_dir(bo)
            left = len(s2["input_N.dna"]) // (L + 1)
            res = []
            for v in range(4):                                     # four random combinations per input
                env = {}
                for group in (FORMS, SCHED, RULES):
                    env.update(group[rs.randint(0, len(group))])
                for k in KEYS:
                    os.environ.pop(k, None)
                os.environ.update(env)
                gd = os.path.join(d, "g%d" % v); os.makedirs(gd)
                bg = ol.stage_dir(gd, {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
                harc_amd.compress(bg, L, num_thr=E, num_chains=K, num_steps=S)
                g2 = ol.read_dir(bg)
                diff = [f for f in ol.stage2_files(E) if g2.get(f) != s2[f]]
                res.append((env, diff))
                bad += bool(diff)
            for k in KEYS:
                os.environ.pop(k, None)
            msg = "OK" if not any(df for _, df in res) else "DIFF " + "; ".join("%r: %s" % (e, ",".join(df)) for e, df in res if df)
            print(f"iter {it}: L={L} dupN={ndup} copies={copies} fail={fail} K={K} E={E} left={left} -> {msg}", flush=True)
    print(f"{4 * iters - bad} / {4 * iters} runs identical to the oracle")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
