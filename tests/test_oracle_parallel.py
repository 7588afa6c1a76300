"""K-chain / E-shard semantics of the oracle (the build's own deterministic definition): lossless
round trip through the decoder, and -- in the build container only -- through the REAL reference decoder."""
import ctypes as C
import os
import shutil
import subprocess

import pytest

from tests import oracle_lib as ol

REF_DECODER = os.path.join(ol.ORACLE_DIR, "_ref", "decoder.out")


def run_pipeline(oracle, reads_txt, L, K, E, tmp_path, S=1):
    base = ol.stage_dir(tmp_path, {})
    assert oracle.harc_oracle_preprocess(reads_txt, len(reads_txt), L, base.encode()) == 0
    assert oracle.harc_oracle_reorder(base.encode(), L, K, S, None, None) == 0
    assert oracle.harc_oracle_encoder(base.encode(), L, E, None, None) == 0
    return base


@pytest.mark.parametrize("S", [1, 8, 64])
@pytest.mark.parametrize("case,K,E", [("L100_err_5k", 4, 3), ("L100_err_5k", 64, 8), ("L150_err_3k", 16, 2),
                                        ("L63_err_3k", 7, 5), ("L100_repeat_dup_4k", 32, 4), ("L100_three", 8, 8),
                                        ("L100_allN_20", 2, 2), ("L255_err_1k", 5, 3), ("L100_lowcov_4k", 128, 1)])
def test_roundtrip_K_E(case, K, E, S, oracle, tmp_path):
    g = ol.load_golden(case)
    reads = g["reads.txt"]
    L = len(reads.split(b"\n")[0])
    base = run_pipeline(oracle, reads, L, K, E, tmp_path, S)
    assert oracle.harc_oracle_decoder(base.encode(), E) == 0
    dec = ol.read_dir(base)["output.dna"]
    assert sorted(dec.split()) == sorted(reads.split())
    if os.path.exists(REF_DECODER):  # container only: the real decoder.cpp agrees with the oracle decoder
        os.remove(os.path.join(base, "output", "output.dna"))
        subprocess.check_call([REF_DECODER, base, "1", str(E)], cwd=base, stdout=subprocess.DEVNULL)
        assert ol.read_dir(base)["output.dna"] == dec


def test_K_chains_deterministic_and_close_to_K1(oracle, tmp_path):
    g = ol.load_golden("L100_err_5k")
    reads = g["reads.txt"]
    outs = []
    for rep in range(2):
        d = tmp_path / f"r{rep}"
        d.mkdir()
        base = run_pipeline(oracle, reads, 100, 16, 4, d, 8)
        outs.append(ol.read_dir(base))
    assert outs[0] == outs[1]
    # compression proxy: consensus bytes within 25 % of the K=1 reference stream on this tiny 25 kb genome
    k1 = len(g["stage2/read_seq.txt.0"])
    kK = sum(len(outs[0][f"read_seq.txt.{e}"]) for e in range(4))
    assert kK < 1.25 * k1 + 400
