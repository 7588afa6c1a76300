"""Parity at CONFIG size (run with -m gpu): the HIP path against the CPU oracle, every stage-I and stage-II file byte for byte, on
BASELINE.json configs[0] (1 M x 100 bp, 35 Mbp genome, 2.9x) and the configs[1] stand-in (3.3 M x 100 bp, 6.3 Mbp genome, 52x, 0.5 %
errors), with num_chains = 0 -- the library picks K itself, on configs[0] through its low-coverage rule (stage1_run_w: more than 98 %
distinct first-dictionary k-mers -> up to 4096 chains), which tests/gen.auto_chains restates for the oracle.  At these sizes the code
paths that small fixtures only reach through environment overrides run by themselves (whole-bucket fetches vs two-slot fetches, the
bitmaps, thousands of chains, tens of super-rounds).  Plus a bounded run of the parity fuzz (tools/fuzz_parity.py) inside pytest."""
import os
import sys

import numpy as np
import pytest

from tests import gen
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


_ORACLE_SIDE = {}


def _synth_text(n, L, G, err, seed):
    """reads of bench.py's generator (made on the GPU: numpy needs minutes for 330 M bases with errors) -> bytes, one read per line"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    r = bench.synth_reads(n, L, G, err, seed, torch.device("cuda", 0)).cpu().numpy()
    out = np.empty((n, L + 1), dtype=np.uint8)
    out[:, :L] = r
    out[:, L] = 10
    return out.tobytes()


@pytest.mark.parametrize("name,n,L,G,err,E,expect_lowcov,env", [
    ("configs0", 1_000_000, 100, 35_000_000, 0.0, 8, True, {}),
    ("configs1", 3_300_000, 100, 6_300_000, 0.005, 8, False, {}),
    # the kernels a 350 M-read run takes, forced at a size the oracle can follow: dense launch with the wave-uniform scan (8 waves per SIMD),
    # k_reseed by 64 workgroups -- with 3.3 M reads its later rounds need more than one pass of a million bitmap bits to find their seeds
    ("configs1", 3_300_000, 100, 6_300_000, 0.005, 8, False, {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_RESEED_MG": "1"}),
    # ... in its specialised form (SPEC: needs the bitmap lines by minimizer, which a 3.3 M-read bitmap gets only when told so)
    ("configs1", 3_300_000, 100, 6_300_000, 0.005, 8, False, {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0"}),
    # ... and with two chains per wave (k_steps_grp; HARC_AMD_GRP=2 fails the run if that kernel cannot be the one that walks)
    ("configs1", 3_300_000, 100, 6_300_000, 0.005, 8, False, {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_GRP": "2"}),
    ("configs0", 1_000_000, 100, 35_000_000, 0.0, 8, True, {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_GRP": "2", "HARC_AMD_RESEED_MG": "1"}),
    ("configs1", 3_300_000, 100, 6_300_000, 0.005, 8, False, {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_GRP": "2", "HARC_AMD_GRP_WIDE": "1", "HARC_AMD_GRP_WIDE_LIMIT": "40"}),      # at 52x half of the chains are beyond 40 in some column
    ("configs0", 1_000_000, 100, 35_000_000, 0.0, 8, True, {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_RESEED_MG": "1", "HARC_AMD_S1BLOOM_TILED": "1", "HARC_AMD_S1BLOOM_VERIFY": "1"})])
def test_config_size_matches_oracle(name, n, L, G, err, E, expect_lowcov, env, oracle, tmp_path, monkeypatch):
    import harc_amd
    if "HARC_AMD_GRP" in env and not harc_amd.build_has("grp"):
        pytest.skip("k_steps_grp is not in this build (make -C harc_amd/csrc GRP=1)")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    (tmp_path / "g").mkdir()
    # the oracle's side of a configuration is made once and shared by its kernel variants (3.3 M reads: 25 s of the oracle per run)
    if (name, n) not in _ORACLE_SIDE:
        txt = _synth_text(n, L, G, err, 20260 + n % 97)
        (tmp_path / "o").mkdir()
        bo = ol.stage_dir(tmp_path / "o", {})
        assert oracle.harc_oracle_preprocess(txt, len(txt), L, bo.encode()) == 0
        inputs = ol.read_dir(bo)
        nclean = len(inputs["input_clean.dna"]) // (L + 1)
        K_plain = gen.auto_chains(nclean)
        K = gen.auto_chains(nclean, clean=inputs["input_clean.dna"])
        assert (K > K_plain) == expect_lowcov, (K, K_plain)               # configs[0]: the low-coverage rule must be what decides K
        S = gen.auto_steps(inputs["input_clean.dna"], K)                  # ... and of S: 32 at configs[1] (2048 chains, not a low-coverage input, no large bins), 16 at configs[0]
        assert S == (16 if expect_lowcov else 32), S
        assert oracle.harc_oracle_reorder(bo.encode(), L, K, S, None, None) == 0
        s1 = {f: v for f, v in ol.read_dir(bo).items() if f in ol.STAGE1_FILES}
        assert oracle.harc_oracle_encoder(bo.encode(), L, E, None, None) == 0
        s2 = {f: v for f, v in ol.read_dir(bo).items() if f in ol.stage2_files(E)}
        _ORACLE_SIDE[(name, n)] = (txt, {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]}, K, s1, s2)
    txt, inputs, K, s1, s2 = _ORACLE_SIDE[(name, n)]
    bg = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.reorder(bg, L, num_chains=0)                              # the library's own choice of K and S
    g1 = ol.read_dir(bg)
    bad = [f for f in ol.STAGE1_FILES if g1.get(f) != s1[f]]
    assert not bad, f"{name}: stage I differs from the oracle (K = {K}): {bad}"
    harc_amd.encoder(bg, L, num_thr=E)
    g2 = ol.read_dir(bg)
    bad = [f for f in ol.stage2_files(E) if g2.get(f) != s2[f]]
    assert not bad, f"{name}: stage II differs from the oracle: {bad}"
    harc_amd.decoder(bg, E)
    dec = np.frombuffer(ol.read_dir(bg)["output.dna"], dtype=np.uint8).reshape(-1, L + 1)
    src = np.frombuffer(txt, dtype=np.uint8).reshape(-1, L + 1)
    v = np.dtype((np.void, L + 1))
    assert np.array_equal(np.sort(np.ascontiguousarray(dec).view(v).ravel()), np.sort(np.ascontiguousarray(src).view(v).ravel())), f"{name}: round trip"


@pytest.mark.parametrize("it", range(24))
def test_parity_fuzz_bounded(it, oracle, tmp_path, monkeypatch):
    """24 iterations of tools/fuzz_parity.py (seed 7): random repeat-rich inputs, random (L, K, S, E), every file against the oracle; every other
    iteration with the steps by successor list (k_succ) whatever the number of chains"""
    import harc_amd
    if it % 2:
        monkeypatch.setenv("HARC_AMD_SUCC", "1")
    if it % 3 == 0:      # ... and every third with the kernels of a large run forced, the walk by several chains per wave (k_steps_grp) where it can run (L >= 100, S <= 16)
        for k, v in {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_GRP": "1", "HARC_AMD_GRP_WIDE": "1",
                     "HARC_AMD_GRP_WIDE_LIMIT": str(3 + it)}.items():
            monkeypatch.setenv(k, v)
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import fuzz_parity as fz
    rs = np.random.RandomState(7000 + it)
    L = int(rs.choice([40, 63, 100, 100, 100, 101, 150]))
    txt = fz.make_reads(rs, L)
    nreads = len(txt) // (L + 1)
    K = int(rs.choice([1, 2, 7, 33, 0, nreads // 64 + 1]))
    S = int(rs.choice([1, 4, 16, 16, 64]))
    E = int(rs.choice([1, 2, 5]))
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    bo = ol.stage_dir(tmp_path / "o", {})
    assert oracle.harc_oracle_preprocess(txt, len(txt), L, bo.encode()) == 0
    inputs = ol.read_dir(bo)
    nclean = len(inputs["input_clean.dna"]) // (L + 1)
    Ko = K if K else gen.auto_chains(nclean, clean=inputs["input_clean.dna"])
    assert oracle.harc_oracle_reorder(bo.encode(), L, Ko, S, None, None) == 0
    s1 = ol.read_dir(bo)
    assert oracle.harc_oracle_encoder(bo.encode(), L, E, None, None) == 0
    s2 = ol.read_dir(bo)
    bg = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.reorder(bg, L, num_chains=K, num_steps=S)
    g1 = ol.read_dir(bg)
    bad = [f for f in ol.STAGE1_FILES if g1.get(f) != s1[f]]
    assert not bad, f"fuzz {it} (L={L} reads={nreads} K={K} S={S} E={E}): stage I {bad}"
    harc_amd.encoder(bg, L, num_thr=E)
    g2 = ol.read_dir(bg)
    bad = [f for f in ol.stage2_files(E) if g2.get(f) != s2[f]]
    assert not bad, f"fuzz {it} (L={L} reads={nreads} K={K} S={S} E={E}): stage II {bad}"
    harc_amd.decoder(bg, E)
    assert sorted(ol.read_dir(bg)["output.dna"].split()) == sorted(txt.split())


def test_reseed_by_many_workgroups_with_late_readers(monkeypatch):
    """k_reseed_mg (64 workgroups that meet at a counter) when a launch needs SEVERAL passes over the claim bitmap and half of the workgroups
    read the per-pass counts late: the same streams as the single workgroup.  HARC_AMD_RESEED_WIN narrows a pass to 2048 words so that 4 M reads
    take extra passes; HARC_AMD_RESEED_STRESS only delays the odd workgroups.  (With ONE set of per-pass counts a workgroup already counting
    pass p + 1 overwrote what a delayed one still had to read of pass p: seeds handed out twice or never, `reads were never emitted` -- seen
    with three ranks sharing a GPU on 100 M reads, where other processes' kernels delay workgroups by themselves.)"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from tests.test_gpu_replicate import _one_gpu
    n, K = 4_000_000, 16384
    arr = bench.synth_reads(n, 100, int(n * 100 / 26), 0.005, 5, torch.device("cuda", 0)).cpu().numpy()
    monkeypatch.setenv("HARC_AMD_RESEED_MG", "0")
    want, cw = _one_gpu(arr, 100, 2, K, 16)
    monkeypatch.setenv("HARC_AMD_RESEED_MG", "1"); monkeypatch.setenv("HARC_AMD_RESEED_WIN", "2048"); monkeypatch.setenv("HARC_AMD_RESEED_STRESS", "5")
    got, cg = _one_gpu(arr, 100, 2, K, 16)
    bad = [k for k in want if got[k] != want[k]]
    assert not bad, bad
    assert cg.rounds == cw.rounds and cg.n_main == cw.n_main


def test_second_run_of_a_context_allocates_nothing_more():
    """a context keeps its device pool: the second reorder + encode of the same reads must end with the same high-water mark (no growth, no leak)
    and the same streams -- what bench.py's steps and tools/shard_sim_big.py rely on when they call a warm-up pass the steady state"""
    import torch
    sys.path.insert(0, ROOT)
    import bench
    import harc_amd
    from tests.test_gpu_replicate import _collect
    from tests import shard_model
    n, L, E = 1_500_000, 100, 3
    arr = bench.synth_reads(n, L, int(n * 100 / 26), 0.005, 11, torch.device("cuda", 0)).cpu().numpy()
    hasN = (arr == ord("N")).any(1)
    h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=E, num_chains=0, num_steps=16))
    h.set_reads_ascii(shard_model.lines(arr[~hasN]), int((~hasN).sum()), L + 1)
    h.set_nreads_ascii(shard_model.lines(arr[hasN]), int(hasN.sum()), L + 1)
    peaks, files = [], []
    for _ in range(3):
        h.reorder(); h.encode()
        peaks.append(h.counters().device_bytes_peak); files.append(_collect(h, E))
    h.close()
    assert peaks[1] == peaks[0] and peaks[2] == peaks[0], peaks
    assert files[1] == files[0] and files[2] == files[0]
