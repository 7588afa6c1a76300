"""GPU parity tests proper (run with -m gpu on an MI355X): libharc_amd.so through its C-ABI file contract vs
 (P1) the golden vectors of the REAL reference at num_thr=1 (K=1, E=1), byte for byte;
 (P4) the reference's own stage-I files fed to the HIP encoder;
 (P2) the CPU oracle for K>1 chains / E>1 shards on the same seeded inputs, byte for byte;
 (P3) decode round trips at larger sizes."""
import ctypes as C
import os

import pytest

from tests import gen
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
CASES = ol.golden_cases()


def _L(g):
    return len(g["reads.txt"].split(b"\n")[0])


def _diff(name, a, b):
    if a == b:
        return None
    n = min(len(a), len(b))
    first = next((i for i in range(n) if a[i] != b[i]), n)
    return f"{name}: len {len(a)} vs {len(b)}, first difference at byte {first}: {a[first:first+16]!r} vs {b[first:first+16]!r}"


def assert_same(got, want, files, what):
    errs = [d for d in (_diff(f, got.get(f, b"<missing>"), want[f]) for f in files) if d]
    assert not errs, what + "\n" + "\n".join(errs)


@pytest.mark.parametrize("env", [{}, {"HARC_AMD_SUCC": "0"}, {"HARC_AMD_FUSED_RR": "0"}])
@pytest.mark.parametrize("case", CASES)
def test_stage1_K1_matches_reference(case, env, tmp_path, monkeypatch):
    """exact mode: by default through the successor lists (k_succ) and with k_resolve + k_reseed in one launch; without the lists; with two launches"""
    import harc_amd
    if "HARC_AMD_GRP" in env and not harc_amd.build_has("grp"):
        pytest.skip("k_steps_grp is not in this build (make -C harc_amd/csrc GRP=1)")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    g = ol.load_golden(case)
    base = ol.stage_dir(tmp_path, {k: g["stage1/" + k] for k in ["input_clean.dna", "numreads.bin"]})
    harc_amd.reorder(base, _L(g), num_chains=1)
    assert_same(ol.read_dir(base), {f: g["stage1/" + f] for f in ol.STAGE1_FILES}, ol.STAGE1_FILES, f"{case}: stage I vs reference -t 1")


@pytest.mark.parametrize("case", CASES)
def test_stage2_E1_on_reference_stage1_files(case, tmp_path):
    import harc_amd
    g = ol.load_golden(case)
    base = ol.stage_dir(tmp_path, {k[len("stage1/"):]: v for k, v in g.items() if k.startswith("stage1/")})
    harc_amd.encoder(base, _L(g), num_thr=1)
    fs = ol.stage2_files(1)
    assert_same(ol.read_dir(base), {f: g["stage2/" + f] for f in fs}, fs, f"{case}: stage II vs reference -t 1")


@pytest.mark.parametrize("case", CASES)
def test_fused_compress_and_pack_order_match_reference(case, tmp_path):
    import harc_amd
    g = ol.load_golden(case)
    base = ol.stage_dir(tmp_path, {k: g["stage1/" + k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.compress(base, _L(g), num_thr=1, num_chains=1)
    fs = ol.stage2_files(1)
    got = ol.read_dir(base)
    assert_same(got, {f: g["stage2/" + f] for f in fs}, fs, f"{case}: fused compress vs reference -t 1")
    if "packed/read_order.bin" in g:
        harc_amd.pack_order(base, _L(g))
        got = ol.read_dir(base)
        assert got["read_order.bin"] == g["packed/read_order.bin"]
        assert got["read_order.bin.tail"] == g["packed/read_order.bin.tail"]
    else:
        with pytest.raises(harc_amd.HarcAmdError):
            harc_amd.pack_order(base, _L(g))


def _oracle_pipeline(oracle, reads_txt, L, K, E, d, S=1):
    base = ol.stage_dir(d, {})
    assert oracle.harc_oracle_preprocess(reads_txt, len(reads_txt), L, base.encode()) == 0
    inputs = ol.read_dir(base)
    assert oracle.harc_oracle_reorder(base.encode(), L, K, S, None, None) == 0
    s1 = ol.read_dir(base)
    assert oracle.harc_oracle_encoder(base.encode(), L, E, None, None) == 0
    return inputs, s1, ol.read_dir(base), base


@pytest.mark.parametrize("S", [1, 4, 16, 32, 64])
@pytest.mark.parametrize("case,K,E", [("L100_err_5k", 4, 3), ("L100_err_5k", 64, 8), ("L150_err_3k", 16, 2), ("L63_err_3k", 7, 5),
                                        ("L100_repeat_dup_4k", 32, 4), ("L100_three", 8, 8), ("L100_allN_20", 2, 2),
                                        ("L255_err_1k", 5, 3), ("L100_lowcov_4k", 128, 1), ("L40_err_3k", 300, 2), ("L101_err_3k", 9, 4)])
def test_K_chains_E_shards_match_oracle(case, K, E, S, oracle, tmp_path):
    import harc_amd
    g = ol.load_golden(case)
    L = _L(g)
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    inputs, s1, s2, _ = _oracle_pipeline(oracle, g["reads.txt"], L, K, E, tmp_path / "o", S)
    base = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.reorder(base, L, num_chains=K, num_steps=S)
    assert_same(ol.read_dir(base), s1, ol.STAGE1_FILES, f"{case} K={K} S={S}: stage I vs oracle")
    harc_amd.encoder(base, L, num_thr=E)
    fs = ol.stage2_files(E)
    assert_same(ol.read_dir(base), s2, fs, f"{case} K={K} E={E}: stage II vs oracle")


@pytest.mark.parametrize("n,L,glen,err,K,E,S", [(60000, 100, 300000, 0.01, 64, 8, 16), (40000, 100, 2000000, 0.0, 40, 3, 5), (30000, 150, 200000, 0.01, 0, 8, 0)])
def test_medium_vs_oracle_and_roundtrip(n, L, glen, err, K, E, S, oracle, tmp_path):
    import harc_amd
    txt = gen.reads_text(1234 + n, n, L, glen, err=err)
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    Ko = K if K else gen.auto_chains(txt.count(b"\n") - sum(1 for l in txt.split(b"\n") if b"N" in l), clean=txt)   # auto_chains() in stage1.hip
    inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, L, Ko, E, tmp_path / "o", S if S else 16)
    base = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.compress(base, L, num_thr=E, num_chains=K, num_steps=S)
    fs = ol.stage2_files(E)
    assert_same(ol.read_dir(base), s2, fs, "fused compress vs oracle")
    assert oracle.harc_oracle_decoder(base.encode(), E) == 0
    assert sorted(ol.read_dir(base)["output.dna"].split()) == sorted(txt.split())


def test_in_memory_api_and_counters(oracle, tmp_path):
    import harc_amd
    g = ol.load_golden("L100_err_5k")
    clean, withN = g["stage1/input_clean.dna"], g["stage1/input_N.dna"]
    p = harc_amd.default_params(100, num_thr=1, num_chains=1, profile=1)
    with harc_amd.HarcAmd(p) as h:
        h.set_reads_ascii(clean, len(clean) // 101, 101)
        h.set_nreads_ascii(withN, len(withN) // 101, 101)
        h.reorder()
        assert h.stream("S1_ORDER") == g["stage1/read_order.bin"]
        assert h.stream("S1_DNA") == g["stage1/temp.dna"]
        h.encode()
        assert h.stream("S2_SEQ", 0) == g["stage2/read_seq.txt.0"]
        assert h.stream("S2_NOISE", 0) == g["stage2/read_noise.txt.0"]
        h.pack_order()
        assert h.stream("P_ORDER") == g["packed/read_order.bin"]
        c = h.counters()
        assert c.unmatched == 346 and c.aligned_singletons == 239 and c.aligned_N == 979      # the reference's printed counters
        assert c.propose_launches > 0 and c.propose_ms > 0 and c.probes > 0


def test_bad_arguments_fail_loudly():
    import harc_amd
    with pytest.raises(harc_amd.HarcAmdError):
        harc_amd.default_params(256)          # harc:46-49 / decoder.cpp:93: readlen 256 is not representable
    p = harc_amd.default_params(100)
    with harc_amd.HarcAmd(p) as h:
        with pytest.raises(harc_amd.HarcAmdError):
            h.encode()                        # no stage-I result yet


def test_bucket_kernel_matches_numpy_restatement():
    import numpy as np
    import torch
    import harc_amd
    from tests.bucket_ref import pack2, bucket_ref
    reads = gen.reads_array(3, 20000, 100, 100000, err=0.0)
    p = harc_amd.default_params(100)
    with harc_amd.HarcAmd(p) as h:
        d = torch.from_numpy(reads).cuda()
        packed = torch.empty((reads.shape[0], 4), dtype=torch.int64, device="cuda")
        h.pack_reads_device(d.data_ptr(), reads.shape[0], 100, packed.data_ptr())
        assert (packed.cpu().numpy() == pack2(reads)).all()
        b = torch.empty((reads.shape[0],), dtype=torch.int32, device="cuda")
        h.bucket_reads_device(packed.data_ptr(), reads.shape[0], 8, b.data_ptr())
        assert (b.cpu().numpy() == bucket_ref(pack2(reads), 100, 8)).all()


@pytest.mark.parametrize("case,K,S,E", [("L100_err_5k", 1, 16, 1), ("L100_err_5k", 16, 8, 4), ("L150_err_3k", 8, 16, 3), ("L63_err_3k", 5, 4, 2),
                                          ("L255_err_1k", 3, 16, 2), ("L100_allN_20", 1, 1, 2), ("L100_repeat_dup_4k", 32, 16, 8)])
def test_gpu_decoder_signature_matches_oracle_decoder_and_inputs(case, K, S, E, oracle, tmp_path):
    """decode side on the GPU (verify.hip) == the oracle's restatement of decoder.cpp == the input reads, as multiset signatures"""
    import torch
    import harc_amd
    from tests.bucket_ref import reads_signature
    g = ol.load_golden(case)
    L = _L(g)
    clean, withN = g["stage1/input_clean.dna"], g["stage1/input_N.dna"]
    p = harc_amd.default_params(L, num_thr=E, num_chains=K, num_steps=S)
    with harc_amd.HarcAmd(p) as h:
        h.set_reads_ascii(clean, len(clean) // (L + 1), L + 1)
        h.set_nreads_ascii(withN, len(withN) // (L + 1), L + 1)
        h.reorder(); h.encode()
        sig = h.decode_signature()
        allreads = torch.frombuffer(bytearray(g["reads.txt"]), dtype=torch.uint8).cuda()
        sig_in = h.reads_signature_device(allreads.data_ptr(), len(g["reads.txt"]) // (L + 1), L + 1)
        # write the streams and decode them with the oracle (CPU restatement of decoder.cpp)
        base = ol.stage_dir(tmp_path, {})
        names = {"S2_SEQ": "read_seq.txt.%d", "S2_SEQ_TAIL": "read_seq.txt.%d.tail", "S2_POS": "read_pos.txt.%d", "S2_NOISE": "read_noise.txt.%d",
                 "S2_NOISEPOS": "read_noisepos.txt.%d", "S2_REV": "read_rev.txt.%d", "S2_REV_TAIL": "read_rev.txt.%d.tail"}
        files = {fmt % e: h.stream(k, e) for k, fmt in names.items() for e in range(E)}
        files.update({"read_singleton.txt": h.stream("S2_SINGLETON"), "read_singleton.txt.tail": h.stream("S2_SINGLETON_TAIL"),
                      "input_N.dna": h.stream("S2_INPUT_N"), "read_meta.txt": h.stream("S2_META")})
        ol.stage_dir(tmp_path, files)
    assert oracle.harc_oracle_decoder(base.encode(), E) == 0
    dec = ol.read_dir(base)["output.dna"].split()
    assert sig == reads_signature(dec)                          # GPU decoder == oracle decoder
    assert sig == sig_in == reads_signature(g["reads.txt"].split())   # == the input multiset


@pytest.mark.parametrize("n,glen,K,S,E,err", [(30000, 60000, 16, 16, 4, 0.0), (30000, 60000, 1, 64, 1, 0.002), (40000, 40000, 64, 8, 2, 0.0)])
def test_low_complexity_big_bins_match_oracle(n, glen, K, S, E, err, oracle, tmp_path):
    """repeats and poly-A runs: bins far above maxsearch, scanned by the whole wave in k_steps -- same bytes as the oracle's serial scan"""
    import harc_amd
    txt = gen.reads_text_lowcomplexity(4321, n, 100, glen, err=err)
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, 100, K, E, tmp_path / "o", S)
    base = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.reorder(base, 100, num_chains=K, num_steps=S)
    assert_same(ol.read_dir(base), s1, ol.STAGE1_FILES, "low-complexity stage I vs oracle")
    harc_amd.encoder(base, 100, num_thr=E)
    # stage II: identical too -- bins above maxsearch see the sliding window of encoder.cpp:293 exactly (k_realign_big)
    assert_same(ol.read_dir(base), s2, ol.stage2_files(E), "low-complexity stage II vs oracle")


@pytest.mark.parametrize("K,S", [(12, 16), (1, 64), (40, 8)])
def test_huge_bin_compacted_from_its_top_matches_oracle(K, S, oracle, tmp_path):
    """a stage-I bin of 9000 reads (k_compact_huge: only the stretch at the top of the bin that a scan can reach -- maxsearch unclaimed entries -- is
    compacted each super-round, the rest when the passes get down to it): the scans see what the oracle's serial scan over the whole bin sees"""
    import harc_amd
    txt = gen.reads_text_hugebin_stage1(2026)
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, 100, K, 2, tmp_path / "o", S)
    base = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.reorder(base, 100, num_chains=K, num_steps=S)
    assert_same(ol.read_dir(base), s1, ol.STAGE1_FILES, "huge-bin stage I vs oracle")
    harc_amd.encoder(base, 100, num_thr=2)
    assert_same(ol.read_dir(base), s2, ol.stage2_files(2), "huge-bin stage II vs oracle")


@pytest.mark.parametrize("n", [1, 255, 1 << 20, (1 << 32) - 256, (1 << 32) - 1, 1 << 32, (1 << 32) + 1, 39_000_000 * 256, (1 << 33) + 12345])
def test_launches_of_more_than_2_32_work_items_visit_every_item(n):
    """a grid is dispatched with its size in work-items as a 32-bit number per dimension: 2^32 and more is taken modulo 2^32 WITHOUT an error
    (tools/micro/grid_limit.hip).  The library's thread-per-item launches fold their workgroups into rows (devutil.h harc_grid256 / harc_gid): every item
    is visited exactly once whatever n, through all three geometries of the library (harc_gid, harc_gid32, folded workgroups of four lanes per item) -- found when 402 M probes into stage II's large bins (x 64 lanes) were cut short and the window passes ended early"""
    import harc_amd
    h = harc_amd.HarcAmd(harc_amd.default_params(100))
    try:
        v, sx = h.selftest_launch(n)
    finally:
        h.close()
    assert v == n and sx == (n * (n - 1) // 2) % (1 << 64)


def test_table_slots_per_read_is_not_visible_in_the_bytes(oracle, tmp_path):
    """harc_amd_params.table_slots_per_read = 2 / 3 / 4 (a memory / speed choice of the caller): the same streams"""
    import harc_amd
    arr = gen.reads_array(606, 9000, 100, 70000, err=0.01)
    hasN = (arr == ord("N")).any(1)
    outs = []
    for m in (0, 2, 3, 4):
        h = harc_amd.HarcAmd(harc_amd.default_params(100, num_thr=2, num_chains=12, num_steps=16, table_slots_per_read=m))
        h.set_reads_ascii(gen.lines_of(arr[~hasN]), int((~hasN).sum()), 101)
        h.set_nreads_ascii(gen.lines_of(arr[hasN]), int(hasN.sum()), 101)
        h.reorder(); h.encode()
        outs.append([h.stream(sid, e) for e in range(2) for sid in ("S2_SEQ", "S2_POS", "S2_NOISE", "S2_NOISEPOS", "S2_REV")] + [h.stream("S2_ORDER"), h.stream("S2_SINGLETON")])
        h.close()
    assert outs[1] == outs[0] and outs[2] == outs[0] and outs[3] == outs[0]


def test_scan_choice_of_a_context_is_measured_once_and_never_visible_in_the_bytes(monkeypatch):
    """a dense run (more than 16 384 chains) over mostly single-read bins MEASURES which scan of the small bins is faster -- eight super-rounds with
    each -- and the context keeps the answer for later inputs of the same shape (round 5: the eight rounds with the slower one were 3.5 ms of every
    configs[2] step).  Both scans compute the same thing: the run that measures, the run that starts from the kept answer, a run made to measure
    again and a run with the other scan forced produce the same streams"""
    import harc_amd
    arr = gen.reads_array(707, 60000, 100, 500000, err=0.002)
    hasN = (arr == ord("N")).any(1)
    h = harc_amd.HarcAmd(harc_amd.default_params(100, num_thr=2, num_chains=17000, num_steps=16))
    outs = []
    try:
        h.set_reads_ascii(gen.lines_of(arr[~hasN]), int((~hasN).sum()), 101)
        h.set_nreads_ascii(gen.lines_of(arr[hasN]), int(hasN.sum()), 101)
        for rep in range(5):
            if rep == 2:
                monkeypatch.setenv("HARC_AMD_SEQ_REMEASURE", "1")
            if rep >= 3:
                monkeypatch.setenv("HARC_AMD_SEQ", str(rep - 3))
            h.reorder(); h.encode()
            outs.append([h.stream(sid, e) for e in range(2) for sid in ("S2_SEQ", "S2_POS", "S2_NOISE", "S2_NOISEPOS", "S2_REV")] + [h.stream("S2_SINGLETON")])
    finally:
        h.close()
    assert all(o == outs[0] for o in outs[1:])


def test_back_off_of_chains_that_keep_losing_bids_matches_the_oracle(oracle, tmp_path):
    """more than 16 384 chains on a repeat-rich input (the bins of more than 16 reads hold more than 2 % of N entries): a chain whose walks keep being cut
    at a lost bid sits out 0, 1, 3, 7 ... super-rounds (stage1.hip HARC_BO_*, oracle BO_*).  The same schedule on both sides: every file is the oracle's"""
    import harc_amd
    txt = gen.reads_text_lowcomplexity(99, 400000, 100, 1000000, n_repeat=600, n_polya=12, err=0.004)
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    K, S, E = 20000, 16, 2
    inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, 100, K, E, tmp_path / "o", S)
    base = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.reorder(base, 100, num_chains=K, num_steps=S)
    assert_same(ol.read_dir(base), s1, ol.STAGE1_FILES, "stage I with the back-off vs oracle")
    harc_amd.encoder(base, 100, num_thr=E)
    assert_same(ol.read_dir(base), s2, ol.stage2_files(E), "stage II behind it vs oracle")


def test_steps_per_super_round_chosen_from_the_index(oracle, tmp_path):
    """num_steps = 0 with more than 16 384 chains: 32 steps per super-round where the bins of more than 16 reads hold less than 2 % of N entries (clean
    data), 16 where repeat families fill such bins (longer walks of chains that meet lose more of what they walked: configs[3] with human-like repeats,
    profiles/r05/s_choice_trace.txt).  A function of the input alone: the automatic choice gives the ORACLE's bytes for that S (every file; round 5
    compared the library with itself here), and the other S gives other bytes on at least one of the inputs.  tests/test_gpu_many_chains.py has the same
    at 20 000 / 65 536 chains on 1-2 M reads"""
    import harc_amd
    clean = gen.lines_of(gen.reads_array(707, 60000, 100, 500000, err=0.002))
    rich = gen.reads_text_lowcomplexity(99, 40000, 100, 100000, err=0.004)
    K, E = 17000, 2
    differ = False
    for name, txt, S_rule in (("clean", clean, 32), ("rich", rich, 16)):
        want = {}
        for S in (16, 32):
            d = tmp_path / f"o_{name}_{S}"; d.mkdir()
            inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, 100, K, E, d, S)
            want[S] = (s1, s2)
        differ = differ or any(want[16][1][f] != want[32][1][f] for f in ol.stage2_files(E))
        g = tmp_path / f"g_{name}"; g.mkdir()
        base = ol.stage_dir(g, {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
        harc_amd.reorder(base, 100, num_chains=K, num_steps=0)
        assert_same(ol.read_dir(base), want[S_rule][0], ol.STAGE1_FILES, f"{name}: stage I with the library's choice of S vs the oracle at S = {S_rule}")
        harc_amd.encoder(base, 100, num_thr=E)
        assert_same(ol.read_dir(base), want[S_rule][1], ol.stage2_files(E), f"{name}: stage II behind it vs the oracle at S = {S_rule}")
    assert differ          # (the two schedules do differ on at least one of the inputs: the test can tell them apart)


@pytest.mark.parametrize("env", [{"HARC_AMD_QUAD": "0"}, {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1"},
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_S1BLOOM_MZMB": "0"}, {"HARC_AMD_S1BLOOM_MZMB": "0"},
                                 {"HARC_AMD_S1BLOOM": "0"}, {"HARC_AMD_BLOOM4_HASHED": "1"}, {"HARC_AMD_BLOOM1": "1"}, {"HARC_AMD_CAPMULT": "2"},
                                 {"HARC_AMD_COOP_WAVES": "1"}, {"HARC_AMD_COOP_WAVES": "2"}, {"HARC_AMD_RESEED_MG": "1"},
                                 {"HARC_AMD_RESEED_MG": "1", "HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1"},
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1"}, {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "0"},
                                 {"HARC_AMD_LAZY": "0"}, {"HARC_AMD_LAZY": "0", "HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1"},
                                 {"HARC_AMD_SUCC": "1"}, {"HARC_AMD_SUCC": "1", "HARC_AMD_S1BLOOM": "0"}, {"HARC_AMD_SUCC": "1", "HARC_AMD_S1BLOOM_MZMB": "0"},      # steps by successor list (k_succ), with chains that lose bids and are rolled back
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0"},      # the specialised dense kernel (k_steps' SPEC) ...
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_SPEC": "0"},      # ... and the general one under the same conditions
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_LAZY": "0"},
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_GRP": "2"},      # two chains per wave (k_steps_grp; 2 = fail if it cannot run)
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_GRP": "2", "HARC_AMD_LAZY": "0"},
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_GRP": "2", "HARC_AMD_RESEED_MG": "1", "HARC_AMD_WEEDMIN": "1"},
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_GRP": "2", "HARC_AMD_GRP_G": "32"},      # two chains per wave instead of four
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_GRP": "2", "HARC_AMD_GRP_WIDE": "1", "HARC_AMD_GRP_WIDE_LIMIT": "6"},      # chains with a count above 6 change to the u32 form of the kernel
                                 {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_GRP": "2", "HARC_AMD_GRP_G": "32", "HARC_AMD_GRP_WIDE": "1", "HARC_AMD_GRP_WIDE_LIMIT": "2", "HARC_AMD_LAZY": "0"},
                                 {"HARC_AMD_LEFT_ALL": "1"},      # the leftover reads emitted by a pass over all candidates instead of over their list
                                 {"HARC_AMD_TABLE_FILL": "0"}, {"HARC_AMD_TABLE_FILL": "1"}, {"HARC_AMD_SORT_BITS": "8"}, {"HARC_AMD_SORT_BITS": "13"}, {"HARC_AMD_SORT_BITS": "1"}, {"HARC_AMD_SORT_BITS": "64"},
                                 {"HARC_AMD_S2BLOOM_TILED": "1", "HARC_AMD_S2BLOOM_VERIFY": "1"}, {"HARC_AMD_S2BLOOM_TILED": "1", "HARC_AMD_S2BLOOM_VERIFY": "1", "HARC_AMD_BLOOM4_HASHED": "1"},
                                 {"HARC_AMD_S1BLOOM_TILED": "1", "HARC_AMD_S1BLOOM_VERIFY": "1"},
                                 {"HARC_AMD_S1BLOOM_TILED": "1", "HARC_AMD_S1BLOOM_VERIFY": "1", "HARC_AMD_S1BLOOM_MZMB": "0", "HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1"}])
def test_kernel_variants_same_bytes(env, oracle, tmp_path, monkeypatch):
    """the variants the library picks by problem size (two-slot vs whole-bucket fetches, 5 vs 6 waves per SIMD, bitmap lines hashed vs by
    minimizer, with / without the bitmaps, the table cleared by a memset or by the placement itself, the index sorted on its top 8 / 13 bits with the mixed stretches fixed up, on 1 bit (stretches too long: falls back to all 64) and on all bits, the stage-I bitmap built tile by tile from sorted keys instead of with atomics (and compared with it word for word), the column counts applied step by step instead of a run of agreeing steps at once (HARC_AMD_LAZY=0), stage-II bitmap kinds, a fuller table, 1 / 2 / 4 waves per cooperative workgroup) are execution details: forced on a small repeat-rich
    input, every stage-I and stage-II file is the oracle's"""
    import harc_amd
    if "HARC_AMD_GRP" in env and not harc_amd.build_has("grp"):
        pytest.skip("k_steps_grp is not in this build (make -C harc_amd/csrc GRP=1)")
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    txt = gen.reads_text_lowcomplexity(99, 20000, 100, 50000, err=0.004)
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    K, S, E = 24, 16, 3
    inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, 100, K, E, tmp_path / "o", S)
    base = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.reorder(base, 100, num_chains=K, num_steps=S)
    assert_same(ol.read_dir(base), s1, ol.STAGE1_FILES, "stage I vs oracle under %r" % env)
    harc_amd.encoder(base, 100, num_thr=E)
    assert_same(ol.read_dir(base), s2, ol.stage2_files(E), "stage II vs oracle under %r" % env)


@pytest.mark.parametrize("K,S,E,maxev,ndup,sched", [(1, 16, 1, 0, 2500, ""), (8, 16, 3, 0, 2500, "rank0=1"), (4, 16, 2, 7, 2500, ""), (3, 16, 1, 0, 5200, ""), (3, 16, 1, 0, 5200, "flat"), (3, 16, 1, 0, 5200, "rank0=2"),
                                                     (3, 16, 1, 0, 5200, "two"), (8, 16, 3, 0, 2500, "rank0=1,two"), (3, 16, 1, 0, 5200, "flat,two"),
                                                     (3, 16, 1, 0, 5200, "wave"), (8, 16, 3, 0, 2500, "rank0=1,wave"), (3, 16, 1, 0, 5200, "flat,wave"), (4, 16, 2, 7, 2500, "wave"), (3, 16, 1, 0, 5200, "auto")])
def test_stage2_bins_above_maxsearch_sliding_window_exact(K, S, E, maxev, ndup, sched, oracle, tmp_path, monkeypatch):
    """ndup N reads sharing their first 50 bases: both stage-II dictionaries hold a bin of ndup > maxsearch.  The reference's
    window slides over the still-unclaimed ids (encoder.cpp:293,321-336); the GPU settles those probes as a fixed point over all of them
    at once (k_realign_big) -- same bytes as the sequential oracle.  maxev: a deliberately tiny event buffer, so that the pass has to be
    repeated with the size it asks for; ndup = 5200: five windows deep.  sched: the passes go over rank ranges of the events inside their
    bin (first range 64 wide; forced to 1 or 2 here so that a small input walks through many ranges) or, "flat", over all events every time;
    the looks with an event per lane (k_realign_block, the default), with a wave per event ("wave"), or that in two kernels ("two")."""
    import harc_amd
    if maxev:
        monkeypatch.setenv("HARC_AMD_MAXEVENTS", str(maxev))
    _set_sched(monkeypatch, sched)
    txt = gen.reads_text_bigbin_stage2(77, n_dupN=ndup)
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, 100, K, E, tmp_path / "o", S)
    base = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.compress(base, 100, num_thr=E, num_chains=K, num_steps=S)
    fs = ol.stage2_files(E)
    assert_same(ol.read_dir(base), s2, fs, "stage II with bins above maxsearch vs oracle")
    # the case really exercises the window: more than maxsearch N reads get aligned
    assert len(s2["read_order_N_pe.bin"]) // 4 == ndup and len(s2["input_N.dna"]) < (ndup - 1000) * 101


def _set_sched(monkeypatch, sched):
    # the looks of a pass with an event per lane (k_realign_block) are the library's choice from a million events of a pass on: forced here, where the inputs
    # are small, unless the case asks for a wave per event ("wave", "two") or for the library's own choice ("auto")
    parts = sched.split(",")
    if not ("wave" in parts or "two" in parts or "auto" in parts):
        monkeypatch.setenv("HARC_AMD_S2_BLOCK", "1")
    for part in parts:
        if part == "flat":
            monkeypatch.setenv("HARC_AMD_S2_FLATPASSES", "1")
        elif part == "nochase":
            monkeypatch.setenv("HARC_AMD_S2_NOCHASE", "1")
        elif part.startswith("rank0="):
            monkeypatch.setenv("HARC_AMD_S2_RANK0", part[6:])
        elif part == "nocompact":                                  # the block form walks the whole bins, not what is left of them for the events that look
            monkeypatch.setenv("HARC_AMD_S2_COMPACT", "0")
        elif part == "noebot":                                     # an event that looks again tests its whole window again
            monkeypatch.setenv("HARC_AMD_S2_EBOT", "0")
        elif part == "norange":                                    # every event behind the earliest moved claim of its bin looks again (not only those between the two tuples of a moved claim)
            monkeypatch.setenv("HARC_AMD_S2_RANGE", "0")
        elif part == "wave":                                       # a wave per event (k_realign_big) instead of an event per lane (k_realign_block)
            monkeypatch.setenv("HARC_AMD_S2_BLOCK", "0")
        elif part == "two":                                        # the passes in two kernels (a thread per event asks who has to look, a wave per event that has to), as on inputs with millions of such probes
            monkeypatch.setenv("HARC_AMD_S2_TWOKERNELS", "1")


@pytest.mark.parametrize("K,S,E,seed,fail,sched", [(1, 16, 1, 5, 0.6, ""), (4, 16, 2, 6, 0.5, "rank0=3"), (1, 16, 1, 7, 0.8, "flat"), (2, 8, 1, 8, 0.3, "rank0=1"), (1, 16, 1, 9, 0.5, "rank0=2"), (1, 16, 1, 9, 0.5, "nochase"),
                                                    (4, 16, 2, 6, 0.5, "rank0=3,two"), (1, 16, 1, 7, 0.8, "flat,two"), (1, 16, 1, 9, 0.5, "two"),
                                                    (4, 16, 2, 6, 0.5, "rank0=3,wave"), (1, 16, 1, 7, 0.8, "flat,wave"), (1, 16, 1, 9, 0.5, "wave"), (2, 8, 1, 8, 0.3, "rank0=1,wave"),
                                                    (4, 16, 2, 6, 0.5, "rank0=3,norange"), (1, 16, 1, 7, 0.8, "flat,wave,norange"), (1, 16, 1, 9, 0.5, "two,norange"), (2, 8, 1, 8, 0.3, "rank0=1,norange"),
                                                    (4, 16, 2, 6, 0.5, "rank0=3,noebot"), (1, 16, 1, 7, 0.8, "flat,noebot"), (1, 16, 1, 9, 0.5, "noebot,norange"), (2, 8, 1, 8, 0.3, "rank0=1,noebot"),
                                                    (4, 16, 2, 6, 0.5, "rank0=3,nocompact"), (1, 16, 1, 9, 0.5, "nocompact"), (2, 8, 1, 8, 0.3, "rank0=1,nocompact,noebot")])
def test_stage2_big_bins_partial_claims_exact(K, S, E, seed, fail, sched, oracle, tmp_path, monkeypatch):
    """bins above maxsearch whose reads only partly pass the Hamming test, probed from several places of the consensus: what a probe
    sees depends on which reads the probes before it took AND on the ones nobody takes (they fill the window).  The passes of
    k_realign_big must go on while ANY lane of any probe still moves a claim -- same bytes as the sequential oracle, whatever the
    order the passes take the probes in (sched, see above)."""
    import harc_amd
    _set_sched(monkeypatch, sched)
    txt = gen.reads_text_bigbin_stage2_mixed(seed, fail_frac=fail)
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, 100, K, E, tmp_path / "o", S)
    base = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.compress(base, 100, num_thr=E, num_chains=K, num_steps=S)
    assert_same(ol.read_dir(base), s2, ol.stage2_files(E), "stage II, partly matching bins above maxsearch vs oracle")
    nleft = len(s2["input_N.dna"]) // 101
    assert 200 < nleft < 3000 - 200                              # some were taken, some never are


@pytest.mark.parametrize("L,sched", [(150, ""), (150, "rank0=2"), (250, ""), (250, "wave"), (106, "rank0=1"), (107, "")])
def test_stage2_big_bins_other_read_lengths(L, sched, oracle, tmp_path, monkeypatch):
    """the same partly matching bins above maxsearch at other read lengths: k_realign_block is compiled for windows of up to 5 / 8 / 12 words of the
    3-bit code (reads of up to 106 / 170 / 255 bases)"""
    import harc_amd
    _set_sched(monkeypatch, sched)
    txt = gen.reads_text_bigbin_stage2_mixed(11 + L, L=L, genome_len=8000, fail_frac=0.5, nsub=max(16, L // 3))
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, L, 2, 2, tmp_path / "o", 16)
    base = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.compress(base, L, num_thr=2, num_chains=2, num_steps=16)
    assert_same(ol.read_dir(base), s2, ol.stage2_files(2), "stage II, bins above maxsearch at L = %d vs oracle" % L)
    nleft = len(s2["input_N.dna"]) // (L + 1)
    assert 100 < nleft < 3000 - 100                              # some were taken, some never are


@pytest.mark.parametrize("case", CASES)
def test_decoder_matches_reference_decoder(case, tmp_path):
    """harc_amd_decoder_files == decoder.out: output.dna byte-identical to the REAL reference decoder's on the reference's streams"""
    import harc_amd
    g = ol.load_golden(case)
    base = ol.stage_dir(tmp_path, {k[len("stage2/"):]: v for k, v in g.items() if k.startswith("stage2/")})
    harc_amd.decoder(base, 1)
    assert ol.read_dir(base)["output.dna"] == g["decoded.txt"]


@pytest.mark.parametrize("slice_bytes,threads", [(0, 0), (300, 7), (65536, 1)])
@pytest.mark.parametrize("case,K,S,E", [("L100_err_5k", 16, 8, 4), ("L150_err_3k", 8, 16, 3), ("L100_bigbin2_5k", 4, 16, 8), ("L100_allN_20", 2, 4, 2)])
def test_decoder_matches_oracle_decoder_E_shards(case, K, S, E, slice_bytes, threads, oracle, tmp_path, monkeypatch):
    """(slice_bytes: output.dna leaves through verify.hip's FileDrain -- pinned slices, writer threads, a mapping of the file sized from the stream files --
    with slices of three reads and seven writers, and with one writer)"""
    import harc_amd
    if slice_bytes:
        monkeypatch.setenv("HARC_AMD_FEED_SLICE", str(slice_bytes)); monkeypatch.setenv("HARC_AMD_FEED_THREADS", str(threads))
    g = ol.load_golden(case)
    L = _L(g)
    base = ol.stage_dir(tmp_path, {k: g["stage1/" + k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
    harc_amd.compress(base, L, num_thr=E, num_chains=K, num_steps=S)
    harc_amd.decoder(base, E)
    mine = ol.read_dir(base)["output.dna"]
    os.remove(os.path.join(base, "output", "output.dna"))
    assert oracle.harc_oracle_decoder(base.encode(), E) == 0
    assert mine == ol.read_dir(base)["output.dna"]
    assert sorted(mine.split()) == sorted(g["reads.txt"].split())


def _fastq(reads, L):
    return b"".join(b"@T.%d\n%s\n+\n%s\n" % (i, r, b"H" * L) for i, r in enumerate(reads))


@pytest.mark.parametrize("case,K,S,E", [(c, 1 + (i % 3) * 7, 16 if i % 2 else 4, 1 + i % 4) for i, c in enumerate(CASES)])
def test_preserve_order_roundtrip_is_the_input_file(case, K, S, E, tmp_path):
    """-p: compress (FASTQ ingest on the GPU, stage I, stage II, pack_order) then unpack_order + decoder_preserve + merge_N in
    harc_amd_decoder_preserve_files gives back the sequence lines of the FASTQ in their original order, N reads included
    (harc:112-113, harc:183-185, merge_N.cpp:37-57)"""
    import harc_amd
    g = ol.load_golden(case)
    L = _L(g)
    reads = g["reads.txt"].split()
    fq = tmp_path / "in.fastq"
    fq.write_bytes(_fastq(reads, L))
    base = str(tmp_path)
    os.makedirs(os.path.join(base, "output"))
    harc_amd.compress_fastq(str(fq), base, L, num_thr=E, num_chains=K, num_steps=S)
    if len(ol.read_dir(base)["read_order.bin"]) == 0:
        pytest.skip("no clean reads: pack_order.cpp:36 is undefined on an empty order")
    harc_amd.pack_order(base, L)
    for f in ("input_clean.dna",):
        if os.path.exists(os.path.join(base, "output", f)):
            os.remove(os.path.join(base, "output", f))
    harc_amd.decoder(base, E, preserve_order=True)
    assert ol.read_dir(base)["output.dna"] == g["reads.txt"]


@pytest.mark.parametrize("case,E", [("L100_err_5k", 2), ("L100_repeat_dup_4k", 3), ("L100_bigbin2_5k", 1), ("L150_err_3k", 2), ("L63_err_3k", 1)])
def test_preserve_order_decoder_matches_reference_chain(case, E, tmp_path):
    """same archive through the REAL reference's unpack_order.out, decoder_preserve.out and merge_N.out (oracle/_ref): same bytes"""
    import shutil, subprocess
    import harc_amd
    g = ol.load_golden(case)
    L = _L(g)
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    dp = os.path.join(ref, "decoder_preserve_L%d_e%d.out" % (L, E))
    if not os.path.exists(dp):
        pytest.skip("oracle/_ref not built")
    reads = g["reads.txt"].split()
    fq = tmp_path / "in.fastq"
    fq.write_bytes(_fastq(reads, L))
    a, b = tmp_path / "a", tmp_path / "b"
    os.makedirs(a / "output")
    harc_amd.compress_fastq(str(fq), str(a), L, num_thr=E, num_chains=5, num_steps=16)
    harc_amd.pack_order(str(a), L)
    shutil.copytree(a, b)
    harc_amd.decoder(str(a), E, preserve_order=True)
    for exe in ("unpack_order.out", os.path.basename(dp), "merge_N.out"):
        subprocess.run([os.path.join(ref, exe), str(b)], check=True, stdout=subprocess.DEVNULL)
    assert (a / "output" / "output.dna").read_bytes() == (b / "output" / "output.dna").read_bytes() == g["reads.txt"]


@pytest.mark.parametrize("n,world,K,S", [(120000, 2, 256, 16), (160000, 4, 0, 16), (700000, 2, 512, 16)])
def test_minimizer_shard_input_matches_oracle(n, world, K, S, oracle, tmp_path):
    """what one GPU sees after the bucket exchange: only the reads of one minimizer bucket (short islands, a reseed every ~15 reads).
    Stresses k_reseed / look-ahead seeds; must be the oracle's bytes and must not depend on timing."""
    import numpy as np
    import harc_amd
    from tests.bucket_ref import pack2, bucket_ref
    arr = gen.reads_array(2024, n, 100, n * 2, err=0.0)
    keep = bucket_ref(pack2(arr), 100, world) == 0
    sel = arr[keep]
    txt = b"".join(bytes(r) + b"\n" for r in sel)
    (tmp_path / "o").mkdir(); (tmp_path / "g1").mkdir(); (tmp_path / "g2").mkdir()
    nclean = sel.shape[0]
    Ko = K if K else gen.auto_chains(nclean, clean=sel)
    inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, 100, Ko, 4, tmp_path / "o", S)
    outs = []
    for d in ("g1", "g2"):
        base = ol.stage_dir(tmp_path / d, {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna"]})
        harc_amd.reorder(base, 100, num_chains=K, num_steps=S)
        outs.append(ol.read_dir(base))
    assert outs[0] == outs[1], "two runs on the same input differ"
    assert_same(outs[0], s1, ol.STAGE1_FILES, "minimizer-shard input, stage I vs oracle")


def test_input_signature_equals_ascii_signature():
    import torch
    import harc_amd
    from tests.bucket_ref import reads_signature
    g = ol.load_golden("L150_err_3k")
    clean, withN = g["stage1/input_clean.dna"], g["stage1/input_N.dna"]
    with harc_amd.HarcAmd(harc_amd.default_params(150)) as h:
        h.set_reads_ascii(clean, len(clean) // 151, 151)
        h.set_nreads_ascii(withN, len(withN) // 151, 151)
        assert h.input_signature() == reads_signature(g["reads.txt"].split())


@pytest.mark.parametrize("case", CASES)
def test_fastq_ingest_on_gpu_matches_reference(case, tmp_path):
    """harc_amd_compress_fastq_files: FASTQ parsed and split on the GPU (preprocess.cpp + readDnaFile), then reorder + encode --
    the reference's read_order_N.bin / numreads.bin and stage-II files, byte for byte (K=1, E=1)"""
    import harc_amd
    g = ol.load_golden(case)
    L = _L(g)
    reads = g["reads.txt"].split()
    fq = tmp_path / "in.fastq"
    fq.write_bytes(b"".join(b"@T.%d some comment\n%s\n+\n%s\n" % (i, r, b"H" * L) for i, r in enumerate(reads)))
    base = ol.stage_dir(tmp_path, {})
    harc_amd.compress_fastq(str(fq), base, L, num_thr=1, num_chains=1)
    got = ol.read_dir(base)
    assert got["read_order_N.bin"] == g["stage1/read_order_N.bin"] and got["numreads.bin"] == g["stage1/numreads.bin"]
    fs = ol.stage2_files(1)
    assert_same(got, {f: g["stage2/" + f] for f in fs}, fs, f"{case}: FASTQ -> streams vs reference")


def test_fastq_file_through_the_feeder_and_the_drain_at_their_real_slice_sizes(oracle, tmp_path):
    """2.5 M reads (a 540-MB FASTQ: nine 64-MB slices of the ingest's pinned ring, sixteen reader threads, default settings): the stream files of
    harc_amd_compress_fastq_files_ex are those of the same library fed with the ORACLE's preprocess files (preprocess.cpp:81-121 restated on the CPU), byte for byte,
    and the decoder's output.dna (through the drain: 250 MB, four slices, sixteen writers into a preallocated mapping) holds exactly the reads written"""
    import numpy as np
    import harc_amd
    n, L, E = 2_500_000, 100, 4
    arr = gen.reads_array_big(77, n, L, 10_000_000, err=0.004)
    rec = np.empty((n, 2 * L + 17), dtype=np.uint8)
    rec[:, 0:3] = np.frombuffer(b"@T.", dtype=np.uint8)
    idx = np.arange(n)
    for k in range(9):
        rec[:, 3 + k] = (idx // 10 ** (8 - k)) % 10 + 48
    rec[:, 12] = 10; rec[:, 13:13 + L] = arr; rec[:, 13 + L] = 10; rec[:, 14 + L] = ord("+"); rec[:, 15 + L] = 10
    rec[:, 16 + L:16 + 2 * L] = ord("H"); rec[:, 16 + 2 * L] = 10
    root = "/dev/shm" if os.path.isdir("/dev/shm") else str(tmp_path)
    import shutil, tempfile
    d = tempfile.mkdtemp(dir=root, prefix="harc_test_e2e_")
    try:
        fq = os.path.join(d, "x.fastq")
        rec.tofile(fq)
        del rec
        a = os.path.join(d, "a"); b = os.path.join(d, "b")
        os.makedirs(os.path.join(a, "output")); os.makedirs(os.path.join(b, "output"))
        harc_amd.compress_fastq(fq, a, L, num_thr=E, num_chains=0)
        lines = gen.lines_of(arr)
        assert oracle.harc_oracle_preprocess(lines, len(lines), L, b.encode()) == 0
        harc_amd.compress(b, L, num_thr=E, num_chains=0)
        ga, gb = ol.read_dir(a), ol.read_dir(b)
        fs = ol.stage2_files(E) + ["numreads.bin", "read_order_N.bin"]
        assert_same(ga, gb, fs, "FASTQ file -> streams vs the oracle's preprocess files -> streams")
        harc_amd.decoder(a, E)
        out = np.fromfile(os.path.join(a, "output", "output.dna"), dtype=np.uint8).reshape(-1, L + 1)
        assert out.shape[0] == n and bool((out[:, L] == 10).all())
        v = np.dtype((np.void, L))
        assert np.array_equal(np.sort(np.ascontiguousarray(out[:, :L]).view(v).ravel()), np.sort(np.ascontiguousarray(arr).view(v).ravel()))
    finally:
        shutil.rmtree(d, ignore_errors=True)


def test_fastq_ingest_edge_cases(tmp_path):
    import harc_amd
    base = ol.stage_dir(tmp_path, {})
    bad = tmp_path / "bad.fastq"
    bad.write_bytes(b"@a\nACGTACGTAC\n+\nHHHHHHHHHH\n@b\nACGTACG\n+\nHHHHHHH\n")
    with pytest.raises(harc_amd.HarcAmdError):
        harc_amd.compress_fastq(str(bad), base, 10)                      # preprocess.cpp:92-97
    ok = tmp_path / "ok.fastq"
    ok.write_bytes(b"@a\r\nACGTACGTAC\r\n+\r\nHHHHHHHHHH\r\n@b\nACNTACGTAC\n+\nHHHHHHHHHH")   # CRLF record, no final newline
    harc_amd.compress_fastq(str(ok), base, 10)
    got = ol.read_dir(base)
    assert got["numreads.bin"] == (1).to_bytes(4, "little") and got["read_order_N.bin"] == (1).to_bytes(4, "little")
    assert got["input_N.dna"] == b"ACNTACGTAC\n"


@pytest.mark.parametrize("tail", [b"", b"@cut\n", b"@cut\nSEQ", b"@cut\nSEQ\n", b"@cut\nSEQN\n+\n", b"@cut\nACGT\n", b"\n", b"\n\n"])
def test_fastq_ingest_truncated_last_record_like_the_reference(tail, tmp_path):
    """a FASTQ that ends inside a record: the reference's getline loop still takes the read if its sequence line is complete
    (preprocess.cpp:90-111) and fails on a short one (:92-97).  GPU ingest == host twin == preprocess.out (oracle/_ref), files and verdict"""
    import subprocess
    import harc_amd
    L = 100
    reads = gen.reads_text(5, 3000, L, 20000, err=0.01).split()
    seq = reads[7][:60] + b"ACGT" * 10
    t = tail.replace(b"SEQN", seq[:40] + b"N" + seq[41:]).replace(b"SEQ", seq)
    fq = tmp_path / "in.fastq"
    fq.write_bytes(_fastq(reads, L) + t)
    outs = []
    for mode in ("gpu", "host", "ref"):
        d = tmp_path / mode
        os.makedirs(d / "output")
        ok = True
        try:
            if mode == "gpu":
                harc_amd.compress_fastq(str(fq), str(d), L, num_thr=1, num_chains=1)
            elif mode == "host":
                harc_amd.preprocess(str(fq), str(d), L)
            else:
                exe = os.path.join(_REF, "preprocess.out")
                if not os.path.exists(exe):
                    continue
                ok = subprocess.run([exe, str(fq), str(d), "False", "False", str(L)], stdout=subprocess.DEVNULL).returncode == 0
        except harc_amd.HarcAmdError:
            ok = False
        got = ol.read_dir(str(d)) if ok else {}
        outs.append((mode, ok, got.get("numreads.bin"), got.get("read_order_N.bin")))
    assert len({o[1:] for o in outs}) == 1, outs
    if outs[0][1] and b"SEQ\n" in tail:
        assert int.from_bytes(outs[0][2], "little") == sum(b"N" not in r for r in reads) + 1


# ------------------------------------------------------------------------------------------------ -q
def _q_stream_env(monkeypatch, stream):
    """-q without -p as on a FASTQ file larger than HBM: ingested in pieces of a few thousand records, quality values and ids permuted by
    streaming the file again once per bin of ~4000 output lines (ingest.hip emit_quality_and_ids_streamed)"""
    if stream:
        monkeypatch.setenv("HARC_AMD_Q_STREAM", "1"); monkeypatch.setenv("HARC_AMD_INGEST_CHUNK", "700000"); monkeypatch.setenv("HARC_AMD_Q_BIN", "400000")
    else:
        monkeypatch.setenv("HARC_AMD_Q_STREAM", "0")


@pytest.mark.parametrize("stream", [False, True])
@pytest.mark.parametrize("case", ol.quality_cases())
def test_quality_ids_match_reference_golden(case, stream, tmp_path, monkeypatch):
    """K=1, E=1 gives the reference's own orders, so output.quality / output.id must be the REAL reference's bytes (reorder_quality.out
    after preprocess / reorder / encoder at num_thr=1), and with -p the files preprocess.out writes directly"""
    import json
    import harc_amd
    _q_stream_env(monkeypatch, stream)
    g = ol.load_golden(case)
    L = json.loads(g["meta.json"])["L"]
    for mode, po in (("np", False), ("p", True)):
        d = tmp_path / mode
        os.makedirs(d / "output")
        (d / "in.fastq").write_bytes(g["in.fastq"])
        harc_amd.compress_fastq(str(d / "in.fastq"), str(d), L, num_thr=1, num_chains=1, num_steps=16, preserve_order=po, preserve_quality=True)
        got = ol.read_dir(str(d))
        if not po:
            assert got["read_order.bin"] == g["np/read_order.bin"] and got["read_order_N_pe.bin"] == g["np/read_order_N_pe.bin"]
        assert got["output.quality"] == g[mode + "/output.quality"], mode
        assert got["output.id"] == g[mode + "/output.id"], mode


@pytest.mark.parametrize("stream", [False, True])
@pytest.mark.parametrize("K,S,E,trunc", [(7, 16, 3, 0), (64, 8, 2, 1), (3, 4, 8, 0)])
def test_quality_ids_match_oracle_any_schedule(K, S, E, trunc, stream, oracle, tmp_path, monkeypatch):
    """other (K, S, E): the gather must follow this run's own orders -- checked against the oracle's restatement fed with the
    GPU's order files; trunc: the FASTQ ends in a partial record (1 = dangling id line, 2 = id + sequence, no newline at the end)"""
    import harc_amd
    _q_stream_env(monkeypatch, stream)
    L = 100
    reads = gen.reads_text(4242, 30000, L, 200000, err=0.006).split()
    import numpy as np
    rs = np.random.RandomState(5)
    recs = []
    for i, r in enumerate(reads):
        q = bytes(35 if c == 78 else 40 + int(x) for c, x in zip(r, rs.randint(0, 30, L)))
        recs.append(b"@id.%d %s\n%s\n+\n%s\n" % (i, b"y" * int(rs.randint(0, 12)), r, q))
    fq = b"".join(recs)
    if trunc == 1:
        fq += b"@dangling id\n"
    elif trunc == 2:
        fq += b"@dangling id\n" + reads[0][:50]
    (tmp_path / "in.fastq").write_bytes(fq)
    os.makedirs(tmp_path / "output")
    harc_amd.compress_fastq(str(tmp_path / "in.fastq"), str(tmp_path), L, num_thr=E, num_chains=K, num_steps=S, preserve_quality=True)
    got = ol.read_dir(str(tmp_path))
    o = tmp_path / "o"
    ol.stage_dir(o, {f: got[f] for f in ("read_order.bin", "read_order_N_pe.bin")})
    assert oracle.harc_oracle_quality(fq, len(fq), L, 0, str(o).encode()) == 0
    exp = ol.read_dir(str(o))
    assert got["output.quality"] == exp["output.quality"]
    assert got["output.id"] == exp["output.id"]
    # and it is a permutation of the input's quality lines that lines up with the decoded reads
    harc_amd.decoder(str(tmp_path), E)
    dec = ol.read_dir(str(tmp_path))["output.dna"].split()
    ql = got["output.quality"].split(b"\n")[:-1]
    assert len(dec) == len(ql) == len(reads)
    da = np.frombuffer(b"".join(dec), dtype=np.uint8).reshape(-1, L)
    qa = np.frombuffer(b"".join(ql), dtype=np.uint8).reshape(-1, L)
    assert ((da == ord("N")) == (qa == ord("#"))).all()            # quality line p belongs to decoded read p
    assert sorted(ql) == sorted(r.split(b"\n")[3] for r in recs)


@pytest.mark.parametrize("stream", [False, True])
def test_quality_wrong_length_rejected(stream, tmp_path, monkeypatch):
    import harc_amd
    _q_stream_env(monkeypatch, stream)
    reads = gen.reads_text(1, 2000, 100, 20000).split()
    fq = b"".join(b"@r%d\n%s\n+\n%s\n" % (i, r, b"H" * (99 if i == 700 else 100)) for i, r in enumerate(reads))
    (tmp_path / "in.fastq").write_bytes(fq)
    os.makedirs(tmp_path / "output")
    with pytest.raises(harc_amd.HarcAmdError):
        harc_amd.compress_fastq(str(tmp_path / "in.fastq"), str(tmp_path), 100, preserve_quality=True)


# ------------------------------------------------------------------------------------------------ against the REAL reference's programs
# oracle/_ref/*.out are the reference's own sources compiled by oracle/build_ref.sh (they travel to the GPU box as binaries).
_REF = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")


def _ref_exe(name):
    p = os.path.join(_REF, name)
    if not os.path.exists(p):
        pytest.skip("oracle/_ref/%s not built" % name)
    return p


def _run_ref(exe, *args, cwd):
    import subprocess
    subprocess.run([exe] + [str(a) for a in args], cwd=str(cwd), check=True, stdout=subprocess.DEVNULL, stderr=subprocess.STDOUT)


@pytest.mark.parametrize("K,S", [(64, 16), (500, 8)])
def test_P4_reference_encoder_on_gpu_stage1_files(K, S, oracle, tmp_path):
    """the 7 stage-I files of a K-chain GPU run, fed to the REAL encoder.out (-t 1), give byte-identical stage-II
    streams to the GPU encoder at E=1"""
    import shutil
    import harc_amd
    L = 100
    enc = _ref_exe("encoder_L100_t1.out")
    txt = gen.reads_text(31, 60000, L, 400000, err=0.008)
    a, b = tmp_path / "a", tmp_path / "b"
    ol.stage_dir(a, {})
    assert oracle.harc_oracle_preprocess(txt, len(txt), L, str(a).encode()) == 0
    harc_amd.reorder(str(a), L, num_thr=1, num_chains=K, num_steps=S)
    shutil.copytree(a, b)
    harc_amd.encoder(str(a), L, num_thr=1)
    _run_ref(enc, b, cwd=b)
    ga, gb = ol.read_dir(str(a)), ol.read_dir(str(b))
    for f in ol.stage2_files(1):
        assert ga[f] == gb[f], f


@pytest.mark.parametrize("K,E", [(0, 8), (33, 3)])
def test_P3_reference_decoder_reads_our_archives(K, E, tmp_path):
    """streams written by this build (any K, E) are decoded by the REAL decoder.out to the same bytes as by
    harc_amd_decoder_files, and to the input multiset"""
    import shutil
    import harc_amd
    L = 100
    dec = _ref_exe("decoder.out")
    reads = gen.reads_text(32, 80000, L, 500000, err=0.01).split()
    fq = tmp_path / "in.fastq"
    fq.write_bytes(_fastq(reads, L))
    a, b = tmp_path / "a", tmp_path / "b"
    os.makedirs(a / "output")
    harc_amd.compress_fastq(str(fq), str(a), L, num_thr=E, num_chains=K)
    shutil.copytree(a, b)
    harc_amd.decoder(str(a), E)
    _run_ref(dec, b, 2, E, cwd=b)
    mine, ref = (a / "output" / "output.dna").read_bytes(), (b / "output" / "output.dna").read_bytes()
    assert mine == ref
    assert sorted(mine.split()) == sorted(reads)


def test_P5_stream_sizes_vs_reference_t8(tmp_path):
    """at the default schedule the streams are not larger than the reference's own multi-threaded run by more than
    the stated margin (xz -6 of every stage-II stream as the stand-in for bsc / 7z): GPU default <= 1.05 x reference -t 8"""
    import lzma
    import shutil
    import harc_amd
    L = 100
    reo, enc, pre = _ref_exe("reorder_L100_t8.out"), _ref_exe("encoder_L100_t8.out"), _ref_exe("preprocess.out")
    reads = gen.reads_text(33, 400000, L, 1500000, err=0.008).split()
    fq = tmp_path / "in.fastq"
    fq.write_bytes(_fastq(reads, L))
    a, b = tmp_path / "a", tmp_path / "b"
    os.makedirs(a / "output"); os.makedirs(b / "output")
    harc_amd.compress_fastq(str(fq), str(a), L, num_thr=8, num_chains=0)
    _run_ref(pre, fq, b, "False", "False", L, cwd=b)
    _run_ref(reo, b, cwd=b)
    _run_ref(enc, b, cwd=b)

    def size(d):
        tot = 0
        for f, v in ol.read_dir(str(d)).items():
            if f.startswith(("read_seq", "read_pos", "read_noise", "read_noisepos", "read_rev", "read_singleton", "input_N", "read_meta")):
                tot += len(lzma.compress(v, preset=6))
        return tot
    sa, sb = size(a), size(b)
    print("stream sizes xz-6: gpu default", sa, "reference -t 8", sb, "ratio %.3f" % (sa / sb))
    assert sa <= 1.05 * sb


def test_preserve_order_decoder_needs_order_files(tmp_path):
    """-d -p on streams that were not kept with their order files fails loudly (harc:146-153), it does not guess"""
    import harc_amd
    g = ol.load_golden("L100_err_5k")
    base = ol.stage_dir(tmp_path, {k[len("stage2/"):]: v for k, v in g.items() if k.startswith("stage2/") and "read_order" not in k})
    with pytest.raises(harc_amd.HarcAmdError) as e:
        harc_amd.decoder(base, 1, preserve_order=True)
    assert "order" in str(e.value)
    # a packed order that does not match the streams is refused too
    base2 = ol.stage_dir(tmp_path / "b", {k[len("stage2/"):]: v for k, v in g.items() if k.startswith("stage2/")})
    for k, v in g.items():
        if k.startswith("packed/"):
            (tmp_path / "b" / "output" / k[len("packed/"):]).write_bytes(v)
    bad = bytearray(g["packed/read_order.bin"]); bad[4:8] = (int.from_bytes(bad[4:8], "little") - 5).to_bytes(4, "little")
    (tmp_path / "b" / "output" / "read_order.bin").write_bytes(bytes(bad))
    with pytest.raises(harc_amd.HarcAmdError):
        harc_amd.decoder(base2, 1, preserve_order=True)


def test_partition_kernel_is_a_stable_sort_by_bucket():
    """harc_amd_partition_reads_device == torch.sort(stable) of the k_bucket values + gather: what BucketSharder sends"""
    import numpy as np
    import torch
    import harc_amd
    from tests.bucket_ref import pack2, bucket_ref
    reads = gen.reads_array(77, 50000, 100, 300000, err=0.002)
    reads = reads[~(reads == ord("N")).any(1)]
    pk = pack2(reads)
    dev = torch.device("cuda", 0)
    packed = torch.from_numpy(pk.view(np.int64)).to(dev)
    h = harc_amd.HarcAmd(harc_amd.default_params(100))
    for world in (1, 2, 5, 8):
        send = torch.empty_like(packed)
        counts = torch.zeros((world,), dtype=torch.int64, device=dev)
        torch.cuda.synchronize()
        h.partition_reads_device(packed.data_ptr(), packed.shape[0], world, send.data_ptr(), counts.data_ptr())
        b = bucket_ref(pk, 100, world)
        order = np.argsort(b, kind="stable")
        assert (send.cpu().numpy() == pk[order]).all()             # both int64 views of the packed words
        assert (counts.cpu().numpy() == np.bincount(b, minlength=world)).all()
    h.close()


_FUZZ_L = [20, 31, 32, 33, 47, 50, 51, 64, 65, 95, 96, 97, 127, 128, 129, 159, 160, 161, 191, 192, 193, 223, 224, 225, 254]


@pytest.mark.parametrize("L", _FUZZ_L)
def test_read_lengths_at_word_boundaries_match_oracle(L, oracle, tmp_path):
    """every packed-word boundary of the 2-bit (W = ceil(2L/64)) and 3-bit (W3 = ceil(3L/64)) stores, with a schedule drawn from L:
    all stage-I / stage-II files equal the oracle's, the decoder gives the input multiset back, and -p its order"""
    import numpy as np
    import harc_amd
    rs = np.random.RandomState(L)
    n = int(rs.randint(1500, 4000))
    K, S, E = int(rs.choice([1, 2, 7, 33, 200])), int(rs.choice([1, 3, 16, 64])), int(rs.choice([1, 2, 5, 8]))
    txt = gen.reads_text(1000 + L, n, L, int(n * L / rs.choice([4, 15, 40])), err=float(rs.choice([0.0, 0.005, 0.02])))
    (tmp_path / "o").mkdir(); (tmp_path / "g").mkdir()
    inputs, s1, s2, _ = _oracle_pipeline(oracle, txt, L, K, E, tmp_path / "o", S)
    base = ol.stage_dir(tmp_path / "g", {k: inputs[k] for k in ["input_clean.dna", "numreads.bin", "input_N.dna", "read_order_N.bin"]})
    harc_amd.reorder(base, L, num_chains=K, num_steps=S)
    assert_same(ol.read_dir(base), s1, ol.STAGE1_FILES, f"L={L} K={K} S={S}: stage I vs oracle")
    harc_amd.encoder(base, L, num_thr=E)
    assert_same(ol.read_dir(base), s2, ol.stage2_files(E), f"L={L} K={K} E={E}: stage II vs oracle")
    if len(ol.read_dir(base)["read_order.bin"]):
        harc_amd.pack_order(base, L)
        harc_amd.decoder(base, E, preserve_order=True)
        assert ol.read_dir(base)["output.dna"] == txt


def test_reads_per_chain_is_the_auto_chain_count(tmp_path):
    """harc_amd_params.reads_per_chain only chooses K in auto mode: same bytes as the explicit num_chains = N / reads_per_chain"""
    import harc_amd
    txt = gen.reads_text(91, 40000, 100, 250000, err=0.0)
    n = len(txt.split())
    outs = []
    for kw in (dict(num_chains=0, reads_per_chain=64), dict(num_chains=n // 64), dict(num_chains=0), dict(num_chains=gen.auto_chains(n))):
        h = harc_amd.HarcAmd(harc_amd.default_params(100, num_thr=2, **kw))
        h.set_reads_ascii(txt, n, 101)
        h.set_nreads_ascii(b"", 0, 101)
        h.reorder(); h.encode()
        outs.append((h.counters().chains, h.stream("S1_ORDER"), h.stream("S2_SEQ", 0), h.stream("S2_NOISE", 1)))
        h.close()
    assert outs[0] == outs[1] and outs[0][0] == n // 64
    assert outs[2] == outs[3] and outs[2][0] == gen.auto_chains(n) == n // 1024


def test_a_failed_run_leaves_the_context_usable(monkeypatch, tmp_path):
    """error-path hygiene: an allocation that fails in the middle of harc_amd_reorder (HARC_AMD_FAIL_ALLOC refuses the n-th pool
    allocation) is reported as HARC_AMD_ENOMEM, and the same context then gives the bytes of a fresh one"""
    import subprocess, sys, textwrap
    script = tmp_path / "w.py"
    script.write_text(textwrap.dedent("""
        import os, sys
        sys.path.insert(0, %r)
        import harc_amd
        from tests import gen
        txt = gen.reads_text(5, 30000, 100, 200000, err=0.005)
        clean = b"".join(l + b"\\n" for l in txt.split() if b"N" not in l); nn = b"".join(l + b"\\n" for l in txt.split() if b"N" in l)
        def run(h):
            h.reorder(); h.encode()
            return [h.stream("S2_SEQ", e) + h.stream("S2_NOISE", e) for e in range(2)] + [h.stream("S2_ORDER")]
        h = harc_amd.HarcAmd(harc_amd.default_params(100, num_thr=2, num_chains=16))
        h.set_reads_ascii(clean, len(clean) // 101, 101); h.set_nreads_ascii(nn, len(nn) // 101, 101)
        try:
            run(h)
            print("NOFAIL")
        except harc_amd.HarcAmdError as e:
            print("FAILED", e.code)
        a = run(h)                                            # the countdown is spent: this run goes through, on the same context
        g = harc_amd.HarcAmd(harc_amd.default_params(100, num_thr=2, num_chains=16))
        g.set_reads_ascii(clean, len(clean) // 101, 101); g.set_nreads_ascii(nn, len(nn) // 101, 101)
        print("SAME" if run(g) == a else "DIFFERENT")
    """ % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    for n in (12, 40):
        out = subprocess.run([sys.executable, str(script)], env=dict(os.environ, HARC_AMD_FAIL_ALLOC=str(n)), stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert out.returncode == 0 and "FAILED -5" in out.stdout and "SAME" in out.stdout, out.stdout[-2000:]


@pytest.mark.parametrize("case,E,bins", [("L100_err_5k", 2, 777), ("L100_repfam_5k", 3, 64), ("L150_err_3k", 2, 1000), ("L100_allN_20", 1, 3), ("L100_gen_noRC_e_3k", 1, 1)])
def test_preserve_order_decoder_in_small_bins(case, E, bins, tmp_path, monkeypatch):
    """-d -p with the order restored in bins of a few output lines (the reference's -m / MAX_BIN_SIZE path, decoder_preserve.cpp:246-290;
    HARC_AMD_BIN_READS forces the bin size): every bin decodes the streams again and keeps its own lines.  Same bytes as the input file
    and as the REAL reference's unpack_order + decoder_preserve + merge_N on the same archive."""
    import shutil, subprocess
    import harc_amd
    g = ol.load_golden(case)
    L = _L(g)
    reads = g["reads.txt"].split()
    fq = tmp_path / "in.fastq"
    fq.write_bytes(_fastq(reads, L))
    a, b = tmp_path / "a", tmp_path / "b"
    os.makedirs(a / "output")
    harc_amd.compress_fastq(str(fq), str(a), L, num_thr=E, num_chains=3, num_steps=16)
    if len(ol.read_dir(str(a))["read_order.bin"]) == 0:
        (a / "output" / "read_order.bin").write_bytes(b"")      # all reads have N: pack_order.cpp:36 is undefined on an empty order
        (a / "output" / "read_order.bin.tail").write_bytes(b"")
    else:
        harc_amd.pack_order(str(a), L)
    shutil.copytree(a, b)
    monkeypatch.setenv("HARC_AMD_BIN_READS", str(bins))
    harc_amd.decoder(str(a), E, preserve_order=True, memory_gb=1)
    assert (a / "output" / "output.dna").read_bytes() == g["reads.txt"]
    ref = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_ref")
    dp = os.path.join(ref, "decoder_preserve_L%d_e%d.out" % (L, E))
    if os.path.exists(dp) and len(ol.read_dir(str(b))["read_order.bin"]) > 0:
        for exe in ("unpack_order.out", os.path.basename(dp), "merge_N.out"):
            subprocess.run([os.path.join(ref, exe), str(b)], check=True, stdout=subprocess.DEVNULL)
        assert (b / "output" / "output.dna").read_bytes() == g["reads.txt"]


@pytest.mark.parametrize("case,piece,slice_bytes,threads", [("L100_err_5k", 3000, 0, 0), ("L150_err_3k", 700, 0, 0), ("L100_repfam_5k", 100000, 0, 0), ("L100_three", 16, 0, 0),
                                                            # the file feeder (ingest.hip FileFeeder: reader threads -> ring of pinned slices -> uploads): pieces of many
                                                            # slices, more pieces than the ring has slices (hundreds of one-slice pieces read ahead), one reader, many readers
                                                            ("L100_repfam_5k", 100000, 4096, 3), ("L100_err_5k", 3000, 256, 8), ("L100_err_5k", 3000, 1 << 20, 1), ("L150_err_3k", 100000000, 777, 5)])
def test_fastq_ingest_in_pieces_is_the_same(case, piece, slice_bytes, threads, tmp_path, monkeypatch):
    """a FASTQ larger than HBM is ingested a piece of whole records at a time (HARC_AMD_INGEST_CHUNK forces pieces of a few records here;
    quality lines that begin with '@' sit at the cuts): same files as the one-piece ingest, -q -p files in file order, truncated last
    record handled as the reference's getline loop does (preprocess.cpp:90-111)"""
    import numpy as np
    import harc_amd
    g = ol.load_golden(case)
    L = _L(g)
    reads = g["reads.txt"].split()
    rs = np.random.RandomState(4)
    quals = [bytes([64] + [50 + int(x) for x in rs.randint(0, 20, L - 1)]) for _ in reads]
    ids = [b"@r%d %s" % (i, b"x" * int(rs.randint(0, 30))) for i in range(len(reads))]
    body = b"".join(b"%s\n%s\n+\n%s\n" % t for t in zip(ids, reads, quals))
    body += b"@last\n" + reads[0]                                  # a truncated last record: its read counts, it has no quality line
    fq = tmp_path / "in.fastq"
    fq.write_bytes(body)
    outs = []
    for d, env in (("whole", None), ("pieces", str(piece))):
        base = tmp_path / d
        os.makedirs(base / "output")
        if env:
            monkeypatch.setenv("HARC_AMD_INGEST_CHUNK", env)
            if slice_bytes:
                monkeypatch.setenv("HARC_AMD_FEED_SLICE", str(slice_bytes)); monkeypatch.setenv("HARC_AMD_FEED_THREADS", str(threads))
        harc_amd.compress_fastq(str(fq), str(base), L, num_thr=2, num_chains=4, num_steps=16, preserve_order=True, preserve_quality=True)
        outs.append(ol.read_dir(str(base)))
    assert outs[0].keys() == outs[1].keys()
    for k in outs[0]:
        assert outs[0][k] == outs[1][k], k
    assert outs[1]["output.quality"].split(b"\n")[:-1] == quals and outs[1]["output.id"].split(b"\n")[:-1] == ids + [b"@last"]
    assert np.frombuffer(outs[1]["numreads.bin"], dtype=np.uint32)[0] == sum(1 for r in reads + [reads[0]] if b"N" not in r)
