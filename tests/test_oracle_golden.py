"""P0 of the parity ladder: the CPU oracle at K=1, E=1 is byte-identical to the REAL reference
(golden vectors made by oracle/make_goldens.py from /root/reference at num_thr=1)."""
import ctypes as C
import os

import pytest

from tests import oracle_lib as ol

CASES = ol.golden_cases()


@pytest.mark.parametrize("case", CASES)
def test_preprocess_matches_reference(case, oracle, tmp_path):
    g = ol.load_golden(case)
    L = len(g["reads.txt"].split(b"\n")[0])
    base = ol.stage_dir(tmp_path, {})
    assert oracle.harc_oracle_preprocess(g["reads.txt"], len(g["reads.txt"]), L, base.encode()) == 0
    got = ol.read_dir(base)
    for f in ["input_clean.dna", "input_N.dna", "numreads.bin", "read_order_N.bin"]:
        assert got[f] == g["stage1/" + f], f


@pytest.mark.parametrize("S", [1, 16])
@pytest.mark.parametrize("case", CASES)
def test_stage1_K1_bit_exact(case, S, oracle, tmp_path):
    g = ol.load_golden(case)
    L = len(g["reads.txt"].split(b"\n")[0])
    base = ol.stage_dir(tmp_path, {k: g["stage1/" + k] for k in ["input_clean.dna", "numreads.bin"]})
    unmatched = C.c_uint32(0)
    assert oracle.harc_oracle_reorder(base.encode(), L, 1, S, C.byref(unmatched), None) == 0
    got = ol.read_dir(base)
    for f in ol.STAGE1_FILES:
        assert got[f] == g["stage1/" + f], f"{case}: {f} differs from reference"
    import json
    log = json.loads(g["meta.json"])["log"]
    ref_unmatched = [int(l.split()[2]) for l in log if l.startswith("Reordering done")][0]
    assert unmatched.value == ref_unmatched


@pytest.mark.parametrize("case", CASES)
def test_stage2_E1_bit_exact(case, oracle, tmp_path):
    g = ol.load_golden(case)
    L = len(g["reads.txt"].split(b"\n")[0])
    base = ol.stage_dir(tmp_path, {k[len("stage1/"):]: v for k, v in g.items() if k.startswith("stage1/")})
    ms, mn = C.c_uint32(0), C.c_uint32(0)
    assert oracle.harc_oracle_encoder(base.encode(), L, 1, C.byref(ms), C.byref(mn)) == 0
    got = ol.read_dir(base)
    for f in ol.stage2_files(1):
        assert got[f] == g["stage2/" + f], f"{case}: {f} differs from reference"
    import json
    log = json.loads(g["meta.json"])["log"]
    assert ms.value == [int(l.split()[0]) for l in log if "singleton reads were aligned" in l][0]
    assert mn.value == [int(l.split()[0]) for l in log if "reads with N were aligned" in l][0]


@pytest.mark.parametrize("case", CASES)
def test_pack_order_and_decoder(case, oracle, tmp_path):
    g = ol.load_golden(case)
    base = ol.stage_dir(tmp_path, {k[len("stage2/"):]: v for k, v in g.items() if k.startswith("stage2/")})
    assert oracle.harc_oracle_decoder(base.encode(), 1) == 0
    assert ol.read_dir(base)["output.dna"] == g["decoded.txt"]
    if "packed/read_order.bin" in g:
        assert oracle.harc_oracle_pack_order(base.encode()) == 0
        got = ol.read_dir(base)
        assert got["read_order.bin"] == g["packed/read_order.bin"]
        assert got["read_order.bin.tail"] == g["packed/read_order.bin.tail"]
    else:
        assert oracle.harc_oracle_pack_order(base.encode()) == -3


@pytest.mark.parametrize("case", ol.quality_cases())
def test_quality_and_ids_match_reference(case, oracle, tmp_path):
    """-q: oracle restatement of preprocess.cpp:61-118 + reorder_quality.cpp == the reference's output.quality / output.id,
    with -p (file order) and without (gathered by the post-encoding orders, id routed by the previous record's N flag)"""
    import json
    g = ol.load_golden(case)
    L = json.loads(g["meta.json"])["L"]
    fq = g["in.fastq"]
    base = ol.stage_dir(tmp_path / "p", {})
    assert oracle.harc_oracle_quality(fq, len(fq), L, 1, base.encode()) == 0
    got = ol.read_dir(base)
    assert got["output.quality"] == g["p/output.quality"] and got["output.id"] == g["p/output.id"]
    base = ol.stage_dir(tmp_path / "np", {f: g["np/" + f] for f in ("read_order.bin", "read_order_N_pe.bin")})
    assert oracle.harc_oracle_quality(fq, len(fq), L, 0, base.encode()) == 0
    got = ol.read_dir(base)
    assert got["output.quality"] == g["np/output.quality"], "quality"
    assert got["output.id"] == g["np/output.id"], "id"


def test_quality_full_pipeline_orders_match_reference(oracle, tmp_path):
    """the order files inside the -q fixture are what the oracle pipeline (K=1, E=1) produces from the same FASTQ"""
    import json
    case = ol.quality_cases()[0]
    g = ol.load_golden(case)
    L = json.loads(g["meta.json"])["L"]
    lines = g["in.fastq"].split(b"\n")
    reads = b"".join(l + b"\n" for l in lines[1::4])
    base = ol.stage_dir(tmp_path, {})
    assert oracle.harc_oracle_preprocess(reads, len(reads), L, base.encode()) == 0
    assert oracle.harc_oracle_reorder(base.encode(), L, 1, 16, None, None) == 0
    assert oracle.harc_oracle_encoder(base.encode(), L, 1, None, None) == 0
    got = ol.read_dir(base)
    assert got["read_order.bin"] == g["np/read_order.bin"] and got["read_order_N_pe.bin"] == g["np/read_order_N_pe.bin"]


@pytest.mark.parametrize("case", ol.md5_cases())
def test_oracle_K1_at_config_size(case, oracle, tmp_path):
    """P0 at CONFIG size: the oracle at K = 1, E = 1 against the md5 of every file the reference leaves at -t 1 on BASELINE.json configs[0]
    (1 M x 100 bp, 2.9x) and the configs[1] stand-in (3.3 M x 100 bp, 52x, 0.5 % errors, 12 % N reads) -- 450 000 reseeds on the first,
    410 000 realigned singleton / N reads on the second (tests/golden/md5_*.json, made by oracle/make_goldens.py)"""
    from tests import gen
    meta, arr = ol.load_md5_case(case)
    L = arr.shape[1]
    txt = gen.lines_of(arr)
    del arr
    shm = "/dev/shm" if os.path.isdir("/dev/shm") else None
    import tempfile
    with tempfile.TemporaryDirectory(dir=shm, prefix="harc_md5_") as td:
        base = ol.stage_dir(td, {})
        assert oracle.harc_oracle_preprocess(txt, len(txt), L, base.encode()) == 0
        del txt
        um = C.c_uint32(0)
        assert oracle.harc_oracle_reorder(base.encode(), L, 1, 1, C.byref(um), None) == 0
        bad = ol.md5_mismatches(ol.read_dir(base), meta["stage1"], ol.STAGE1_FILES + ["input_clean.dna", "input_N.dna", "numreads.bin", "read_order_N.bin"])
        assert not bad, f"{case}: stage I differs from the reference: {bad}"
        assert um.value == [int(l.split()[2]) for l in meta["log"] if l.startswith("Reordering done")][0]
        ms, mn = C.c_uint32(0), C.c_uint32(0)
        assert oracle.harc_oracle_encoder(base.encode(), L, 1, C.byref(ms), C.byref(mn)) == 0
        bad = ol.md5_mismatches(ol.read_dir(base), meta["stage2"], ol.stage2_files(1))
        assert not bad, f"{case}: stage II differs from the reference: {bad}"
        assert ms.value == [int(l.split()[0]) for l in meta["log"] if "singleton reads were aligned" in l][0]
        assert mn.value == [int(l.split()[0]) for l in meta["log"] if "reads with N were aligned" in l][0]
