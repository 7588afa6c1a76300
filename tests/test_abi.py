"""CPU-side checks of the drop-in boundary: libharc_amd.so loads, exports every symbol include/harc_amd.h declares,
fills the reference's parameter formulae (harc:52-60), and refuses to compute without a device (no CPU fallback)."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    import harc_amd
    hdr = open(os.path.join(ROOT, "include", "harc_amd.h")).read()
    declared = sorted(set(re.findall(r"\b(harc_amd_[a-z0-9_]+)\s*\(", hdr)))
    assert len(declared) >= 18
    l = harc_amd.lib()
    for name in declared:
        assert hasattr(l, name), f"{name} declared in include/harc_amd.h but not exported"


@pytest.mark.parametrize("L,d1,d2", [(100, (18, 49), (50, 81)), (150, (43, 74), (75, 106)), (40, (8, 19), (20, 31)), (101, (18, 49), (50, 81)),
                                      (63, (11, 30), (31, 50)), (255, (95, 126), (127, 158))])
def test_default_params_follow_harc_driver(L, d1, d2):
    import harc_amd
    p = harc_amd.default_params(L)
    assert (p.dict_start[0], p.dict_end[0]) == d1 and (p.dict_start[1], p.dict_end[1]) == d2
    assert p.maxmatch == L // 2 and p.thresh == 4 and p.thresh_s == 24 and p.maxsearch == 1000 and p.num_thr == 8
    assert p.table_slots_per_read == 0 and harc_amd.default_params(L, table_slots_per_read=2).table_slots_per_read == 2      # 0 = the library chooses by free memory


def test_readlen_limits():
    import harc_amd
    for bad in (0, 256, 1000, -3):
        with pytest.raises(harc_amd.HarcAmdError):
            harc_amd.default_params(bad)


def test_no_cpu_fallback():
    import torch
    import harc_amd
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(harc_amd.HarcAmdError) as e:
        harc_amd.HarcAmd(harc_amd.default_params(100))
    assert e.value.code == -2


@pytest.mark.parametrize("case", ["L100_err_5k", "L150_err_3k", "L100_allN_20", "L100_one"])
def test_preprocess_matches_reference_files(case, tmp_path):
    """harc_amd_preprocess_files is host code (no device needed): same four files as the reference's preprocess.out"""
    import harc_amd
    from tests import oracle_lib as ol
    g = ol.load_golden(case)
    reads = g["reads.txt"].split()
    L = len(reads[0])
    fq = tmp_path / "in.fastq"
    fq.write_bytes(b"".join(b"@T.%d\n%s\n+\n%s\n" % (i, r, b"H" * L) for i, r in enumerate(reads)))
    base = ol.stage_dir(tmp_path, {})
    harc_amd.preprocess(str(fq), base, L)
    got = ol.read_dir(base)
    for f in ["input_clean.dna", "input_N.dna", "numreads.bin", "read_order_N.bin"]:
        assert got[f] == g["stage1/" + f], f


def test_preprocess_rejects_variable_length(tmp_path):
    import harc_amd
    from tests import oracle_lib as ol
    fq = tmp_path / "bad.fastq"
    fq.write_bytes(b"@a\nACGT\n+\nHHHH\n@b\nACG\n+\nHHH\n")
    base = ol.stage_dir(tmp_path, {})
    with pytest.raises(harc_amd.HarcAmdError):
        harc_amd.preprocess(str(fq), base, 4)          # preprocess.cpp:92-97


def test_harc_driver_usage_without_gpu(tmp_path):
    """the bash driver's argument handling mirrors the reference's (harc:208-230): -h prints the usage, no mode is an error,
    a read longer than 255 is refused before anything touches the GPU (harc:46-49)"""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([os.path.join(root, "harc"), "-h"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0 and "Compression" in r.stdout and "-p Preserve order of reads" in r.stdout
    r = subprocess.run([os.path.join(root, "harc"), "-t", "4"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode != 0 or "required" in r.stdout
    fq = tmp_path / "long.fastq"
    fq.write_bytes(b"@a\n" + b"A" * 300 + b"\n+\n" + b"H" * 300 + b"\n")
    r = subprocess.run([os.path.join(root, "harc"), "-c", str(fq)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 1 and "Maximum read length exceeded" in r.stdout
