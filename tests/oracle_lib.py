"""ctypes loader for the CPU oracle (oracle/harc_oracle.c).  Test infrastructure only."""
import ctypes as C
import io
import os
import subprocess
import tarfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
GOLDEN_DIR = os.path.join(ROOT, "tests", "golden")

_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    so = os.environ.get("HARC_ORACLE_LIB") or os.path.join(ORACLE_DIR, "liboracle.so")      # override: e.g. the ASan/UBSan build (make -C oracle asan)
    src = os.path.join(ORACLE_DIR, "harc_oracle.c")
    if "HARC_ORACLE_LIB" not in os.environ and (not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src)):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    lib = C.CDLL(so)
    u32p, u64p, u8p = C.POINTER(C.c_uint32), C.POINTER(C.c_uint64), C.POINTER(C.c_uint8)
    lib.harc_oracle_reorder.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.c_uint32, u32p, u64p]
    lib.harc_oracle_encoder.argtypes = [C.c_char_p, C.c_int, C.c_uint32, u32p, u32p]
    lib.harc_oracle_pack_order.argtypes = [C.c_char_p]
    lib.harc_oracle_decoder.argtypes = [C.c_char_p, C.c_uint32]
    lib.harc_oracle_preprocess.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_char_p]
    lib.harc_oracle_quality.argtypes = [C.c_char_p, C.c_size_t, C.c_int, C.c_int, C.c_char_p]
    lib.harc_oracle_stage1_mem.argtypes = [C.c_char_p, C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, u32p, u8p, u8p, u8p, u32p, u32p, u32p, u32p, u64p]
    _lib = lib
    return lib


def golden_cases():
    return sorted(f[:-7] for f in os.listdir(GOLDEN_DIR) if f.endswith(".tar.xz") and not f.startswith("q_"))


def quality_cases():
    """-q fixtures (ids + quality values), oracle/make_goldens.py QCASES"""
    return sorted(f[:-7] for f in os.listdir(GOLDEN_DIR) if f.endswith(".tar.xz") and f.startswith("q_"))


def load_golden(name):
    """-> dict arcname -> bytes"""
    out = {}
    with tarfile.open(os.path.join(GOLDEN_DIR, name + ".tar.xz"), "r:xz") as tf:
        for m in tf.getmembers():
            out[m.name] = tf.extractfile(m).read()
    return out


def stage_dir(tmpdir, files):
    """write {name: bytes} into <tmpdir>/output/ and return tmpdir as the reference's <basedir>"""
    od = os.path.join(str(tmpdir), "output")
    os.makedirs(od, exist_ok=True)
    for k, v in files.items():
        with open(os.path.join(od, k), "wb") as f:
            f.write(v)
    return str(tmpdir)


def read_dir(basedir):
    od = os.path.join(basedir, "output")
    out = {}
    for f in sorted(os.listdir(od)):
        p = os.path.join(od, f)
        if os.path.isfile(p):
            with open(p, "rb") as fh:
                out[f] = fh.read()
    return out


STAGE1_FILES = ["temp.dna", "temp.dna.singleton", "read_rev.txt", "tempflag.txt", "temppos.txt", "read_order.bin",
                "read_order.bin.singleton"]


def stage2_files(E=1):
    fs = ["read_order.bin", "read_order_N_pe.bin", "input_N.dna", "read_meta.txt", "read_singleton.txt", "read_singleton.txt.tail"]
    for e in range(E):
        for stem in ["read_seq.txt", "read_pos.txt", "read_noise.txt", "read_noisepos.txt", "read_rev.txt"]:
            fs.append(f"{stem}.{e}")
        fs += [f"read_seq.txt.{e}.tail", f"read_rev.txt.{e}.tail"]
    return fs


def md5_cases():
    """exact mode at CONFIG size (oracle/make_goldens.py MD5CASES): generator call + md5 of every file of the reference at -t 1"""
    return sorted(f[4:-5] for f in os.listdir(GOLDEN_DIR) if f.startswith("md5_") and f.endswith(".json"))


def load_md5_case(name):
    """-> (meta dict, reads as [n, L] uint8) -- the reads are regenerated and checked against the recorded md5"""
    import hashlib
    import json
    from tests import gen
    with open(os.path.join(GOLDEN_DIR, "md5_" + name + ".json")) as f:
        meta = json.load(f)
    kw = {k: v for k, v in meta["gen"].items() if k != "function"}
    arr = gen.reads_array_big(**kw)
    got = hashlib.md5(arr.tobytes()).hexdigest()
    assert got == meta["reads_md5"], f"{name}: the generator gives other reads here ({got}) than where the fixture was made ({meta['reads_md5']})"
    return meta, arr


def md5_mismatches(files, want, names):
    """names whose bytes in `files` do not have the md5 (and size) recorded in want = {name: [md5, size]}"""
    import hashlib
    return [f for f in names if f not in files or [hashlib.md5(files[f]).hexdigest(), len(files[f])] != want[f]]
