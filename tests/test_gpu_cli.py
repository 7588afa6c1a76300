"""The drop-in driver: ./harc -c FASTQ [-p] [-t N] [-k K] end to end on the GPU; the archive is unpacked and decoded with the
oracle's restatement of decoder.cpp (the reference's own -d would read the same files)."""
import os
import shutil
import subprocess
import tarfile

import pytest

from tests import gen
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("flags,E,packer", [([], 8, "auto"), (["-t", "3", "-k", "1"], 3, "none"), (["-p", "-t", "2"], 2, "xz")])
def test_harc_c_roundtrip(flags, E, packer, oracle, tmp_path, monkeypatch):
    monkeypatch.setenv("HARC_AMD_STAGE3", packer)                # stage III (harc:102-109): auto == xz in this image (no bsc / 7z)
    L = 100
    txt = gen.reads_text(99, 20000, L, 150000, err=0.01)
    reads = txt.split()
    fq = tmp_path / "sample.fastq"
    fq.write_bytes(b"".join(b"@T.%d\n%s\n+\n%s\n" % (i, r, b"H" * L) for i, r in enumerate(reads)))
    r = subprocess.run([os.path.join(ROOT, "harc"), "-c", str(fq)] + flags, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "were unmatched" in r.stdout and "singleton reads were aligned" in r.stdout        # the reference's counters (reorder.cpp:701, encoder.cpp:506-508)
    arc = tmp_path / "sample.harc"
    assert arc.exists() and not (tmp_path / "output").exists()
    out = tmp_path / "x" / "output"
    out.mkdir(parents=True)
    with tarfile.open(arc) as tf:
        tf.extractall(out)
    xzs = [f for f in os.listdir(out) if f.endswith(".xz")]
    assert bool(xzs) == (packer != "none")
    for f in xzs:
        subprocess.check_call(["xz", "-d", str(out / f)])
    for s in ["read_noise", "read_noisepos", "read_pos", "read_seq", "read_rev"]:
        with tarfile.open(out / (s + ".tar")) as tf:
            tf.extractall(out)
    assert len([f for f in os.listdir(out) if f.startswith("read_pos.txt")]) == E              # harc:171 discovers num_thr_e this way
    if "-p" in flags:
        assert (out / "read_order.bin").exists() and (out / "read_order.bin.tail").exists() and (out / "read_order_N.bin").exists()
    else:
        assert not (out / "read_order.bin").exists()
    assert oracle.harc_oracle_decoder(str(tmp_path / "x").encode(), E) == 0
    assert sorted((out / "output.dna").read_bytes().split()) == sorted(reads)
    r = subprocess.run([os.path.join(ROOT, "harc"), "-d", str(arc)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)   # GPU decoder
    assert r.returncode == 0, r.stdout[-2000:]
    assert (tmp_path / "sample.dna.d").read_bytes() == (out / "output.dna").read_bytes()
    if "-p" in flags:                                                                           # ./harc -d -p: the FASTQ's own order
        os.remove(tmp_path / "sample.dna.d")
        r = subprocess.run([os.path.join(ROOT, "harc"), "-d", str(arc), "-p"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-2000:]
        assert (tmp_path / "sample.dna.d").read_bytes() == txt
    else:
        r = subprocess.run([os.path.join(ROOT, "harc"), "-d", str(arc), "-p"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 1 and "Not compressed using -p flag" in r.stdout                 # harc:149-153


def test_harc_refuses_existing_output_dir(tmp_path):
    fq = tmp_path / "s.fastq"
    fq.write_bytes(b"@a\nACGT\n+\nHHHH\n")
    (tmp_path / "output").mkdir()
    r = subprocess.run([os.path.join(ROOT, "harc"), "-c", str(fq)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 1 and "already exists" in r.stdout                                   # harc:38-41


@pytest.mark.parametrize("flags", [["-q", "-t", "2"], ["-p", "-q", "-t", "3"]])
def test_harc_q_writes_quality_and_ids(flags, tmp_path):
    """-q: <name>.quality and <name>.id next to the archive (harc:116-128); in file order with -p, else lined up with ./harc -d"""
    import numpy as np
    L = 100
    reads = gen.reads_text(7, 12000, L, 90000, err=0.01).split()
    rs = np.random.RandomState(3)
    quals = [bytes(35 if c == 78 else 50 + int(x) for c, x in zip(r, rs.randint(0, 20, L))) for r in reads]
    ids = [b"@run.%d/%d" % (i, 1 + i % 2) for i in range(len(reads))]
    fq = tmp_path / "s.fastq"
    fq.write_bytes(b"".join(b"%s\n%s\n+\n%s\n" % t for t in zip(ids, reads, quals)))
    r = subprocess.run([os.path.join(ROOT, "harc"), "-c", str(fq)] + flags, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    q = (tmp_path / "s.quality").read_bytes().split(b"\n")[:-1]
    i = (tmp_path / "s.id").read_bytes().split(b"\n")[:-1]
    assert not (tmp_path / "output").exists()
    if "-p" in flags:
        assert q == quals and i == ids
        return
    assert sorted(q) == sorted(quals) and len(i) == len(ids)
    r = subprocess.run([os.path.join(ROOT, "harc"), "-d", str(tmp_path / "s.harc")], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    dec = (tmp_path / "s.dna.d").read_bytes().split()
    da = np.frombuffer(b"".join(dec), dtype=np.uint8).reshape(-1, L)
    qa = np.frombuffer(b"".join(q), dtype=np.uint8).reshape(-1, L)
    assert ((da == ord("N")) == (qa == ord("#"))).all()
