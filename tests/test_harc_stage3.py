"""Stage III of the ./harc driver (harc:102-109 and :155-162 of the reference) without a GPU: the stage binary is replaced by a
stand-in that writes fixed stream files, so that what is tested is the script's own work -- which packer is chosen
(HARC_AMD_STAGE3), that the independent jobs run side by side and all finish, that a failing job fails the run, and that
-d hands the decoder the very bytes -c got from the encoder."""
import hashlib
import os
import stat
import subprocess
import tarfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

STUB = r"""#!/bin/bash
# stand-in for harc_amd_stage: fixed stream files in, checksums out
set -e
cmd=$1; base=$2; out=$base/output
case $cmd in
compressfq)
    E=$5
    for ((e = 0; e < E; e++)); do
        for s in read_seq read_pos read_noise read_noisepos read_rev; do head -c $((20000 + 977 * e)) /dev/zero | tr '\0' 'A' > $out/$s.txt.$e; done
        printf 'AC' > $out/read_seq.txt.$e.tail; printf '1' > $out/read_rev.txt.$e.tail
    done
    printf 'ACGTACGT\n' > $out/input_N.dna; printf 'clean\n' > $out/input_clean.dna
    head -c 5000 /dev/zero | tr '\0' 'G' > $out/read_singleton.txt; printf 'T' > $out/read_singleton.txt.tail
    printf '100\n' > $out/read_meta.txt
    for s in read_order.bin read_order_N.bin read_order_N_pe.bin numreads.bin read_order.bin.singleton temp.dna.singleton; do head -c 4000 /dev/urandom > $out/$s; done
    echo "Reordering done, 0 were unmatched";;
pack_order) printf 'tail' > $out/read_order.bin.tail;;
decoder|decoder_preserve)
    (cd $out && find . -type f ! -name 'output.dna' | sort | xargs sha256sum) > $out/output.dna;;
*) echo "stub: unknown command $cmd"; exit 1;;
esac
"""


def _setup(tmp_path):
    stub = tmp_path / "stage_stub.sh"
    stub.write_text(STUB)
    stub.chmod(stub.stat().st_mode | stat.S_IXUSR)
    fq = tmp_path / "s.fastq"
    fq.write_bytes(b"@a\n" + b"ACGT" * 25 + b"\n+\n" + b"H" * 100 + b"\n")
    env = dict(os.environ, HARC_AMD_STAGE_BIN=str(stub))
    return fq, env


def _run(args, env):
    return subprocess.run([os.path.join(ROOT, "harc")] + args, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)


@pytest.mark.parametrize("packer,flags", [("xz", []), ("xz", ["-p", "-t", "3"]), ("none", ["-p"]), ("auto", ["-t", "16"])])
def test_stage3_roundtrip(packer, flags, tmp_path):
    fq, env = _setup(tmp_path)
    env["HARC_AMD_STAGE3"] = packer
    r = _run(["-c", str(fq)] + flags, env)
    assert r.returncode == 0, r.stdout[-2000:]
    arc = tmp_path / "s.harc"
    assert arc.exists() and not (tmp_path / "output").exists()
    with tarfile.open(arc) as tf:
        names = sorted(os.path.basename(n) for n in tf.getnames() if os.path.basename(n) not in ("", "."))
    packed = packer != "none"                                    # this image has xz and neither bsc nor 7z: auto == xz
    streams = ["read_seq.tar", "read_pos.tar", "read_noise.tar", "read_noisepos.tar", "read_rev.tar", "input_N.dna", "read_singleton.txt"]
    for s in streams:
        assert (s + ".xz" in names) == packed and (s in names) != packed, names
    assert "input_clean.dna" not in names and "temp.dna.singleton" not in names
    if "-p" in flags:
        for s in ["read_order.bin", "read_order_N.bin", "read_order_N_pe.bin"]:
            assert (s + ".xz" if packed else s) in names, names
        assert "read_order.bin.tail" in names and "numreads.bin" not in names
    else:
        assert not [n for n in names if n.startswith("read_order")]
    # -d: every file the decoder sees is what the encoder wrote (the stand-in decoder lists their checksums)
    r = _run(["-d", str(arc)] + (["-p"] if "-p" in flags else []), env)
    assert r.returncode == 0, r.stdout[-2000:]
    sums = dict(reversed(l.split(None, 1)) for l in (tmp_path / "s.dna.d").read_text().splitlines())
    E = int(flags[flags.index("-t") + 1]) if "-t" in flags else 8
    for e in range(E):
        for st in ["read_seq", "read_pos", "read_noise", "read_noisepos", "read_rev"]:
            assert sums[f"./{st}.txt.{e}"] == hashlib.sha256(b"A" * (20000 + 977 * e)).hexdigest()
        assert sums[f"./read_seq.txt.{e}.tail"] == hashlib.sha256(b"AC").hexdigest() and f"./read_rev.txt.{e}.tail" in sums
    assert sums["./input_N.dna"] == hashlib.sha256(b"ACGTACGT\n").hexdigest()
    assert sums["./read_singleton.txt"] == hashlib.sha256(b"G" * 5000).hexdigest()
    assert sums["./read_meta.txt"] == hashlib.sha256(b"100\n").hexdigest()
    assert not [k for k in sums if k.endswith(".xz")]


def test_stage3_unknown_packer_is_refused_before_any_work(tmp_path):
    fq, env = _setup(tmp_path)
    env["HARC_AMD_STAGE3"] = "bsc"                               # not installed in this image
    r = _run(["-c", str(fq)], env)
    assert r.returncode != 0 and "not installed" in r.stdout
    assert not (tmp_path / "output").exists() and not (tmp_path / "s.harc").exists()
    env["HARC_AMD_STAGE3"] = "zip"
    r = _run(["-c", str(fq)], env)
    assert r.returncode != 0 and "must be auto, bsc, xz or none" in r.stdout


def test_stage3_failing_job_fails_the_run(tmp_path):
    """one of the side-by-side jobs fails (xz refuses to overwrite an existing output): the driver must not produce an archive"""
    fq, env = _setup(tmp_path)
    stub = tmp_path / "stage_stub.sh"
    stub.write_text(STUB.replace("printf '100\\n' > $out/read_meta.txt", "printf '100\\n' > $out/read_meta.txt; printf 'x' > $out/input_N.dna.xz"))
    env["HARC_AMD_STAGE3"] = "xz"
    r = _run(["-c", str(fq)], env)
    assert r.returncode != 0 and "a stage III job failed" in r.stdout
    assert not (tmp_path / "s.harc").exists()


def test_stage3_starts_while_the_stage_program_still_writes(tmp_path):
    """the stage program names a stream on descriptor HARC_AMD_READY_FD when all its shard files are closed (ingest.hip write_shard_family); the
    driver must start that stream's tar + coder at once: the stand-in announces read_seq, waits, and finds read_seq.tar(.xz) already there
    before it writes the next stream.  A failing stage program still fails the run with jobs in flight."""
    fq, env = _setup(tmp_path)
    stub = tmp_path / "stage_stub.sh"
    body = STUB.replace("""    for ((e = 0; e < E; e++)); do
        for s in read_seq read_pos read_noise read_noisepos read_rev; do head -c $((20000 + 977 * e)) /dev/zero | tr '\\0' 'A' > $out/$s.txt.$e; done
        printf 'AC' > $out/read_seq.txt.$e.tail; printf '1' > $out/read_rev.txt.$e.tail
    done
""", """    for s in read_seq read_pos read_noise read_noisepos read_rev; do
        for ((e = 0; e < E; e++)); do head -c $((20000 + 977 * e)) /dev/zero | tr '\\0' 'A' > $out/$s.txt.$e; done
        if [ $s = read_seq ]; then for ((e = 0; e < E; e++)); do printf 'AC' > $out/read_seq.txt.$e.tail; done; fi
        if [ $s = read_rev ]; then for ((e = 0; e < E; e++)); do printf '1' > $out/read_rev.txt.$e.tail; done; fi
        echo $s >&$HARC_AMD_READY_FD
        if [ $s = read_seq ]; then
            for i in $(seq 50); do ls $out/read_seq.tar* > /dev/null 2>&1 && { touch $base/overlap_seen; break; }; sleep 0.1; done
            [ -n "$STUB_FAIL" ] && exit 7
        fi
    done
""")
    assert body != STUB
    stub.write_text(body)
    for packer in ("xz", "none"):
        env["HARC_AMD_STAGE3"] = packer
        r = _run(["-c", str(fq), "-t", "3"], env)
        assert r.returncode == 0, r.stdout[-2000:]
        assert (tmp_path / "overlap_seen").exists(), "read_seq was not packed while the stage program was still writing"
        (tmp_path / "overlap_seen").unlink()
        with tarfile.open(tmp_path / "s.harc") as tf:
            names = sorted(os.path.basename(n) for n in tf.getnames() if os.path.basename(n) not in ("", "."))
        for s in ["read_seq.tar", "read_pos.tar", "read_noise.tar", "read_noisepos.tar", "read_rev.tar"]:
            assert (s + ".xz" if packer == "xz" else s) in names, names
        assert ".ready" not in names
        r = _run(["-d", str(tmp_path / "s.harc")], env)
        assert r.returncode == 0, r.stdout[-2000:]
        sums = dict(reversed(l.split(None, 1)) for l in (tmp_path / "s.dna.d").read_text().splitlines())
        for e in range(3):
            assert sums[f"./read_seq.txt.{e}"] == hashlib.sha256(b"A" * (20000 + 977 * e)).hexdigest() and f"./read_seq.txt.{e}.tail" in sums
        (tmp_path / "s.harc").unlink(); (tmp_path / "s.dna.d").unlink()
    env["STUB_FAIL"] = "1"
    r = _run(["-c", str(fq), "-t", "3"], env)
    assert r.returncode != 0 and not (tmp_path / "s.harc").exists() and not (tmp_path / "output").exists()


def test_mistyped_multi_gpu_mode_is_refused_before_any_rank_starts(tmp_path):
    """HARC_AMD_MG_MODE is 'bucket' or 'replicate': anything else (a typo would silently give the archive with the larger consensus) ends the run with
    a message, before a rank is started and with the output directory gone"""
    fq, env = _setup(tmp_path)
    marker = tmp_path / "stage_was_started"
    (tmp_path / "stage_stub.sh").write_text("#!/bin/bash\ntouch %s\nexit 1\n" % marker)
    env["HARC_AMD_MG_MODE"] = "replicated"
    r = _run(["-c", str(fq), "-g", "2"], env)
    assert r.returncode != 0 and "HARC_AMD_MG_MODE" in r.stdout, r.stdout[-1000:]
    assert not marker.exists() and not (tmp_path / "output").exists() and not (tmp_path / "s.harc").exists()
