"""The multi-GPU path on ONE GPU (the box has one):
 * the RCCL transport inside libharc_amd.so at world size 1, in a fresh child process under torch.distributed.run (the launcher
   runs before anything touches the GPU): communicator bootstrap through multigpu.init_comm, harc_amd_shard_exchange, then the shard's
   streams against the oracle;
 * the whole sharded logic at world size 2 and 3 with the mailbox transport (two ranks may share a GPU there, RCCL refuses that):
   every rank's shard and streams == the CPU model (tests/shard_model.py: bucket plan + oracle per shard + global ids);
 * ./harc -c -g 2 [-p] [-q] end to end -> ./harc -d [-p] gives the input back."""
import os
import subprocess
import sys
import tarfile
import textwrap
import threading

import numpy as np
import pytest

from tests import gen, shard_model
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

RCCL_WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    import harc_amd
    from harc_amd import multigpu
    from tests import gen, shard_model, oracle_lib as ol
    L, E, K, S = 100, 2, 6, 16
    arr = gen.reads_array(404, 9000, L, 70000, err=0.01)
    hasN = (arr == ord("N")).any(1)
    h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=E, num_chains=K, num_steps=S))
    world, rank = multigpu.init_comm(h, dist, dev)
    assert (world, rank) == (1, 0)
    h.set_reads_ascii(shard_model.lines(arr[~hasN]), int((~hasN).sum()), L + 1)
    h.set_nreads_ascii(shard_model.lines(arr[hasN]), int(hasN.sum()), L + 1)
    sig_in = h.input_signature()
    oracle = ol.load()
    plan = shard_model.shard_plan([arr], L, 1)
    want = shard_model.oracle_shard(oracle, plan[0], L, E, K, S, sys.argv[1])
    for rep in range(2):                                   # the second exchange reuses the shard buffers
        info = h.shard_exchange()
        assert info[0] == int((~hasN).sum()) and info[1] == int(hasN.sum()) and info[3] == 0 and info[6] == info[0] and info[7] == info[1], info
        h.reorder(); h.encode()
        assert h.decode_signature() == multigpu.allreduce_signature(dist, sig_in, dev)
        for e in range(E):
            for stem, sid in [("read_seq.txt", "S2_SEQ"), ("read_pos.txt", "S2_POS"), ("read_noise.txt", "S2_NOISE"), ("read_noisepos.txt", "S2_NOISEPOS"), ("read_rev.txt", "S2_REV")]:
                assert h.stream(sid, e) == want["%%s.%%d" %% (stem, e)], (stem, e)
        assert h.stream("S2_ORDER") == want["read_order.bin"] and h.stream("S2_ORDER_N_PE") == want["read_order_N_pe.bin"]
        assert h.stream("S2_SINGLETON") == want["read_singleton.txt"] and h.stream("S2_INPUT_N") == want["input_N.dna"]
    h.comm_barrier()
    h.close()
    dist.destroy_process_group()
    print("RCCL_OK")
""") % ROOT


@pytest.mark.parametrize("piece", ["", "40000"])
def test_rccl_exchange_world1_in_a_fresh_process(piece, tmp_path):
    """piece: the exchange posts every (array, peer) chunk as sends / receives of at most that many bytes (1 GiB by default: a single
    10 GB send never came back) -- forced small here, so that a chunk is a few dozen pieces"""
    script = tmp_path / "worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if piece:
        env["HARC_AMD_XCHG_PIECE"] = piece
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29547", str(script), str(tmp_path / "oracle")], env=env, cwd=ROOT, stdout=subprocess.PIPE,
                         stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-4000:]


def _run_ranks(world, slices, L, E, K, S, mbox):
    """`world` contexts on device 0, one thread each (the mailbox transport blocks until the peers have written)"""
    import harc_amd
    res, errs = [None] * world, []

    def work(r):
        try:
            s = slices[r]
            hasN = (s == ord("N")).any(1)
            h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=E, num_chains=K, num_steps=S))
            h.comm_init_mailbox(mbox, world, r)
            h.set_reads_ascii(shard_model.lines(s[~hasN]), int((~hasN).sum()), L + 1)
            h.set_nreads_ascii(shard_model.lines(s[hasN]), int(hasN.sum()), L + 1)
            sig = h.input_signature()
            info = h.shard_exchange()
            h.reorder(); h.encode()
            f = {}
            for e in range(E):
                for stem, sid in [("read_seq.txt", "S2_SEQ"), ("read_pos.txt", "S2_POS"), ("read_noise.txt", "S2_NOISE"), ("read_noisepos.txt", "S2_NOISEPOS"), ("read_rev.txt", "S2_REV")]:
                    f["%s.%d" % (stem, e)] = h.stream(sid, e)
                f["read_seq.txt.%d.tail" % e] = h.stream("S2_SEQ_TAIL", e); f["read_rev.txt.%d.tail" % e] = h.stream("S2_REV_TAIL", e)
            for name, sid in [("read_order.bin", "S2_ORDER"), ("read_order_N_pe.bin", "S2_ORDER_N_PE"), ("read_singleton.txt", "S2_SINGLETON"),
                              ("read_singleton.txt.tail", "S2_SINGLETON_TAIL"), ("input_N.dna", "S2_INPUT_N")]:
                f[name] = h.stream(sid)
            res[r] = dict(files=f, info=info, sig_in=sig, sig_out=h.decode_signature())
            h.comm_barrier()
            h.close()
        except Exception as ex:                                  # noqa: BLE001
            errs.append((r, repr(ex)))

    th = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not errs, errs
    return res


@pytest.mark.parametrize("world,n,L,err,E,K,S", [(2, 30000, 100, 0.01, 2, 9, 16), (3, 24000, 100, 0.02, 1, 0, 16), (2, 9000, 150, 0.01, 1, 4, 8)])
def test_two_ranks_one_gpu_match_the_cpu_model(world, n, L, err, E, K, S, oracle, tmp_path):
    os.environ["HARC_AMD_MAILBOX_TIMEOUT"] = "120"
    arr = gen.reads_array(77 + world, n, L, 8 * n, err=err)
    sl = shard_model.slices_of(arr, world)
    plan = shard_model.shard_plan(sl, L, world)
    mbox = tmp_path / "mbox"
    mbox.mkdir()
    res = _run_ranks(world, sl, L, E, K, S, str(mbox))
    tot_sig = [0, 0, 0]
    out_sig = [0, 0, 0]
    for r in range(world):
        info = res[r]["info"]
        assert info[6] == plan[r]["clean"].shape[0] and info[7] == plan[r]["withN"].shape[0], (r, info)
        assert info[0] == sum(p["clean"].shape[0] for p in plan) and info[2] == n
        Ko = K if K else gen.auto_chains(plan[r]["clean"].shape[0], clean=plan[r]["clean"])
        want = shard_model.oracle_shard(oracle, plan[r], L, E, Ko, S, str(tmp_path / ("o%d" % r)))
        for k, v in res[r]["files"].items():
            assert v == want[k], "rank %d: %s differs from the oracle on the modelled shard" % (r, k)
        for k in range(3):
            a, b = res[r]["sig_in"][k], res[r]["sig_out"][k]
            tot_sig[k] = (tot_sig[k] ^ a) if k == 2 else (tot_sig[k] + a) % (1 << 64)
            out_sig[k] = (out_sig[k] ^ b) if k == 2 else (out_sig[k] + b) % (1 << 64)
    assert tot_sig == out_sig and tot_sig[0] == n                # what went in on all ranks == what decodes on all ranks


def _fastq(reads, quals=None, ids=None):
    L = len(reads[0])
    quals = quals or [b"H" * L] * len(reads)
    ids = ids or [b"@T.%d" % i for i in range(len(reads))]
    return b"".join(b"%s\n%s\n+\n%s\n" % t for t in zip(ids, reads, quals))


@pytest.mark.parametrize("world,flags,E", [(2, ["-p", "-t", "2"], 2), (3, ["-t", "1"], 1), (2, ["-p", "-q", "-t", "1"], 1)])
def test_harc_g_roundtrip_on_one_gpu(world, flags, E, tmp_path):
    """./harc -c -g <world> with all ranks on device 0 and the mailbox transport; quality lines that begin with '@' sit at the cut"""
    L = 100
    txt = gen.reads_text(123, 21000, L, 160000, err=0.01)
    reads = txt.split()
    rs = np.random.RandomState(9)
    quals = [bytes([64] + [50 + int(x) for x in rs.randint(0, 20, L - 1)]) for _ in reads]      # every quality line starts with '@'
    ids = [b"@run.%d/%d" % (i, 1 + i % 2) for i in range(len(reads))]
    fq = tmp_path / "s.fastq"
    fq.write_bytes(_fastq(reads, quals, ids))
    env = dict(os.environ, HARC_AMD_XPORT="mailbox", HARC_AMD_SHARE_DEVICE="0", HARC_AMD_MAILBOX_TIMEOUT="120", HARC_AMD_STAGE3="none")   # raw tars: the test opens read_pos.tar
    r = subprocess.run([os.path.join(ROOT, "harc"), "-c", str(fq), "-g", str(world)] + flags, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:]
    assert "Total number of reads: %d" % len(reads) in r.stdout and "were unmatched" in r.stdout
    arc = tmp_path / "s.harc"
    assert arc.exists() and not (tmp_path / "output").exists()
    with tarfile.open(arc) as tf:
        names = tf.getnames()
    assert not any(".shard" in n or ".mbox" in n for n in names), names
    out = tmp_path / "x"
    out.mkdir()
    with tarfile.open(arc) as tf:
        tf.extractall(out)
    with tarfile.open(out / "read_pos.tar") as tf:
        assert len([n for n in tf.getnames() if "read_pos.txt" in n]) == world * E               # harc:171 discovers num_thr_e this way
    r = subprocess.run([os.path.join(ROOT, "harc"), "-d", str(arc)], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:]
    assert sorted((tmp_path / "s.dna.d").read_bytes().split()) == sorted(reads)
    if "-p" in flags:
        os.remove(tmp_path / "s.dna.d")
        r = subprocess.run([os.path.join(ROOT, "harc"), "-d", str(arc), "-p"], cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:]
        assert (tmp_path / "s.dna.d").read_bytes() == txt
    if "-q" in flags:
        assert (tmp_path / "s.quality").read_bytes().split(b"\n")[:-1] == quals
        assert (tmp_path / "s.id").read_bytes().split(b"\n")[:-1] == ids


def test_harc_g_q_without_p_is_refused(tmp_path):
    fq = tmp_path / "s.fastq"
    fq.write_bytes(_fastq(gen.reads_text(1, 2000, 100, 30000).split()))
    env = dict(os.environ, HARC_AMD_XPORT="mailbox", HARC_AMD_SHARE_DEVICE="0", HARC_AMD_MAILBOX_TIMEOUT="20")
    r = subprocess.run([os.path.join(ROOT, "harc"), "-c", str(fq), "-g", "2", "-q"], cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=300)
    assert r.returncode != 0 and "needs -p" in r.stdout
