"""Design (R) of the multi-GPU split -- replicate the reads, partition the chains (harc_amd_replicate_exchange): after
one all-gather every rank holds the whole job, walks only the chains it owns and all-gathers the walked steps once per super-round.  What
harc_amd_reorder / harc_amd_encode deliver must be, on EVERY rank, byte for byte what ONE GPU delivers on the concatenated input -- which
in turn is the oracle's (tests/test_gpu_parity.py).  World 2 and 3 share the one GPU of the box over the mailbox transport (RCCL refuses
two ranks on a device); the RCCL calls themselves run at world size 1 in a fresh process (tests/test_gpu_multigpu.py does that for the
bucket exchange; here for the replicating one)."""
import os
import subprocess
import sys
import textwrap
import threading

import pytest

from tests import gen, shard_model
from tests import oracle_lib as ol

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STREAMS = [("read_seq.txt", "S2_SEQ"), ("read_pos.txt", "S2_POS"), ("read_noise.txt", "S2_NOISE"), ("read_noisepos.txt", "S2_NOISEPOS"), ("read_rev.txt", "S2_REV")]
WHOLE = [("read_order.bin", "S2_ORDER"), ("read_order_N_pe.bin", "S2_ORDER_N_PE"), ("read_singleton.txt", "S2_SINGLETON"),
         ("read_singleton.txt.tail", "S2_SINGLETON_TAIL"), ("input_N.dna", "S2_INPUT_N")]
S1 = [("temp.dna", "S1_TEMP_DNA"), ("read_order.bin.s1", "S1_ORDER"), ("tempflag.txt", "S1_FLAG"), ("temppos.txt", "S1_POS"), ("read_rev.txt.s1", "S1_REV")]


def _collect(h, E):
    f = {}
    for e in range(E):
        for stem, sid in STREAMS:
            f["%s.%d" % (stem, e)] = h.stream(sid, e)
        f["read_seq.txt.%d.tail" % e] = h.stream("S2_SEQ_TAIL", e)
        f["read_rev.txt.%d.tail" % e] = h.stream("S2_REV_TAIL", e)
    for name, sid in WHOLE:
        f[name] = h.stream(sid)
    return f


def _one_gpu(arr, L, E, K, S):
    import harc_amd
    hasN = (arr == ord("N")).any(1)
    h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=E, num_chains=K, num_steps=S))
    h.set_reads_ascii(shard_model.lines(arr[~hasN]), int((~hasN).sum()), L + 1)
    h.set_nreads_ascii(shard_model.lines(arr[hasN]), int(hasN.sum()), L + 1)
    h.reorder(); h.encode()
    f, c = _collect(h, E), h.counters()
    h.close()
    return f, c


def _ranks(world, slices, L, E, K, S, mbox):
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
            info = h.replicate_exchange()
            h.reorder(); h.encode()
            res[r] = dict(files=_collect(h, E), info=info, counters=h.counters(), sig=h.decode_signature())
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


def _assemble(res, E, L):
    """the single-GPU streams from the pieces of a run with stage II partitioned over the ranks (what harc_amd_merge_shard_files does with the
    rank parts): a shard's streams from the rank that owns it, read_order.bin / read_order_N_pe.bin = the aligned parts of all ranks, then the
    unaligned parts (encoder.cpp:457-503), read_singleton.txt re-packed across the joints, input_N.dna concatenated"""
    out = {}
    for e in range(E):
        for stem in [st for st, _ in STREAMS] + ["read_seq.txt.tail", "read_rev.txt.tail"]:
            k = "%s.%d" % (stem, e) if not stem.endswith(".tail") else "%s.%d.tail" % (stem[:-5], e)
            owners = [r for r in res if r["files"]["read_pos.txt.%d" % e] or r is res[-1]]
            out[k] = owners[0]["files"][k]
        assert sum(1 for r in res if r["files"]["read_pos.txt.%d" % e]) <= 1, "shard %d written by two ranks" % e
    a, u, na, nu, bases, ntext = b"", b"", b"", b"", b"", b""
    for r in res:
        f = r["files"]
        sg, tl = f["read_singleton.txt"], f["read_singleton.txt.tail"]
        US = (4 * len(sg) + len(tl)) // L
        UN = len(f["input_N.dna"]) // (L + 1)
        o, on = f["read_order.bin"], f["read_order_N_pe.bin"]
        a += o[:len(o) - 4 * US]; u += o[len(o) - 4 * US:]
        na += on[:len(on) - 4 * UN]; nu += on[len(on) - 4 * UN:]
        bases += b"".join(bytes(b"ACGT"[(b >> (2 * k)) & 3] for k in range(4)) for b in sg) + tl
        ntext += f["input_N.dna"]
    out["read_order.bin"] = a + u
    out["read_order_N_pe.bin"] = na + nu
    code = {65: 0, 67: 1, 71: 2, 84: 3}
    nb = len(bases) // 4
    out["read_singleton.txt"] = bytes(code[bases[4 * i]] | (code[bases[4 * i + 1]] << 2) | (code[bases[4 * i + 2]] << 4) | (code[bases[4 * i + 3]] << 6) for i in range(nb))
    out["read_singleton.txt.tail"] = bases[4 * nb:]
    out["input_N.dna"] = ntext
    return out


def test_replicated_ranks_large_run_kernels(tmp_path, monkeypatch):
    """three ranks with the kernels of a run of hundreds of millions of reads forced (dense launch, wave-uniform scan, k_reseed by 64 workgroups with
    narrowed passes and delayed workgroups) == one GPU with the plain ones; the per-batch digest check of the replicas runs on the way"""
    import numpy as np
    os.environ["HARC_AMD_MAILBOX_TIMEOUT"] = "180"
    arr = gen.reads_array(411, 300000, 100, 1200000, err=0.01)
    want, cw = _one_gpu(arr, 100, 2, 6000, 16)
    for k, v in {"HARC_AMD_QUAD": "0", "HARC_AMD_DENSE": "1", "HARC_AMD_SEQ": "1", "HARC_AMD_RESEED_MG": "1", "HARC_AMD_RESEED_WIN": "256", "HARC_AMD_RESEED_STRESS": "3"}.items():
        monkeypatch.setenv(k, v)
    res = _ranks(3, shard_model.slices_of(arr, 3), 100, 2, 6000, 16, str(tmp_path))
    got = _assemble(res, 2, 100)                                   # stage II partitioned over the three ranks (two shards: one rank has none)
    bad = [k for k in want if got[k] != want[k]]
    assert not bad, bad
    for r in range(3):
        assert res[r]["counters"].rounds == cw.rounds


@pytest.mark.parametrize("part", [True, False])
@pytest.mark.parametrize("world,n,L,err,E,K,S,lowc", [(2, 30000, 100, 0.01, 2, 9, 16, False), (3, 24000, 100, 0.02, 1, 0, 16, False), (2, 9000, 150, 0.01, 1, 64, 8, False),
                                                       (3, 20000, 100, 0.004, 3, 24, 16, True), (2, 5000, 100, 0.0, 1, 1, 16, False), (3, 40000, 100, 0.01, 8, 0, 16, False),
                                                       (2, 30000, 100, 0.01, 5, 16, 16, True)])
def test_replicated_ranks_equal_one_gpu(world, n, L, err, E, K, S, lowc, part, tmp_path, monkeypatch):
    """a design-(R) run == the single-GPU run on the concatenated input, every stage-II file.  part: stage II partitioned over the ranks by
    encoder shard, ONE all-reduce(min) of the claims (the pieces of the ranks put together as harc_amd_merge_shard_files does); else
    (HARC_AMD_S2_PART=0) stage II replicated: every rank holds every file.  lowc: repeats and poly-A runs (the cooperative kernel's walks are
    partitioned too; stage-II bins above maxsearch: their probes travel to every rank with their window words); K = 1: one chain, one owner,
    the other ranks only follow; E = 1 with three ranks: two ranks without a shard"""
    import numpy as np
    os.environ["HARC_AMD_MAILBOX_TIMEOUT"] = "180"
    monkeypatch.setenv("HARC_AMD_S2_PART", "1" if part else "0")
    if lowc:
        txt = gen.reads_text_lowcomplexity(31 + world, n, L, 50000, err=err)
        arr = np.frombuffer(txt, dtype=np.uint8).reshape(-1, L + 1)[:, :L].copy()
    else:
        arr = gen.reads_array(177 + world, n, L, 8 * n, err=err)
    want, cw = _one_gpu(arr, L, E, K, S)
    sl = shard_model.slices_of(arr, world)
    mbox = tmp_path / "mbox"
    mbox.mkdir()
    res = _ranks(world, sl, L, E, K, S, str(mbox))
    nclean = int((~(arr == ord("N")).any(1)).sum())
    if part:
        got = _assemble(res, E, L)
        for k, v in want.items():
            assert got[k] == v, "%d ranks, stage II partitioned: %s differs from the single-GPU run" % (world, k)
        assert sum(res[r]["sig"][0] for r in range(world)) == n           # every read decoded by exactly one rank
        assert sum(res[r]["counters"].aligned_singletons for r in range(world)) == cw.aligned_singletons and sum(res[r]["counters"].aligned_N for r in range(world)) == cw.aligned_N
    for r in range(world):
        info = res[r]["info"]
        assert info[0] == nclean and info[2] == n and info[6] == nclean, (r, info)
        if not part:
            for k, v in want.items():
                assert res[r]["files"][k] == v, "rank %d of %d: %s differs from the single-GPU run" % (r, world, k)
            assert res[r]["sig"][0] == n
        c = res[r]["counters"]
        assert (c.unmatched, c.rounds, c.conflicts, c.n_main, c.n_singleton) == (cw.unmatched, cw.rounds, cw.conflicts, cw.n_main, cw.n_singleton)


RCCL_WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    import harc_amd
    from harc_amd import multigpu
    from tests import gen, shard_model
    L, E, K, S = 100, 2, 40, 16
    arr = gen.reads_array(505, 12000, L, 90000, err=0.01)
    hasN = (arr == ord("N")).any(1)
    def run(repl):
        h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=E, num_chains=K, num_steps=S))
        if repl:
            assert multigpu.init_comm(h, dist, dev) == (1, 0)
        h.set_reads_ascii(shard_model.lines(arr[~hasN]), int((~hasN).sum()), L + 1)
        h.set_nreads_ascii(shard_model.lines(arr[hasN]), int(hasN.sum()), L + 1)
        if repl:
            info = h.replicate_exchange()
            assert info[0] == int((~hasN).sum()) and info[1] == int(hasN.sum()), info
        h.reorder(); h.encode()
        out = [h.stream(sid, e) for e in range(E) for sid in ("S2_SEQ", "S2_POS", "S2_NOISE", "S2_NOISEPOS", "S2_REV")] + [h.stream("S2_ORDER"), h.stream("S2_SINGLETON")]
        if repl:
            h.comm_barrier()
        h.close()
        return out
    one = run(False)
    assert run(True) == one
    # ... and stage II PARTITIONED at world 1 (HARC_AMD_S2_PART=2): the all-reduce(min) of the claims and the exchange of the large-bin events go
    # through RCCL on this one GPU; the rank owns every encoder shard, so the streams are the single GPU's
    os.environ["HARC_AMD_S2_PART"] = "2"
    assert run(True) == one
    txt = gen.reads_text_bigbin_stage2(77, n_dupN=2500)      # bins above maxsearch in stage II: their probes travel with their window words
    arr = np.frombuffer(txt, dtype=np.uint8).reshape(-1, L + 1)[:, :L].copy()
    hasN = (arr == ord("N")).any(1)
    K = 4
    part = run(True)
    os.environ["HARC_AMD_S2_PART"] = "0"
    assert part == run(True) == run(False)
    dist.destroy_process_group()
    print("RCCL_OK")
""") % ROOT


def test_replicate_exchange_over_rccl_world1(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(RCCL_WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
                          "--master-port", "29549", str(script)], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, out.stdout[-4000:]


def _members(arc, tmp):
    """every stream file of an archive made with HARC_AMD_STAGE3=none -> {name: bytes}"""
    import tarfile
    out = {}
    d = tmp / ("x_" + arc.stem)
    d.mkdir()
    with tarfile.open(arc) as tf:
        tf.extractall(d)
    for p in sorted(d.iterdir()):
        if p.suffix == ".tar":
            with tarfile.open(p) as tf:
                for m in tf.getmembers():
                    if m.isfile():
                        out[os.path.basename(m.name)] = tf.extractfile(m).read()
        elif p.is_file():
            out[p.name] = p.read_bytes()
    return out


@pytest.mark.parametrize("world,flags", [(2, ["-p", "-t", "2"]), (3, ["-t", "1"]), (2, ["-p", "-q", "-t", "2"])])
def test_harc_g_replicate_archive_equals_single_gpu_archive(world, flags, tmp_path):
    """HARC_AMD_MG_MODE=replicate ./harc -c -g <world>: the archive holds byte for byte the stream files of ./harc -c on one GPU"""
    import numpy as np
    L = 100
    txt = gen.reads_text(321, 18000, L, 140000, err=0.01)
    reads = txt.split()
    rs = np.random.RandomState(11)
    quals = [bytes([64] + [50 + int(x) for x in rs.randint(0, 20, L - 1)]) for _ in reads]
    ids = [b"@run.%d/%d" % (i, 1 + i % 2) for i in range(len(reads))]
    fq = b"".join(b"%s\n%s\n+\n%s\n" % t for t in zip(ids, reads, quals))
    got = {}
    for tag, extra_env, extra in (("one", {}, []), ("repl", {"HARC_AMD_MG_MODE": "replicate", "HARC_AMD_XPORT": "mailbox", "HARC_AMD_SHARE_DEVICE": "0", "HARC_AMD_MAILBOX_TIMEOUT": "180"}, ["-g", str(world)])):
        d = tmp_path / tag
        d.mkdir()
        (d / "s.fastq").write_bytes(fq)
        env = dict(os.environ, HARC_AMD_STAGE3="none", **extra_env)
        r = subprocess.run([os.path.join(ROOT, "harc"), "-c", str(d / "s.fastq")] + extra + flags, cwd=ROOT, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:]
        assert not (d / "output").exists()
        got[tag] = _members(d / "s.harc", d)
        if "-q" in flags:
            got[tag]["s.quality"] = (d / "s.quality").read_bytes(); got[tag]["s.id"] = (d / "s.id").read_bytes()
    assert sorted(got["one"]) == sorted(got["repl"]), (sorted(got["one"]), sorted(got["repl"]))
    for k in got["one"]:
        assert got["one"][k] == got["repl"][k], "%s differs between the single-GPU archive and the design-(R) archive of %d ranks" % (k, world)
    assert not any(".shard" in k or ".mbox" in k for k in got["repl"])
