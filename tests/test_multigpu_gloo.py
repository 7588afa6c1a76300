"""N>1 path on CPU: world_size 2 over gloo, one process per rank.  The exchange protocol of harc_amd_shard_exchange
(csrc/shard.hip: all-gather of the slice sizes -> global id offsets, stable grouping by minimizer bucket, all-gather of the
count matrix, ONE all-to-all(v) of packed reads + u32 global ids, clean reads and reads with N alike) is restated with
torch.distributed calls on numpy data and must deliver exactly tests/bucket_ref.shard_plan; every rank then runs the oracle on
its shard and writes the rank parts, rank 0 merges them with the library's host code (harc_amd_merge_shard_files) and the merged
archive must decode to the input.  On the GPU the same protocol runs inside the library (tests/test_gpu_multigpu.py)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from tests.bucket_ref import pack2, bucket_ref, bucket3_ref, shard_plan, reads_signature
    from tests import gen, shard_model, oracle_lib as ol
    from harc_amd import multigpu
    import harc_amd
    dist.init_process_group("gloo")
    r, w = dist.get_rank(), dist.get_world_size()
    L, E, K, S = 100, 1, 5, 16
    work = sys.argv[1]
    arr = gen.reads_array(2026, 7001, L, 60000, err=0.01)            # the whole job, the same on every rank
    slices = shard_model.slices_of(arr, w)
    mine = slices[r]
    hasN = (mine == ord("N")).any(1)
    clean, withN = mine[~hasN], mine[hasN]

    def a2a(rows, counts_to, width, dtype):
        # one all-to-all(v): rows grouped by destination, counts_to[p] rows for peer p
        cnt = torch.tensor(counts_to, dtype=torch.int64)
        allc = [torch.zeros(w, dtype=torch.int64) for _ in range(w)]
        dist.all_gather(allc, cnt)                                     # the count matrix, as the library's second all-gather
        recv_counts = [int(allc[p][r]) for p in range(w)]
        send = torch.from_numpy(np.ascontiguousarray(rows).view(np.uint8).reshape(-1))
        recv = torch.empty(sum(recv_counts) * width, dtype=torch.uint8)
        dist.all_to_all_single(recv, send, output_split_sizes=[c * width for c in recv_counts], input_split_sizes=[c * width for c in counts_to])
        return recv.numpy().view(dtype)

    # (1) slice sizes of every rank -> offsets of the global ids
    sizes = [torch.zeros(3, dtype=torch.int64) for _ in range(w)]
    dist.all_gather(sizes, torch.tensor([clean.shape[0], withN.shape[0], mine.shape[0]], dtype=torch.int64))
    off = [int(sum(int(s[k]) for s in sizes[:r])) for k in range(3)]
    # (2) bucket + stable grouping by destination
    b = bucket_ref(pack2(clean), L, w); o = np.argsort(b, kind="stable")
    b3 = bucket3_ref(withN, w, off[1]); o3 = np.argsort(b3, kind="stable")
    # (3)+(4) counts, then reads and ids
    cnt = np.bincount(b, minlength=w).tolist(); cnt3 = np.bincount(b3, minlength=w).tolist()
    got_clean = a2a(clean[o], cnt, L, np.uint8).reshape(-1, L)
    got_gid = a2a((off[0] + o).astype(np.uint32), cnt, 4, np.uint32)
    got_N = a2a(withN[o3], cnt3, L, np.uint8).reshape(-1, L)
    got_ngid = a2a((off[1] + o3).astype(np.uint32), cnt3, 4, np.uint32)
    plan = shard_plan(slices, L, w)[r]
    assert (got_clean == plan["clean"]).all() and (got_gid == plan["gid"]).all()
    assert (got_N == plan["withN"]).all() and (got_ngid == plan["ngid"]).all()
    # the job-wide signature check bench.py makes around the exchange: what went in == what came out, over all ranks
    sig_in = multigpu.allreduce_signature(dist, reads_signature([bytes(x) for x in mine]), torch.device("cpu"))
    sig_out = multigpu.allreduce_signature(dist, reads_signature([bytes(x) for x in got_clean] + [bytes(x) for x in got_N]), torch.device("cpu"))
    assert sig_in == sig_out == reads_signature([bytes(x) for x in arr]), (sig_in, sig_out)
    # every rank compresses its shard (oracle) and leaves its parts; rank 0 merges and decodes
    oracle = ol.load()
    f = shard_model.oracle_shard(oracle, dict(clean=got_clean, gid=got_gid, withN=got_N, ngid=got_ngid), L, E, K, S, os.path.join(work, "rank%%d" %% r))
    od = os.path.join(work, "job", "output")
    os.makedirs(od, exist_ok=True)
    shard_model.write_rank_parts(od, r, E, L, f, mine, off[2])
    dist.barrier()
    if r == 0:
        base = os.path.join(work, "job")
        harc_amd.merge_shards(base, w)
        assert oracle.harc_oracle_decoder(base.encode(), w * E) == 0
        dec = open(os.path.join(od, "output.dna"), "rb").read()
        assert sorted(dec.split()) == sorted(bytes(x) for x in arr)
        o_all = np.frombuffer(open(os.path.join(od, "read_order.bin"), "rb").read(), dtype=np.uint32)
        assert sorted(o_all.tolist()) == list(range(int((~(arr == ord("N")).any(1)).sum())))
        print("OK", got_clean.shape[0], got_N.shape[0])
    dist.barrier()
    dist.destroy_process_group()
""") % ROOT


def test_shard_protocol_and_merge_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script), str(tmp_path)], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-3000:]


def test_bucket_ref_groups_overlapping_reads():
    from tests.bucket_ref import pack2, bucket_ref
    rs = np.random.RandomState(5)
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=5000)]
    a = np.stack([genome[s:s + 100] for s in range(0, 4000, 1)])
    b = bucket_ref(pack2(a), 100, 8)
    # neighbouring reads (99-base overlap) almost always share their minimizer
    assert (b[1:] == b[:-1]).mean() > 0.9
    # a read and its reverse complement land in the same bucket (canonical k-mers)
    comp = np.zeros(256, dtype=np.uint8)
    for x, y in zip(b"ACGT", b"TGCA"):
        comp[x] = y
    rc = comp[a[:, ::-1]]
    assert (bucket_ref(pack2(rc), 100, 8) == b).all()


def test_bucket3_ref_follows_the_clean_reads():
    """a read with one N goes, most of the time, where its N-free copy goes (same minimizer unless the N destroyed it)"""
    from tests.bucket_ref import pack2, bucket_ref, bucket3_ref
    rs = np.random.RandomState(6)
    a = np.frombuffer(b"ACGT", dtype=np.uint8)[rs.randint(0, 4, size=(3000, 100))]
    withN = a.copy()
    withN[np.arange(3000), rs.randint(0, 100, size=3000)] = ord("N")
    same = (bucket3_ref(withN, 8) == bucket_ref(pack2(a), 100, 8)).mean()
    assert same > 0.7, same
    allN = np.full((4, 100), ord("N"), dtype=np.uint8)
    assert (bucket3_ref(allN, 8, 10) < 8).all()
