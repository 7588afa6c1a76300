"""N>1 path on CPU: world_size 2 over gloo.  The exchange plumbing of harc_amd.multigpu.BucketSharder (stable grouping by
bucket, count exchange, one all-to-all(v)) is checked with a numpy restatement of the HIP bucket kernel injected as
bucket_fn; on the GPU the same class calls k_bucket through the C-ABI (tests/test_gpu_multigpu.py)."""
import os
import subprocess
import sys
import textwrap

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = textwrap.dedent("""
    import os, sys, numpy as np, torch, torch.distributed as dist
    sys.path.insert(0, %r)
    from harc_amd.multigpu import BucketSharder
    from tests.bucket_ref import pack2, bucket_ref
    from tests import gen
    dist.init_process_group("gloo")
    r, w = dist.get_rank(), dist.get_world_size()
    L = 100
    reads = gen.reads_array(77 + r, 3000 + 500 * r, L, 50000, err=0.0)
    packed = torch.from_numpy(pack2(reads))
    sh = BucketSharder(None, dist, torch.device("cpu"), L, bucket_fn=lambda p, nb: torch.from_numpy(bucket_ref(p.numpy(), L, nb)))
    recv = sh.exchange(packed)
    # every received read belongs to this rank's bucket
    assert (bucket_ref(recv.numpy(), L, w) == r).all()
    # global multiset preserved: gather everything on rank 0
    allp = [None] * w; allr = [None] * w
    dist.all_gather_object(allp, packed.numpy()); dist.all_gather_object(allr, recv.numpy())
    if r == 0:
        a = np.concatenate(allp); b = np.concatenate(allr)
        assert a.shape == b.shape
        assert (np.sort(a.view([('', a.dtype)] * a.shape[1]).ravel()) == np.sort(b.view([('', b.dtype)] * b.shape[1]).ravel())).all()
        # deterministic order: source-rank-major, original order inside a source
        exp = np.concatenate([p[bucket_ref(p, L, w) == 0] for p in allp])
        assert (exp == allr[0]).all()
        print("OK", [x.shape[0] for x in allr])
    dist.destroy_process_group()
""") % ROOT


def test_bucket_exchange_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", "29541", str(script)], env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert out.returncode == 0 and "OK" in out.stdout, out.stdout[-3000:]


def test_bucket_ref_groups_overlapping_reads():
    from tests.bucket_ref import pack2, bucket_ref
    from tests import gen
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
