import os, sys, threading
sys.path.insert(0, '/root/repo')
os.chdir('/root/repo')
import numpy as np, torch
torch.cuda.init()
import harc_amd, bench
from tests import shard_model
from tests.test_gpu_replicate import _one_gpu, _ranks
os.environ["HARC_AMD_MAILBOX_TIMEOUT"] = "600"
import tempfile
n, L, W = int(sys.argv[1]), 100, int(sys.argv[2])
arr = bench.synth_reads(n, L, int(n * 100 / 26), 0.005, int(os.environ.get("PROBE_SEED", "77")), torch.device("cuda", 0)).cpu().numpy()
for K, env in [(int(x.split(':')[0]), x.split(':')[1]) for x in sys.argv[3:]]:
    for kv in env.split(','):
        if kv: k, v = kv.split('='); os.environ[k] = v
    try:
        want, cw = _one_gpu(arr, L, 2, K, 16)
        with tempfile.TemporaryDirectory(dir="/dev/shm") as d:
            res = _ranks(W, shard_model.slices_of(arr, W), L, 2, K, 16, d)
        bad = [k for k in want if any(res[r]["files"][k] != want[k] for r in range(W))]
        print("K", K, env, "->", "OK" if not bad else "DIFF " + ",".join(bad[:3]), flush=True)
    except Exception as e:
        print("K", K, env, "-> EXC", repr(e)[:300], flush=True)
    for kv in env.split(','):
        if kv: os.environ.pop(kv.split('=')[0], None)
