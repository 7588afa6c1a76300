# k_reseed_mg with several passes per launch and workgroups that read the per-pass counts late (HARC_AMD_RESEED_STRESS): against the single workgroup
import os, sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench, harc_amd
from tests.test_gpu_replicate import _one_gpu
n, K = int(sys.argv[1]), int(sys.argv[2])
arr = bench.synth_reads(n, 100, int(n * 100 / 26), 0.005, 5, torch.device("cuda", 0)).cpu().numpy()
os.environ["HARC_AMD_RESEED_MG"] = "0"
want, _ = _one_gpu(arr, 100, 2, K, 16)
for st in sys.argv[3:]:
    os.environ["HARC_AMD_RESEED_MG"] = "1"; os.environ["HARC_AMD_RESEED_STRESS"] = st
    try:
        t0 = time.time(); got, _ = _one_gpu(arr, 100, 2, K, 16); dt = time.time() - t0
        bad = [k for k in want if got[k] != want[k]]
        print("stress", st, "->", "OK" if not bad else "DIFF " + ",".join(bad[:4]), "%.2f s" % dt, flush=True)
    except Exception as e:
        print("stress", st, "-> EXC", repr(e)[:200], flush=True)
