import os, sys, time, torch
sys.path.insert(0, '.')
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29578"); os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import torch.distributed as dist
import harc_amd, bench
from harc_amd import multigpu
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", device_id=dev)
n, L, G, err, _ = bench.WORKLOADS["c2"]
reads = bench.synth_reads(n, L, G, err, 1000, dev)
clean = reads[~(reads == ord("N")).any(1)].contiguous()
h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=8, reads_per_chain=1024))
sh = multigpu.BucketSharder(h, dist, dev, L)
packed = sh.pack(clean)
for it in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    b = sh.buckets(packed); torch.cuda.synchronize(); t1 = time.perf_counter()
    order = torch.sort(b, stable=True).indices; send = packed[order].contiguous(); counts = torch.bincount(b, minlength=1); torch.cuda.synchronize(); t2 = time.perf_counter()
    m = sh.exchange_and_set(packed); torch.cuda.synchronize(); t3 = time.perf_counter()
    print(f"bucket {1e3*(t1-t0):.2f} ms, sort+gather {1e3*(t2-t1):.2f} ms, whole exchange_and_set {1e3*(t3-t2):.2f} ms")
dist.destroy_process_group()
