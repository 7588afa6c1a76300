import sys, time, torch, numpy as np
sys.path.insert(0, '.')
import harc_amd
from tests import gen
L = 100
world = int(sys.argv[1]); n = int(sys.argv[2]); seeds = [int(x) for x in sys.argv[3].split(',')]
dev = torch.device("cuda", 0)
p = harc_amd.default_params(L, num_thr=8)
h = harc_amd.HarcAmd(p)
for seed in seeds:
    parts = []
    for r in range(world):
        arr = gen.reads_array(seed * 100 + r, n, L, int(n * 1.9) * world, err=0.005)
        arr = arr[~(arr == ord('N')).any(1)]
        t = torch.from_numpy(arr).to(dev)
        packed = torch.empty((t.shape[0], 4), dtype=torch.int64, device=dev)
        h.pack_reads_device(t.data_ptr(), t.shape[0], L, packed.data_ptr())
        b = torch.empty((t.shape[0],), dtype=torch.int32, device=dev)
        h.bucket_reads_device(packed.data_ptr(), t.shape[0], world, b.data_ptr())
        parts.append(packed[b == 0])
    shard = torch.cat(parts).contiguous(); torch.cuda.synchronize()
    t0 = time.time(); h.set_reads_packed_device(shard.data_ptr(), shard.shape[0]); h.reorder(); dt = time.time() - t0
    c = h.counters()
    print(f"seed={seed} shard={shard.shape[0]} {dt*1e3:.0f} ms rounds={c.rounds} unmatched={c.unmatched} conflicts={c.conflicts} probes={c.probes} cands={c.candidates} chains={c.chains}", flush=True)
