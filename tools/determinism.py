import sys, time, torch, hashlib
sys.path.insert(0, '.')
import harc_amd, bench
world = int(sys.argv[1]); n, L, G, err, _ = bench.WORKLOADS["c2"]
scale = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
n = int(n * scale); G = int(G * scale)
dev = torch.device("cuda", 0)
p = harc_amd.default_params(L, num_thr=8)
h0 = harc_amd.HarcAmd(p)
parts = []
for r in range(world):
    reads = bench.synth_reads(n, L, G * world, err, 1000 + r, dev)
    clean = reads[~(reads == ord("N")).any(1)].contiguous()
    packed = torch.empty((clean.shape[0], 4), dtype=torch.int64, device=dev)
    h0.pack_reads_device(clean.data_ptr(), clean.shape[0], L, packed.data_ptr())
    b = torch.empty((clean.shape[0],), dtype=torch.int32, device=dev)
    h0.bucket_reads_device(packed.data_ptr(), clean.shape[0], world, b.data_ptr())
    parts.append(packed[b == 0])
shard = torch.cat(parts).contiguous(); torch.cuda.synchronize()
print("shard", shard.shape[0], hashlib.md5(shard.cpu().numpy().tobytes()).hexdigest(), flush=True)
for i in range(4):
    h = harc_amd.HarcAmd(p) if i % 2 else h0
    t0 = time.time(); h.set_reads_packed_device(shard.data_ptr(), shard.shape[0]); h.reorder(); dt = time.time() - t0
    c = h.counters()
    print(i, f"{dt*1e3:.0f} ms rounds={c.rounds} unmatched={c.unmatched} conflicts={c.conflicts} md5={hashlib.md5(h.stream('S1_ORDER')).hexdigest()[:12]}", flush=True)
