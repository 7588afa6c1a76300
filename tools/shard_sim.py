# scratch: what one GPU sees in the 8-GPU bucket-shard run (weak scaling): 8x genome, 8x reads, keep bucket 0
import sys, time, torch
sys.path.insert(0, '.')
import harc_amd, bench
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
n, L, G, err, _ = bench.WORKLOADS[sys.argv[2] if len(sys.argv) > 2 else "c2"]
dev = torch.device("cuda", 0)
p = harc_amd.default_params(L, num_thr=8)
h = harc_amd.HarcAmd(p)
parts = []
for r in range(world):
    reads = bench.synth_reads(n, L, G * world, err, 1000 + r, dev)
    clean = reads[~(reads == ord("N")).any(1)].contiguous()
    torch.cuda.synchronize()          # the library works on its own stream: inputs must be complete
    packed = torch.empty((clean.shape[0], (2 * L + 63) // 64), dtype=torch.int64, device=dev)
    h.pack_reads_device(clean.data_ptr(), clean.shape[0], L, packed.data_ptr())
    b = torch.empty((clean.shape[0],), dtype=torch.int32, device=dev)
    h.bucket_reads_device(packed.data_ptr(), clean.shape[0], world, b.data_ptr())
    parts.append(packed[b == 0])
shard = torch.cat(parts).contiguous()
import hashlib
print("shard reads", shard.shape[0], hashlib.md5(shard.cpu().numpy().tobytes()).hexdigest(), flush=True)
for it, (K, rpc) in enumerate([(0, 0), (0, 0), (0, 1024), (0, 512), (0, 256)]):
    p = harc_amd.default_params(L, num_thr=8, num_chains=K, reads_per_chain=rpc); h = harc_amd.HarcAmd(p)
    torch.cuda.synchronize(); t0 = time.time()
    h.set_reads_packed_device(shard.data_ptr(), shard.shape[0]); h.reorder(); h.encode()
    torch.cuda.synchronize(); dt = time.time() - t0
    c = h.counters()
    print(f"iter {it}: {dt*1e3:.1f} ms -> {shard.shape[0]/dt/1e6:.1f} Mreads/s/GPU rounds={c.rounds} unmatched={c.unmatched} contigs={c.contigs} seq_bases={c.seq_bases} K={c.chains} conflicts={c.conflicts} probes={c.probes} cands={c.candidates} bigbins2={c.bins_over_maxsearch} chain_ms={c.chain_ms:.1f} encode_ms={c.encode_ms:.1f} index_ms={c.index_ms:.1f} md5={hashlib.md5(h.stream('S1_ORDER')).hexdigest()[:10]}", flush=True)
