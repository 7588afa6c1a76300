# What ONE GPU of a bucket-sharded weak-scaling run processes, simulated on one GPU with the library's bucket function (not a multi-GPU
# measurement): `world` ranks x the workload's reads from a genome `world` times larger, bucket 0 kept, chunk by chunk.
#   python tools/shard_sim_big.py <world> <workload> [reads_per_chain]
import sys, time, torch
sys.path.insert(0, '.')
import harc_amd, bench
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
wl = sys.argv[2] if len(sys.argv) > 2 else "c3"
rpc = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
n, L, G, err, _ = bench.WORKLOADS[wl]
dev = torch.device("cuda", 0)
h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=8))
Wd = (2 * L + 63) // 64
parts, tot = [], 0
t0 = time.time()
for r in range(world):
    for reads in bench.synth_chunks(n, L, G * world, err, 1000 + r, dev):
        clean = reads[~(reads == ord("N")).any(1)].contiguous()
        torch.cuda.synchronize()
        packed = torch.empty((clean.shape[0], Wd), dtype=torch.int64, device=dev)
        h.pack_reads_device(clean.data_ptr(), clean.shape[0], L, packed.data_ptr())
        b = torch.empty((clean.shape[0],), dtype=torch.int32, device=dev)
        h.bucket_reads_device(packed.data_ptr(), clean.shape[0], world, b.data_ptr())
        parts.append(packed[b == 0].clone()); tot += clean.shape[0]
        del reads, clean, packed, b
    print(f"rank {r}: {tot} reads made, {sum(p.shape[0] for p in parts)} kept, {time.time()-t0:.0f}s", flush=True)
shard = torch.cat(parts).contiguous(); del parts
torch.cuda.empty_cache()
h.close()
print("shard reads", shard.shape[0], flush=True)
# ONE context, as a rank of bench.py keeps: the first pass grows its device pool (tens of GB of hipMalloc inside the run: 0.1 ... 3 s on this box,
# which is what the "index_ms = 2995" outlier of round 2 and its cousins were -- a fresh context per iteration paid it every time), the timed ones reuse it
K, r_ = 0, rpc
h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=8, num_chains=K, reads_per_chain=r_))
torch.cuda.synchronize()
h.set_reads_packed_device(shard.data_ptr(), shard.shape[0])
for it in ("warm-up", 0, 1):
    torch.cuda.synchronize(); t0 = time.time()
    h.reorder(); h.encode()
    torch.cuda.synchronize(); dt = time.time() - t0
    c = h.counters()
    print(f"iter {it} reads_per_chain {r_}: {dt*1e3:.1f} ms -> {shard.shape[0]/dt/1e6:.1f} Mreads/s/GPU rounds={c.rounds} unmatched={c.unmatched} contigs={c.contigs} seq_bases={c.seq_bases} K={c.chains} "
          f"lookups/read={c.useful_probes/max(1,c.n_clean):.1f} index_ms={c.index_ms:.1f} chain_ms={c.chain_ms:.1f} encode_ms={c.encode_ms:.1f}", flush=True)
h.close()
