#!/bin/bash
# exact mode at 5 M reads: with and without the successor lists, same stream digest
R=r04ay
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
timeout 900 python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r04ay/exact_5M.txt
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench, harc_amd
dev = torch.device("cuda", 0)
n, L = 5_000_000, 100
for err in (0.0, 0.005):
    reads = bench.synth_reads(n, L, int(n * L / 11.3), err, 77, dev)
    out = {}
    for tag, env in (("lists", None), ("no lists", "0")):
        if env is None: os.environ.pop("HARC_AMD_SUCC", None)
        else: os.environ["HARC_AMD_SUCC"] = env
        h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=1, num_chains=1, stream_digest=1))
        hasN = (reads == ord("N")).any(1)
        cl, wn = reads[~hasN].contiguous(), reads[hasN].contiguous()
        torch.cuda.synchronize()
        h.set_reads_ascii_device(cl.data_ptr(), cl.shape[0], L); h.set_nreads_ascii_device(wn.data_ptr(), wn.shape[0], L)
        t0 = time.perf_counter(); h.reorder(); h.encode(); dt = time.perf_counter() - t0
        c = h.counters(); out[tag] = (h.stream_digest(), c.unmatched, c.contigs, c.rounds)
        print("err %.3f %-9s %.3f Mreads/s  rounds %d  contigs %d  unmatched %d  peak %.2f GB" % (err, tag, n / dt / 1e6, c.rounds, c.contigs, c.unmatched, c.device_bytes_peak / 1e9), flush=True)
        h.close()
    print("   same digest and counters:", out["lists"][:3] == out["no lists"][:3])
PY
