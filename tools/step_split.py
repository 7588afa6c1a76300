"""wall time of reorder() and encode() apart, against the library's own phase counters:  python tools/step_split.py <workload> [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench, harc_amd
w = sys.argv[1] if len(sys.argv) > 1 else "c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n, L, G, err, _ = bench.WORKLOADS[w]
dev = torch.device("cuda", 0)
h = harc_amd.HarcAmd(harc_amd.default_params(L, num_thr=8, num_chains=0, device=0, profile=1))
bench.install_synthetic(h, n, L, G, err, 1000, dev, bench.SPIKES.get(w))
for _ in range(3):
    h.reorder(); h.encode()
tr = te = 0.0; ph = [0.0, 0.0, 0.0]
for _ in range(steps):
    torch.cuda.synchronize(); t0 = time.perf_counter(); h.reorder(); t1 = time.perf_counter(); h.encode(); t2 = time.perf_counter()
    tr += t1 - t0; te += t2 - t1
    c = h.counters(); ph[0] += c.index_ms; ph[1] += c.chain_ms; ph[2] += c.encode_ms
print("%s: reorder() %.2f ms (index %.2f + chain %.2f inside), encode() %.2f ms (counter %.2f)" % (w, tr / steps * 1e3, ph[0] / steps, ph[1] / steps, te / steps * 1e3, ph[2] / steps))
h.close()
