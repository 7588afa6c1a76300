"""exact mode (num_chains = 1: the reference at -t 1 byte for byte) on a bounded sample of a workload: Mreads/s under the environment given
   usage: [ENV=..] [PROBE_STEPS=<steps per super-round>] python tools/exact_probe.py [workload] [reads]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch, bench, harc_amd
w = sys.argv[1] if len(sys.argv) > 1 else "c3"
ne = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
n, L, G, err, _ = bench.WORKLOADS[w]
dev = torch.device("cuda", 0)
s = bench.synth_reads(ne, L, max(4 * L, int(G * (ne / n))), err, 998, dev, None)
kw = {"num_steps": int(os.environ["PROBE_STEPS"])} if os.environ.get("PROBE_STEPS") else {}
h, c, dt = bench.gpu_run_sample(harc_amd, s, L, 0, 1, num_chains=1, **kw)
print("exact mode %s: %d reads, %.4f Mreads/s (%.2f us per read), rounds %d, env %s" % (w, ne, ne / dt / 1e6, dt / ne * 1e6, c.rounds, {k: v for k, v in os.environ.items() if k.startswith("HARC_AMD_")}))
h.close()
