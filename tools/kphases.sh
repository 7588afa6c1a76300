#!/bin/bash
# kernels of the LAST step of one bench workload in launch order, runs of the same kernel collapsed:
#   tools/kphases.sh <workload> [tag]   -> gpurun_out/<tag>/kphases_<workload>.txt
# columns: start (ms after the step's first kernel), launches in the run, busy ms, idle ms in front of / inside the run
WL=${1:-c3}; R=${2:-r05}
mkdir -p gpurun_out/$R
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$R/rpp_$WL -- python3 bench.py --workload $WL --steps 1 --warmup 1 --no-cpu --no-other-configs > gpurun_out/$R/kphases_${WL}_bench.json 2> gpurun_out/$R/kphases_$WL.err
f=$(ls gpurun_out/$R/rpp_$WL/*/*kernel_trace.csv | head -1)
python3 - "$f" > gpurun_out/$R/kphases_$WL.txt <<'PY'
import csv, sys
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
# the last step starts at the last k_keygen2 (the index build's first kernel)
starts = [i for i, r in enumerate(rows) if r[2].startswith("void k_keygen2") or r[2].startswith("k_keygen2")]
i0 = starts[-1] if starts else 0
rows = rows[i0:]
t0 = rows[0][0]
runs = []
for s, e, n in rows:
    n = n[:90]
    if runs and runs[-1][0] == n:
        runs[-1][2] += 1; runs[-1][3] += e - s; runs[-1][5] += max(0, s - runs[-1][4]); runs[-1][4] = e
    else:
        gap = s - (runs[-1][4] if runs else s)
        runs.append([n, s, 1, e - s, e, 0, gap])
print("%9s %6s %9s %9s %9s  %s" % ("start_ms", "n", "busy_ms", "gap_in", "gap_before", "kernel"))
ROUND = ("k_steps", "k_resolve", "k_reseed", "k_compact", "k_huge", "fillBuffer")      # the super-rounds: one line per kernel at the end
rsum = {}
for n, s, k, busy, e, gin, gb in runs:
    if any(x in n for x in ROUND):
        a = rsum.setdefault(n, [0, 0, 0]); a[0] += k; a[1] += busy; a[2] += gin + gb
        continue
    if busy + gin + gb < 20000 and k == 1: continue          # below 20 us: not listed
    print("%9.2f %6d %9.3f %9.3f %9.3f  %s" % ((s - t0) / 1e6, k, busy / 1e6, gin / 1e6, gb / 1e6, n))
print("-- the super-rounds (launches, busy ms, idle ms in front of them)")
for n, a in sorted(rsum.items(), key=lambda x: -x[1][1]): print("%6d %9.3f %9.3f  %s" % (a[0], a[1] / 1e6, a[2] / 1e6, n))
print("total %.2f ms from first to last kernel; busy %.2f ms" % ((rows[-1][1] - t0) / 1e6, sum(e - s for s, e, _ in rows) / 1e6))
PY
rm -rf gpurun_out/$R/rpp_$WL
tail -3 gpurun_out/$R/kphases_$WL.txt
