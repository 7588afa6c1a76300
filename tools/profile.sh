#!/bin/bash
# Produces the round's profile artefacts under gpurun_out/ (copy the summaries into profiles/):
#   bench JSON lines for c2 (default), c3s; rocprofv3 kernel stats of the same bench command.
R=${1:-r01}
mkdir -p gpurun_out/$R
python bench.py --steps 5 --warmup 1 > gpurun_out/$R/bench_c2.json 2> gpurun_out/$R/bench_c2.err
python bench.py --workload c3s --steps 3 --warmup 1 --no-cpu > gpurun_out/$R/bench_c3s.json 2> gpurun_out/$R/bench_c3s.err
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/rocprof_c2 -- python3 bench.py --steps 5 --warmup 1 --no-cpu > gpurun_out/$R/rocprof_c2_bench.json 2> gpurun_out/$R/rocprof_c2.err
f=$(ls gpurun_out/$R/rocprof_c2/*/*kernel_stats.csv | head -1)
python3 - "$f" > gpurun_out/$R/rocprof_c2_kernel_stats.txt <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("%-72s %8s %12s %12s %7s" % ("kernel","calls","total_ms","avg_us","pct"))
for r in rows[:40]:
    print("%-72s %8s %12.3f %12.2f %7s" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
rm -rf gpurun_out/$R/rocprof_c2
cat gpurun_out/$R/bench_c2.json; cat gpurun_out/$R/bench_c3s.json | cut -c1-300; head -12 gpurun_out/$R/rocprof_c2_kernel_stats.txt
