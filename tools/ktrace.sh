#!/bin/bash
# per-launch durations of one kernel in launch order:  tools/ktrace.sh <workload> <kernel-substring> [tag]
WL=${1:-c3sd}; K=${2:-k_realign_big}; R=${3:-r03}
mkdir -p gpurun_out/$R
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$R/kt_$WL -- python3 bench.py --workload $WL --steps 1 --warmup 0 --no-cpu > /dev/null 2> gpurun_out/$R/ktrace_$WL.err
f=$(ls gpurun_out/$R/kt_$WL/*/*kernel_trace.csv | head -1)
python3 - "$f" "$K" > gpurun_out/$R/ktrace_${WL}.txt <<PY
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if sys.argv[2] in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
for i,r in enumerate(rows):
    print(i, r["Kernel_Name"][:40], "grid", r.get("Grid_Size_X",r.get("Grid_Size","?")), "%.3f ms" % ((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
PY
rm -rf gpurun_out/$R/kt_$WL
cat gpurun_out/$R/ktrace_${WL}.txt
