#!/bin/bash
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_x -- python3 tools/repeat_probe.py $1 > /dev/null 2>&1
f=$(ls gpurun_out/rp_x/*/*kernel_stats.csv | head -1)
python3 - "$f" <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:9]:
    print("%-60s %8s %12.3f %12.2f" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3))
PY
rm -rf gpurun_out/rp_x
