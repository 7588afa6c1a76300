#!/bin/bash
# exact mode: kernel table of one run on 200 k reads
R=r04y
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p "$ROOT/gpurun_out/$R"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/rp -- python3 tools/exact_probe.py c3 200000 > gpurun_out/$R/probe.txt 2> gpurun_out/$R/probe.err
f=$(ls gpurun_out/$R/rp/*/*kernel_stats.csv | head -1)
python3 - "$f" > gpurun_out/$R/kstats_exact.txt <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("%-72s %8s %12s %12s %7s" % ("kernel","calls","total_ms","avg_us","pct"))
for r in rows[:30]:
    print("%-72s %8s %12.3f %12.2f %7s" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
rm -rf gpurun_out/$R/rp
tail -1 gpurun_out/$R/probe.txt; head -14 gpurun_out/$R/kstats_exact.txt | cut -c1-125
HARC_AMD_TRACE=1 timeout 100 python tools/exact_probe.py c3 200000 2>&1 | grep -v "^\[stage I\] round" | tail -30
