#!/bin/bash
# rocprofv3 kernel stats of one bench workload:  tools/kstats.sh <workload> [tag]   -> gpurun_out/<tag>/kstats_<workload>.txt
WL=${1:-c3}; R=${2:-r02}; STEPS=${3:-2}
mkdir -p gpurun_out/$R
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/rp_$WL -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu > gpurun_out/$R/kstats_${WL}_bench.json 2> gpurun_out/$R/kstats_$WL.err
f=$(ls gpurun_out/$R/rp_$WL/*/*kernel_stats.csv | head -1)
python3 - "$f" > gpurun_out/$R/kstats_$WL.txt <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("%-72s %8s %12s %12s %7s" % ("kernel","calls","total_ms","avg_us","pct"))
for r in rows[:45]:
    print("%-72s %8s %12.3f %12.2f %7s" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
rm -rf gpurun_out/$R/rp_$WL
head -30 gpurun_out/$R/kstats_$WL.txt
