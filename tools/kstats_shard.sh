#!/bin/bash
# rocprofv3 kernel stats of the shard simulation: tools/kstats_shard.sh <world> <workload>  -> gpurun_out/kstats_shard_<workload>_w<world>.txt
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
W=${1:-8}; WL=${2:-c3}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/rp_shard -- python3 tools/shard_sim_big.py $W $WL 1024 > gpurun_out/kstats_shard_${WL}_w$W.log 2>&1
f=$(ls gpurun_out/rp_shard/*/*kernel_stats.csv | head -1)
python3 - "$f" > gpurun_out/kstats_shard_${WL}_w$W.txt <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
print("%-72s %8s %12s %12s %7s" % ("kernel","calls","total_ms","avg_us","pct"))
for r in rows[:40]:
    print("%-72s %8s %12.3f %12.2f %7s" % (r["Name"][:72], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
rm -rf gpurun_out/rp_shard
tail -2 gpurun_out/kstats_shard_${WL}_w$W.log; head -40 gpurun_out/kstats_shard_${WL}_w$W.txt
