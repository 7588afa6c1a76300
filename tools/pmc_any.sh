#!/bin/bash
# one PMC pass with a given counter list on the chain kernel:  tools/pmc_any.sh <tag> <workload> <name> "<COUNTERS ...>" [ENV=..]...
R=$1; W=$2; NAME=$3; CNT=$4; shift 4
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"; mkdir -p gpurun_out/$R
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc $CNT --kernel-trace --output-format csv -d gpurun_out/$R/pv_$NAME -- python3 bench.py --workload $W --steps 1 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/$R/pv_$NAME.err
f=$(ls gpurun_out/$R/pv_$NAME/*/*counter_collection.csv | head -1)
python3 - "$f" "$NAME" <<PY | tee gpurun_out/$R/pv_$NAME.txt
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:56]
    if "k_steps" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in sorted(agg):
    print(sys.argv[2], k, "launches", max(cnt[(k,c)] for c in agg[k]))
    for c,v in sorted(agg[k].items()): print("    %-28s %14.1f per launch" % (c, v/cnt[(k,c)]))
PY
rm -rf gpurun_out/$R/pv_$NAME
