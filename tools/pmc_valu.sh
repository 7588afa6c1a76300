#!/bin/bash
# one PMC pass (instruction counts of the chain kernel) on a workload:  tools/pmc_valu.sh <tag> <workload> <name> [ENV=..]...
R=$1; W=$2; NAME=$3; shift 3
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"; mkdir -p gpurun_out/$R
for kv in "$@"; do export "$kv"; done
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --kernel-trace --output-format csv -d gpurun_out/$R/pv_$NAME -- python3 bench.py --workload $W --steps 1 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/$R/pv_$NAME.err
f=$(ls gpurun_out/$R/pv_$NAME/*/*counter_collection.csv | head -1)
python3 - "$f" "$NAME" <<PY | tee gpurun_out/$R/pv_$NAME.txt
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:48]
    if "k_steps" not in k: continue
    agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in sorted(agg):
    print(sys.argv[2], k, " ".join("%s=%.1fM" % (c, v/cnt[(k,c)]/1e6) for c,v in sorted(agg[k].items())), "launches", max(cnt[(k,c)] for c in agg[k]))
PY
rm -rf gpurun_out/$R/pv_$NAME
