#!/bin/bash
# PMC passes for the dominant kernel on the default bench command (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together)
R=${1:-r01}; W=${2:-c2}
mkdir -p gpurun_out/$R
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  tag=$(echo $pmc | tr " " "_" | cut -c1-24)
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d gpurun_out/$R/pmc_$tag -- python3 bench.py --workload $W --steps 2 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/$R/pmc_$tag.err
  f=$(ls gpurun_out/$R/pmc_$tag/*/*counter_collection.csv | head -1)
  python3 - "$f" >> gpurun_out/$R/pmc_${W}_summary.txt <<PY
import csv,sys,collections
agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
for r in csv.DictReader(open(sys.argv[1])):
    k=r["Kernel_Name"][:40]; agg[k][r["Counter_Name"]]+=float(r["Counter_Value"]); cnt[(k,r["Counter_Name"])]+=1
for k in sorted(agg):
    if k.startswith(("void k_", "k_")):
        for c,v in agg[k].items(): print("%-42s %-20s launches=%-6d avg_per_launch=%.1f" % (k, c, cnt[(k,c)], v/cnt[(k,c)]))
PY
  rm -rf gpurun_out/$R/pmc_$tag
done
cat gpurun_out/$R/pmc_${W}_summary.txt | grep -E "k_steps<|k_resolve|k_reseed"
