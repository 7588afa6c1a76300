#!/bin/bash
# PMC passes for the dominant kernel on one bench workload (separate passes: FETCH_SIZE and WRITE_SIZE do not fit together, and
# no trace domain beyond --kernel-trace is combined with --pmc):   tools/pmc.sh <tag> <workload> [steps]
#   -> gpurun_out/<tag>/pmc_<workload>_summary.txt   per-kernel averages of every counter
#   -> gpurun_out/<tag>/k_steps_traffic_<workload>.json   {"fetch_kb_per_launch", "write_kb_per_launch", ...} for profiles/k_steps_traffic.json
R=${1:-r02}; W=${2:-c3}; STEPS=${3:-2}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
mkdir -p gpurun_out/$R
: > gpurun_out/$R/pmc_${W}_summary.txt
for pmc in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"; do
  tag=$(echo $pmc | tr " " "_" | cut -c1-24)
  rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d gpurun_out/$R/pmc_$tag -- python3 bench.py --workload $W --steps $STEPS --warmup 1 --no-cpu > /dev/null 2> gpurun_out/$R/pmc_${W}_$tag.err
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
BID=$(python3 -c "import sys; sys.path.insert(0, '$ROOT'); import harc_amd; print(harc_amd.build_id())")
python3 - gpurun_out/$R/pmc_${W}_summary.txt $W $R $BID > gpurun_out/$R/k_steps_traffic_$W.json <<PY
import sys, json, re
vals = {}
for line in open(sys.argv[1]):
    if "k_steps<" not in line: continue
    m = re.search(r"(\S+)\s+launches=(\d+)\s+avg_per_launch=([0-9.]+)", line)
    if m: vals[m.group(1)] = (int(m.group(2)), float(m.group(3)))
out = {sys.argv[2]: {"profile": "profiles/%s/pmc_%s_summary.txt" % (sys.argv[3], sys.argv[2]), "build_id": sys.argv[4],
       "fetch_kb_per_launch": vals.get("FETCH_SIZE", (0, 0.0))[1], "write_kb_per_launch": vals.get("WRITE_SIZE", (0, 0.0))[1],
       "launches": vals.get("FETCH_SIZE", (0, 0.0))[0],
       "tcc_hit": vals.get("TCC_HIT_sum", (0, 0.0))[1], "tcc_miss": vals.get("TCC_MISS_sum", (0, 0.0))[1],
       "sq_wave_cycles": vals.get("SQ_WAVE_CYCLES", (0, 0.0))[1], "sq_wait_any": vals.get("SQ_WAIT_ANY", (0, 0.0))[1],
       "sq_active_inst_any": vals.get("SQ_ACTIVE_INST_ANY", (0, 0.0))[1], "sq_waves": vals.get("SQ_WAVES", (0, 0.0))[1],
       "sq_insts_valu": vals.get("SQ_INSTS_VALU", (0, 0.0))[1], "sq_insts_salu": vals.get("SQ_INSTS_SALU", (0, 0.0))[1], "sq_busy_cycles": vals.get("SQ_BUSY_CYCLES", (0, 0.0))[1]}}
print(json.dumps(out, indent=1))
PY
grep -E "k_steps<|k_resolve|k_reseed" gpurun_out/$R/pmc_${W}_summary.txt
cat gpurun_out/$R/k_steps_traffic_$W.json
