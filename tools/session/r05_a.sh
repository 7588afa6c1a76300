#!/bin/bash
# round 5, call A: the random-access ceiling by two tools in the shape of k_steps' misses + what an L2 miss of that shape fetches (PMC) + today's baseline of c3
R=r05a
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
timeout -k 10 120 tools/micro/gups sector 64 > gpurun_out/$R/gups_sector.txt 2>&1 && cat gpurun_out/$R/gups_sector.txt
timeout -k 10 120 tools/micro/lat2 indep 64 > gpurun_out/$R/lat2_indep.txt 2>&1 && cat gpurun_out/$R/lat2_indep.txt
timeout -k 10 200 tools/micro/gups > gpurun_out/$R/gups_all.txt 2>&1; tail -12 gpurun_out/$R/gups_all.txt
for pmc in "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "FETCH_SIZE" "TCC_MISS_sum TCC_HIT_sum"; do
  tag=$(echo $pmc | tr " " "_" | cut -c1-24)
  timeout -k 10 200 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d gpurun_out/$R/pmc_$tag -- tools/micro/gups sector 64 > /dev/null 2> gpurun_out/$R/pmc_gups_$tag.err
  f=$(ls gpurun_out/$R/pmc_$tag/*/*counter_collection.csv | head -1)
  python3 - "$f" <<PY | tee -a gpurun_out/$R/pmc_gups_sector.txt
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    if "k_sector" in r["Kernel_Name"]: print(r["Kernel_Name"][:48], r.get("Dispatch_Id"), r["Counter_Name"], r["Counter_Value"])
PY
  rm -rf gpurun_out/$R/pmc_$tag
done
timeout -k 10 400 python bench.py --steps 5 --warmup 2 --no-cpu > gpurun_out/$R/bench_c3.json 2> gpurun_out/$R/bench_c3.err; cut -c1-600 gpurun_out/$R/bench_c3.json
