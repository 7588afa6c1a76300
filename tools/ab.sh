#!/bin/bash
# A/B of kernel variants on one workload in ONE gpurun call:  tools/ab.sh <tag> <workload> <steps> "ENV1=.. ENV2=.." "ENV=.." ...
# each quoted argument is one configuration (environment assignments; "-" = the defaults); HARC_AMD_LIB=<path> picks an experiment build
R=$1; W=$2; STEPS=$3; shift 3
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
i=0
for cfg in "$@"; do
  i=$((i+1))
  [ "$cfg" = "-" ] && cfg=""
  env $cfg timeout -k 10 280 python bench.py --workload $W --steps $STEPS --warmup 2 --no-cpu > gpurun_out/$R/ab_${W}_$i.json 2> gpurun_out/$R/ab_${W}_$i.err || { echo "config $i ($cfg) failed"; tail -3 gpurun_out/$R/ab_${W}_$i.err; continue; }
  python - "$cfg" gpurun_out/$R/ab_${W}_$i.json <<PY
import json,sys
d=json.load(open(sys.argv[2])); r=d["roofline"]
print("%-60s %8.2f Mreads/s  %8.2f ms/step  k_steps %7.1f us x %d  phases %s  roundtrip %s" % (sys.argv[1] or "(defaults)", d["value"], d["ms_per_step"], r["avg_launch_us"], r["launches"]//d["steps"], d["phases_ms_last_step"], d["roundtrip"]["ok"]))
PY
done
