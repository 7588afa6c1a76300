#!/bin/bash
# NEEDS an experiments build (make -C harc_amd/csrc clean && make -C harc_amd/csrc EXPERIMENTS=1): a product build does not read HARC_AMD_S2_SIM
# What ONE rank of an N-rank design-(R) run computes in stage II, measured on one GPU (HARC_AMD_S2_SIM=rank/world: the partition without
# peers; the claims of the other ranks' columns are missing, so the round trip of this run fails by construction):  tools/s2_share.sh <tag> <workload> <world>
R=$1; W=$2; N=$3
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
echo "== $W, stage II on one GPU (all shards)" > gpurun_out/$R/s2_share_$W.txt
HARC_AMD_TRACE=1 timeout -k 10 400 python bench.py --workload $W --steps 1 --warmup 1 --no-cpu 2>&1 > /dev/null | grep -E "^\[stage II|^\[index\]" | tail -24 >> gpurun_out/$R/s2_share_$W.txt
for r in 0 $((N/2)) $((N-1)); do
  echo "== $W, rank $r of $N (HARC_AMD_S2_SIM)" >> gpurun_out/$R/s2_share_$W.txt
  HARC_AMD_S2_SIM=$r/$N HARC_AMD_TRACE=1 timeout -k 10 400 python bench.py --workload $W --steps 1 --warmup 1 --no-cpu 2>&1 > gpurun_out/$R/s2_share_${W}_$r.json | grep -E "^\[stage II" | tail -8 >> gpurun_out/$R/s2_share_$W.txt
  python - gpurun_out/$R/s2_share_${W}_$r.json <<PY >> gpurun_out/$R/s2_share_$W.txt
import json,sys
try:
    d=json.load(open(sys.argv[1])); print("   encode %.1f ms (whole step %.1f ms); round trip %s (expected False: one rank's share)" % (d["phases_ms_last_step"]["encode"], d["ms_per_step"], d["roundtrip"]["ok"]))
except Exception as e: print("   no result line:", e)
PY
done
cat gpurun_out/$R/s2_share_$W.txt
