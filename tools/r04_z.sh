#!/bin/bash
R=r04z
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "K1_matches or variants_same or K_chains or low_complexity or medium_vs" ) > gpurun_out/$R/pytest1.log 2>&1; rc=$?
tail -5 gpurun_out/$R/pytest1.log
[ $rc -eq 0 ] || exit $rc
( timeout -k 10 700 python -m pytest tests/test_gpu_config_size.py tests/test_gpu_full_size.py -m gpu -x -q -k "fuzz or exact" ) > gpurun_out/$R/pytest2.log 2>&1; rc=$?
tail -5 gpurun_out/$R/pytest2.log
[ $rc -eq 0 ] || exit $rc
for cfg in "-" "HARC_AMD_FUSED_RR=0"; do
  [ "$cfg" = "-" ] && cfg=""
  env $cfg timeout -k 10 200 python tools/exact_probe.py c3 200000 2>&1 | tail -1
  env $cfg timeout -k 10 200 python tools/exact_probe.py c2 200000 2>&1 | tail -1
  env $cfg timeout -k 10 200 python tools/exact_probe.py c1 200000 2>&1 | tail -1
done | tee gpurun_out/$R/exact.txt
