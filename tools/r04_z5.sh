#!/bin/bash
R=r04z5
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 700 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "variants_same or K_chains or low_complexity or medium_vs" ) > gpurun_out/$R/pytest1.log 2>&1; rc=$?
tail -3 gpurun_out/$R/pytest1.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c3 4 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so" "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
