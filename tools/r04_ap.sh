#!/bin/bash
R=r04ap
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "stage2 or K_chains or medium_vs or low_complexity or variants_same or decoder or preserve" ) > gpurun_out/$R/pytest1.log 2>&1; rc=$?
tail -3 gpurun_out/$R/pytest1.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c4 2 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so" "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
bash tools/ab.sh $R c5g 1 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so" "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
bash tools/ab.sh $R c4s 3 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
