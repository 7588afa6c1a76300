#!/bin/bash
R=r04ah
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "K_chains or low_complexity or medium_vs or stage2 or variants_same" ) > gpurun_out/$R/pytest1.log 2>&1; rc=$?
tail -3 gpurun_out/$R/pytest1.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c3 3 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_m13.so" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_m12.so"
bash tools/ab.sh $R c4s 4 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_m13.so" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_m12.so"
bash tools/ab.sh $R c4 2 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_m13.so"
