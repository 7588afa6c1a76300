#!/bin/bash
R=r04ax
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replicate.py tests/test_gpu_config_size.py -m gpu -x -q ) > gpurun_out/$R/pytest1.log 2>&1; rc=$?
tail -3 gpurun_out/$R/pytest1.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c3 3 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so" "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
bash tools/ab.sh $R c4 2 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
bash tools/ab.sh $R c2 20 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
bash tools/ab.sh $R c2r 5 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
