#!/bin/bash
R=r04aj
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_size.py -m gpu -x -q ) > gpurun_out/$R/pytest1.log 2>&1; rc=$?
tail -3 gpurun_out/$R/pytest1.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c4s 3 "-" "HARC_AMD_LEFT_ALL=1"
bash tools/ab.sh $R c4 2 "-" "HARC_AMD_LEFT_ALL=1"
bash tools/ab.sh $R c5g 1 "-"
bash tools/ab.sh $R c3 3 "-"
