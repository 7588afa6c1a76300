#!/bin/bash
R=r04b
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_size.py tests/test_gpu_replicate.py -m gpu -x -q ) > gpurun_out/$R/pytest.log 2>&1; rc=$?
tail -5 gpurun_out/$R/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c3 4 "-" "HARC_AMD_LAZY=0"
bash tools/ab.sh $R c4 2 "-" "HARC_AMD_LAZY=0"
bash tools/ab.sh $R c2 10 "-" "HARC_AMD_LAZY=0"
