#!/bin/bash
R=r04n
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_config_size.py -m gpu -x -q ) > gpurun_out/$R/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/$R/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c3 3 "-" "HARC_AMD_TABLE_FILL=0"
bash tools/ab.sh $R c4 2 "-"
bash tools/ab.sh $R c2 10 "-"
bash tools/ab.sh $R c1 10 "-"
