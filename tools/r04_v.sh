#!/bin/bash
R=r04v
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_replicate.py tests/test_gpu_config_size.py -m gpu -x -q ) > gpurun_out/$R/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/$R/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c3 4 "-"
bash tools/ab.sh $R c4 2 "-"
bash tools/ab.sh $R c5g 2 "-"
bash tools/ab.sh $R c2r 5 "-"
bash tools/pmc_any.sh $R c3 wr "WRITE_SIZE" | tail -3
bash tools/pmc_any.sh $R c3 fe "FETCH_SIZE" | tail -3
