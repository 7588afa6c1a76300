#!/bin/bash
R=r04af
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_replicate.py -m gpu -x -q ) > gpurun_out/$R/pytest1.log 2>&1; rc=$?
tail -3 gpurun_out/$R/pytest1.log
[ $rc -eq 0 ] || exit $rc
( timeout -k 10 700 python -m pytest tests/test_gpu_config_size.py tests/test_gpu_full_size.py -m gpu -x -q -k "fuzz or exact" ) > gpurun_out/$R/pytest2.log 2>&1; rc=$?
tail -3 gpurun_out/$R/pytest2.log
[ $rc -eq 0 ] || exit $rc
for w in c3 c2 c1; do timeout -k 10 200 python tools/exact_probe.py $w 200000 2>&1 | tail -1; done | tee gpurun_out/$R/exact.txt
