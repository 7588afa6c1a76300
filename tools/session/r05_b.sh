#!/bin/bash
# round 5, call B: k_steps_grp -- parity first (variants + config size), then speed against the default on c3
R=r05b
ROOT="$(cd "$(dirname "$0")/../.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "test_kernel_variants_same_bytes" ) > gpurun_out/$R/pytest_grp.log 2>&1; rc=$?
tail -5 gpurun_out/$R/pytest_grp.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c3 3 "$@"
