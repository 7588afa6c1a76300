#!/bin/bash
R=r04h
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( HARC_TEST_RCCL_WORLD=1 timeout -k 10 600 python -m pytest tests/test_gpu_rccl_world2.py tests/test_gpu_replicate.py -m gpu -x -q ) > gpurun_out/$R/pytest.log 2>&1; rc=$?
tail -6 gpurun_out/$R/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/s2_share.sh $R c4 8
bash tools/s2_share.sh $R c3 8
