#!/bin/bash
R=r04g
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 ) > gpurun_out/$R/pytest_gpu.log 2>&1; rc=$?
tail -22 gpurun_out/$R/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c3sd 2 "-"
bash tools/ab.sh $R c4 2 "-"
