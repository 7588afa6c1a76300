#!/bin/bash
R=r04av
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
python -c "import harc_amd; print('build', harc_amd.build_id())"
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=4 ) > gpurun_out/$R/pytest_gpu.log 2>&1; rc=$?
tail -9 gpurun_out/$R/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py > gpurun_out/$R/bench_default.json 2> gpurun_out/$R/bench_default.err; cut -c1-260 gpurun_out/$R/bench_default.json
