#!/bin/bash
# round 4, first GPU call: the whole -m gpu suite (with the new full-size tests), then kernel tables of configs[3] / configs[4]'s share,
# a PMC pass of configs[3], and the round's starting bench line
R=r04a
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( time timeout -k 10 1100 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/$R/pytest_gpu.log 2>&1; rc=$?
tail -25 gpurun_out/$R/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
echo "== kstats c4"; timeout -k 10 400 bash tools/kstats.sh c4 $R 2 > /dev/null 2>&1; head -24 gpurun_out/$R/kstats_c4.txt
echo "== kstats c5g"; timeout -k 10 400 bash tools/kstats.sh c5g $R 2 > /dev/null 2>&1; head -24 gpurun_out/$R/kstats_c5g.txt
echo "== bench c3"; timeout -k 10 300 python bench.py --steps 10 --warmup 3 > gpurun_out/$R/bench_c3.json 2> gpurun_out/$R/bench_c3.err; cut -c1-600 gpurun_out/$R/bench_c3.json
