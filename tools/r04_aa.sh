#!/bin/bash
R=r04aa
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=6 ) > gpurun_out/$R/pytest_gpu.log 2>&1; rc=$?
tail -14 gpurun_out/$R/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python bench.py --steps 10 --warmup 3 > gpurun_out/$R/bench_c3.json 2> gpurun_out/$R/bench_c3.err; cut -c1-300 gpurun_out/$R/bench_c3.json; python - <<PY
import json
d=json.loads(open("gpurun_out/$R/bench_c3.json").read().strip().splitlines()[-1])
print(d["exact_mode"]); print(d["cpu_baseline"])
PY
