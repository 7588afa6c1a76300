#!/bin/bash
R=r04ao
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( time timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=6 ) > gpurun_out/$R/pytest_gpu.log 2>&1; rc=$?
tail -12 gpurun_out/$R/pytest_gpu.log
[ $rc -eq 0 ] || exit $rc
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
bash tools/pmc.sh $R c3 2 > gpurun_out/$R/pmc_c3.log 2>&1; grep -E "build_id|fetch_kb|write_kb|sq_insts_valu" gpurun_out/$R/k_steps_traffic_c3.json
bash tools/kstats.sh c3 $R 3 > /dev/null 2>&1; head -6 gpurun_out/$R/kstats_c3.txt | cut -c1-125
