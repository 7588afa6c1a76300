#!/bin/bash
R=r04l
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_replicate.py tests/test_gpu_multigpu.py -m gpu -x -q ) > gpurun_out/$R/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/$R/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c4 2 "-"
bash tools/ab.sh $R c5g 2 "-"
bash tools/ab.sh $R c3 3 "-"
python - <<PY
import json
for w,n in (("c4",810e6),("c5g",500e6),("c3",350e6)):
    d=json.load(open("gpurun_out/$R/ab_%s_1.json" % w)); print(w, "device_bytes_peak %.1f GB = %.0f B/read" % (d["counters_last_step"]["device_bytes_peak"]/1e9, d["counters_last_step"]["device_bytes_peak"]/n))
PY
