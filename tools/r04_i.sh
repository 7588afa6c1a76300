#!/bin/bash
R=r04i
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 700 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_replicate.py -m gpu -x -q ) > gpurun_out/$R/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/$R/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c4 2 "-" "HARC_AMD_CAPMULT=3" "HARC_AMD_CAPMULT=2"
bash tools/ab.sh $R c5g 2 "-"
bash tools/ab.sh $R c3 3 "-"
for w in c3 c2; do python tools/exact_probe.py $w 2> /dev/null | tail -1; HARC_AMD_S1BLOOM=0 python tools/exact_probe.py $w 2> /dev/null | tail -1; done
python - <<PY
import json
for i in (1,2,3):
    d=json.load(open("gpurun_out/$R/ab_c4_%d.json" % i)); print("c4 config", i, "device_bytes_peak %.1f GB = %.0f B/read" % (d["counters_last_step"]["device_bytes_peak"]/1e9, d["counters_last_step"]["device_bytes_peak"]/810e6))
PY
