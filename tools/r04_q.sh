#!/bin/bash
R=r04q
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_size.py tests/test_gpu_full_size.py -m gpu -x -q ) > gpurun_out/$R/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/$R/pytest.log
[ $rc -eq 0 ] || exit $rc
for w in c3 c2 c1; do python tools/exact_probe.py $w 2> /dev/null | tail -1; done
HARC_AMD_LAZY=0 python tools/exact_probe.py c3 2> /dev/null | tail -1
bash tools/ab.sh $R c2 20 "-" "HARC_AMD_LAZY=0" "HARC_AMD_LIB=$PWD/harc_amd/libharc_amd_r04base.so"
bash tools/ab.sh $R c1 20 "-" "HARC_AMD_LAZY=0" "HARC_AMD_LIB=$PWD/harc_amd/libharc_amd_r04base.so"
bash tools/ab.sh $R c2r 5 "-" "HARC_AMD_LIB=$PWD/harc_amd/libharc_amd_r04base.so"
