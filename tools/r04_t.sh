#!/bin/bash
R=r04t
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_replicate.py tests/test_gpu_multigpu.py tests/test_gpu_cli.py -m gpu -x -q ) > gpurun_out/$R/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/$R/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c4 2 "-"
bash tools/ab.sh $R c5g 2 "-"
bash tools/ab.sh $R c3 3 "-"
bash tools/ab.sh $R c2 20 "-" "HARC_AMD_LIB=$PWD/harc_amd/libharc_amd_r04base.so"
