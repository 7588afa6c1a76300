#!/bin/bash
R=r04e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_config_size.py -m gpu -x -q -k "variants or config_size_matches" ) > gpurun_out/$R/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/$R/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c3 4 "-" "HARC_AMD_SPEC=0" "HARC_AMD_SEQ=1" "HARC_AMD_SEQ=1 HARC_AMD_SPEC=0"
bash tools/ab.sh $R c4 2 "-" "HARC_AMD_SPEC=0"
bash tools/pmc_valu.sh $R c3 spec
