#!/bin/bash
R=r04w
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py tests/test_gpu_replicate.py tests/test_gpu_config_size.py -m gpu -x -q ) > gpurun_out/$R/pytest.log 2>&1; rc=$?
tail -4 gpurun_out/$R/pytest.log
[ $rc -eq 0 ] || exit $rc
bash tools/ab.sh $R c3 4 "-"
bash tools/ab.sh $R c4 2 "-"
bash tools/ab.sh $R c5g 2 "-"
bash tools/ab.sh $R c2 20 "-"
bash tools/ab.sh $R c1 20 "-"
bash tools/ab.sh $R c2r 5 "-"
bash tools/ab.sh $R c3sd 2 "-"
bash tools/kstats.sh c3 $R 2 > /dev/null 2>&1; head -12 gpurun_out/$R/kstats_c3.txt | cut -c1-125; grep -E "k_bid_insert|k_resolve|k_reseed|fillBuffer" gpurun_out/$R/kstats_c3.txt | cut -c1-125
