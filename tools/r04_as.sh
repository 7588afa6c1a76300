#!/bin/bash
R=r04as
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
( HARC_AMD_SUCC=1 timeout -k 10 500 python tools/fuzz_parity.py 250 41 ) > gpurun_out/$R/fuzz_succ1.txt 2>&1; tail -1 gpurun_out/$R/fuzz_succ1.txt; grep -c DIFF gpurun_out/$R/fuzz_succ1.txt
( FUZZ_K=1 FUZZ_S=64 timeout -k 10 400 python tools/fuzz_parity.py 200 42 ) > gpurun_out/$R/fuzz_exact.txt 2>&1; tail -1 gpurun_out/$R/fuzz_exact.txt; grep -c DIFF gpurun_out/$R/fuzz_exact.txt
( timeout -k 10 300 python tools/fuzz_parity.py 150 43 ) > gpurun_out/$R/fuzz_default.txt 2>&1; tail -1 gpurun_out/$R/fuzz_default.txt; grep -c DIFF gpurun_out/$R/fuzz_default.txt
