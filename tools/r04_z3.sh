#!/bin/bash
R=r04z3
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
for st in 64 32 16 8; do PROBE_STEPS=$st timeout -k 10 200 python tools/exact_probe.py c3 200000 2>&1 | tail -1; done | tee gpurun_out/$R/exact.txt
