#!/bin/bash
R=r04ag
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
for i in 1 2; do for cfg in "-" "HARC_AMD_MEGA=0"; do [ "$cfg" = "-" ] && cfg=""; for w in c3 c1; do env $cfg timeout -k 10 200 python tools/exact_probe.py $w 200000 2>&1 | tail -1; done; done; done | tee gpurun_out/$R/exact.txt
