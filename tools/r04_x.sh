#!/bin/bash
# exact mode (K = 1) under the kernel knobs that exist
R=r04x
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
for cfg in "-" "HARC_AMD_S1BLOOM=0" "HARC_AMD_QUAD=0 HARC_AMD_DENSE=1 HARC_AMD_SEQ=1" "HARC_AMD_QUAD=0 HARC_AMD_DENSE=1 HARC_AMD_SEQ=1 HARC_AMD_S1BLOOM=0" "HARC_AMD_QUAD=0 HARC_AMD_DENSE=0" "HARC_AMD_WEEDMIN=99" "HARC_AMD_LAZY=0"; do
  [ "$cfg" = "-" ] && cfg=""
  env $cfg timeout -k 10 200 python tools/exact_probe.py c3 200000 2>&1 | tail -1
done | tee gpurun_out/$R/exact.txt
