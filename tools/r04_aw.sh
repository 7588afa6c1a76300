#!/bin/bash
R=r04aw
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
for e in 0 1 2 3; do
  HARC_AMD_CONS_EXP=$e bash tools/kmedian.sh c3 ${R}_$e 1 > /dev/null 2>&1
  echo "CONS_EXP=$e: $(grep k_consensus gpurun_out/${R}_$e/kmedian_c3.txt | head -1 | cut -c1-120)"
done
