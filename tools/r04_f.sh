#!/bin/bash
R=r04f
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
bash tools/ab.sh $R c3 4 "-" "HARC_AMD_PREFETCH=0"
bash tools/ab.sh $R c4 2 "-" "HARC_AMD_PREFETCH=0"
