#!/bin/bash
R=r04d
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
bash tools/ab.sh $R c3 4 "-" "HARC_AMD_LAZY=0"
bash tools/pmc_valu.sh $R c3 lazy
bash tools/pmc_valu.sh $R c3 eager HARC_AMD_LAZY=0
