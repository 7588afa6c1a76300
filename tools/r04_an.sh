#!/bin/bash
R=r04an
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
bash tools/ab.sh $R c4 2 "-" "HARC_AMD_COPY_CUS=0" "HARC_AMD_COPY_CUS=8" "HARC_AMD_COPY_CUS=32"
bash tools/ab.sh $R c3 3 "-" "HARC_AMD_COPY_CUS=0"
bash tools/ab.sh $R c5g 1 "-" "HARC_AMD_COPY_CUS=0"
bash tools/ab.sh $R c2 10 "-" "HARC_AMD_COPY_CUS=0"
