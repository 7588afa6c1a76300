#!/bin/bash
R=r04ap
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
bash tools/ab.sh $R c4 2 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
bash tools/ab.sh $R c3 3 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
bash tools/ab.sh $R c5g 1 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
bash tools/ab.sh $R c4s 3 "-" "HARC_AMD_LIB=$ROOT/harc_amd/libharc_amd_head.so"
