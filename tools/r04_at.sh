#!/bin/bash
R=r04at
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
L=$ROOT/harc_amd/libharc_amd_exp.so; bash tools/ab.sh $R c3 3 "HARC_AMD_LIB=$L" "HARC_AMD_LIB=$L HARC_AMD_NSUGG=16" "HARC_AMD_LIB=$L HARC_AMD_NSUGG=32" "HARC_AMD_LIB=$L HARC_AMD_NSUGG=4"
for i in 1 2 3 4; do python - gpurun_out/$R/ab_c3_$i.json <<PY
import json,sys
d=json.load(open(sys.argv[1])); c=d["counters_last_step"]; print("rounds", c["rounds"], "contigs", c["contigs"], "seq_bases", c["seq_bases"], "unmatched", c["unmatched"], "conflicts", c["conflicts"])
PY
done
bash tools/ab.sh $R c4s 3 "HARC_AMD_LIB=$L" "HARC_AMD_LIB=$L HARC_AMD_NSUGG=16" "HARC_AMD_LIB=$L HARC_AMD_NSUGG=32"
