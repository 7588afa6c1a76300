#!/bin/bash
R=r04aq
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
bash tools/pmc.sh $R c3 2 > gpurun_out/$R/pmc_c3.log 2>&1; grep -E "build_id|fetch_kb|write_kb|sq_insts_valu" gpurun_out/$R/k_steps_traffic_c3.json
bash tools/kstats.sh c3 $R 3 > /dev/null 2>&1; head -4 gpurun_out/$R/kstats_c3.txt | cut -c1-125
bash tools/kmedian.sh c4 $R 2 > /dev/null 2>&1
bash tools/kmedian.sh c5g $R 1 > /dev/null 2>&1
bash tools/kmedian.sh c3 $R 2 > /dev/null 2>&1; grep -v "at::native\|rocclr\|k_sig_ascii\|k_decode_sig" gpurun_out/$R/kmedian_c3.txt | head -30 | cut -c1-140
