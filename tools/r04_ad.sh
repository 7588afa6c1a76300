#!/bin/bash
# final PMC passes of the round's build (the bench lines are taken in a second call, after profiles/k_steps_traffic.json has this build's numbers)
R=r04ad
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
bash tools/pmc.sh $R c3 2 > gpurun_out/$R/pmc_c3.log 2>&1; tail -5 gpurun_out/$R/pmc_c3.log; cat gpurun_out/$R/k_steps_traffic_c3.json
bash tools/kstats.sh c3 $R 3 > /dev/null 2>&1; head -12 gpurun_out/$R/kstats_c3.txt | cut -c1-125
