#!/bin/bash
R=r04ac
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
for W in c4 c5g; do
echo "== $W" >> gpurun_out/$R/s2_laps.txt
HARC_AMD_TRACE=1 timeout -k 10 500 python bench.py --workload $W --steps 1 --warmup 1 --no-cpu 2>&1 > gpurun_out/$R/bench_$W.json | grep -E "^\[stage II|^\[index\]|^\[pool\]" | tail -40 >> gpurun_out/$R/s2_laps.txt
done
cat gpurun_out/$R/s2_laps.txt
