#!/bin/bash
# the round's final bench lines (after tools/r04_ad.sh put this build's PMC numbers into profiles/k_steps_traffic.json)
R=r04ae
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
python bench.py --steps 20 --warmup 5 > gpurun_out/$R/bench_c3.json 2> gpurun_out/$R/bench_c3.err
python bench.py --workload c2 --steps 20 --warmup 5 > gpurun_out/$R/bench_c2.json 2> /dev/null
for w in c1 c3s c4s c2r c2d; do python bench.py --workload $w --steps 5 --warmup 1 --no-cpu > gpurun_out/$R/bench_$w.json 2> /dev/null; done
python bench.py --workload c3sd --steps 2 --warmup 1 --no-cpu > gpurun_out/$R/bench_c3sd.json 2> /dev/null
python bench.py --workload c2 --steps 5 --warmup 1 --force-dist > gpurun_out/$R/bench_c2_forcedist.json 2> /dev/null
python bench.py --workload c3s --steps 3 --warmup 1 --force-dist --mg-mode replicate > gpurun_out/$R/bench_c3s_replicate_w1.json 2> /dev/null
python bench.py --workload c4 --steps 2 --warmup 1 --no-cpu > gpurun_out/$R/bench_c4_full.json 2> /dev/null
python bench.py --workload c5s --steps 2 --warmup 1 --no-cpu > gpurun_out/$R/bench_c5s_full.json 2> /dev/null
python bench.py --workload c5g --steps 1 --warmup 1 --no-cpu > gpurun_out/$R/bench_c5g.json 2> /dev/null
for f in gpurun_out/$R/bench_*.json; do python - "$f" <<PY
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['phases_ms_last_step'], d['roundtrip']['ok'], d['roofline'].get('avg_launch_us'), d['counters_last_step']['device_bytes_peak'], d['roofline'].get('traffic_stale'))
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
