#!/bin/bash
# The round's profile artefacts (run through gpurun; copy gpurun_out/<tag>/ into profiles/<round>/):
#   default bench line (configs[2]) with all legs; rocprofv3 kernel stats and PMC passes of the same command for c3, kernel stats for configs[3] and
#   configs[4]'s share; bench lines of the other workloads; the N>1 paths on one GPU (--force-dist, bucket and replicate); one rank's share of a
#   partitioned stage II (tools/s2_share.sh)
R=${1:-r04final}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"; mkdir -p gpurun_out/$R
python bench.py --steps 20 --warmup 5 > gpurun_out/$R/bench_c3.json 2> gpurun_out/$R/bench_c3.err
bash tools/kstats.sh c3 $R 3 > /dev/null 2>&1
bash tools/pmc.sh $R c3 2 > gpurun_out/$R/pmc_c3.log 2>&1
python bench.py --workload c2 --steps 20 --warmup 5 > gpurun_out/$R/bench_c2.json 2> /dev/null
for w in c1 c3s c4s c2r c2d; do python bench.py --workload $w --steps 5 --warmup 1 --no-cpu > gpurun_out/$R/bench_$w.json 2> /dev/null; done
python bench.py --workload c3sd --steps 2 --warmup 1 --no-cpu > gpurun_out/$R/bench_c3sd.json 2> /dev/null
python bench.py --workload c2 --steps 5 --warmup 1 --force-dist > gpurun_out/$R/bench_c2_forcedist.json 2> /dev/null
python bench.py --workload c3s --steps 3 --warmup 1 --force-dist --mg-mode replicate > gpurun_out/$R/bench_c3s_replicate_w1.json 2> /dev/null
# the full-size configurations (one GPU): configs[3], configs[4] at 1/16, and one GPU's share of configs[4]
python bench.py --workload c4 --steps 2 --warmup 1 --no-cpu > gpurun_out/$R/bench_c4_full.json 2> /dev/null
python bench.py --workload c5s --steps 2 --warmup 1 --no-cpu > gpurun_out/$R/bench_c5s_full.json 2> /dev/null
python bench.py --workload c5g --steps 1 --warmup 1 --no-cpu > gpurun_out/$R/bench_c5g.json 2> /dev/null
bash tools/kstats.sh c4 $R 2 > /dev/null 2>&1
bash tools/kstats.sh c5g $R 2 > /dev/null 2>&1
for f in gpurun_out/$R/bench_*.json; do python - "$f" <<PY
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['phases_ms_last_step'], d['roundtrip']['ok'], d['roofline'].get('avg_launch_us'), d['counters_last_step']['device_bytes_peak'])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
