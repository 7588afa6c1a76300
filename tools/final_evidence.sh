#!/bin/bash
# The evidence of a round's LAST build in one gpurun call (copy gpurun_out/<tag>/ into profiles/<round>/ afterwards, and merge
# gpurun_out/<tag>/k_steps_traffic_c3.json into profiles/k_steps_traffic.json as this script does on the box):
#   PMC passes of the default command FIRST, so that the bench lines that follow can say `traffic_stale: false` for their own library; the default line
#   twice (as the driver runs it, and with 20 timed steps); kernel tables of configs[2] / [3] / [4]'s share; the repeat workloads; the small ones; the GPU suite.
#   tools/final_evidence.sh <tag> a|b      (two calls: a gpurun call lasts at most 20 minutes; a = counters, default lines, kernel tables; b = the other workloads, the suite)
R=${1:-r06f}; PART=${2:-a}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
cd "$ROOT"; mkdir -p gpurun_out/$R
if [ "$PART" = a ]; then
bash tools/pmc.sh $R c3 2 > gpurun_out/$R/pmc_c3.log 2>&1
python - $R <<PY
import json,sys
p="profiles/k_steps_traffic.json"; d=json.load(open(p)); n=json.load(open("gpurun_out/%s/k_steps_traffic_c3.json" % sys.argv[1]))
if n["c3"]["fetch_kb_per_launch"] > 0: d["c3"]=n["c3"]; json.dump(d, open(p,"w"), indent=1)
PY
echo "[final_evidence] pmc done"
python bench.py > gpurun_out/$R/bench_c3_default_flags.json 2> gpurun_out/$R/bench_c3_default_flags.err
echo "[final_evidence] default line done"
python bench.py --steps 20 --warmup 5 > gpurun_out/$R/bench_c3.json 2> gpurun_out/$R/bench_c3.err
bash tools/kstats.sh c3 $R 3 > /dev/null 2>&1
echo "[final_evidence] c3 done"
bash tools/kstats.sh c4 $R 2 > /dev/null 2>&1
bash tools/kstats.sh c5g $R 2 > /dev/null 2>&1
echo "[final_evidence] kernel tables done"
else
for w in c1 c2 c3s c4s c2r c2d; do python bench.py --workload $w --steps 5 --warmup 1 --no-cpu > gpurun_out/$R/bench_$w.json 2> /dev/null; done
for w in c3sd c3r; do python bench.py --workload $w --steps 2 --warmup 1 --no-cpu > gpurun_out/$R/bench_$w.json 2> /dev/null; done
echo "[final_evidence] small and repeat workloads done"
python bench.py --workload c4r --steps 1 --warmup 1 --no-cpu > gpurun_out/$R/bench_c4r.json 2> /dev/null
echo "[final_evidence] c4r done"
( time timeout -k 10 900 python -m pytest tests -q -m gpu ) > gpurun_out/$R/pytest_gpu_final.log 2>&1
tail -4 gpurun_out/$R/pytest_gpu_final.log
fi
for f in gpurun_out/$R/bench_*.json; do python - "$f" <<PY
import json,sys
try:
    d=json.load(open(sys.argv[1])); print(sys.argv[1].split('/')[-1], d['value'], d['ms_per_step'], d['phases_ms_last_step'], d['roundtrip']['ok'], d['roofline'].get('avg_launch_us'), d['roofline'].get('traffic_stale'), d['counters_last_step']['device_bytes_peak'])
except Exception as e: print(sys.argv[1], 'failed', e)
PY
done
