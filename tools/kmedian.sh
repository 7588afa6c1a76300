#!/bin/bash
# rocprofv3 kernel trace of one bench workload, per kernel: calls, MEDIAN and minimum duration, calls per step x median (the first, cold step of a
# context -- page faults of the growing pool -- weighs on the averages of --stats):  tools/kmedian.sh <workload> <tag> [steps]
WL=${1:-c3}; R=${2:-r04}; STEPS=${3:-2}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p "$ROOT/gpurun_out/$R"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$R/rpm_$WL -- python3 bench.py --workload $WL --steps $STEPS --warmup 1 --no-cpu > gpurun_out/$R/kmedian_${WL}_bench.json 2> gpurun_out/$R/kmedian_$WL.err
f=$(ls gpurun_out/$R/rpm_$WL/*/*kernel_trace.csv | head -1)
python3 - "$f" $((STEPS+1)) > gpurun_out/$R/kmedian_$WL.txt <<PY
import csv,sys,collections,statistics
d=collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    d[r["Kernel_Name"]].append((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3)
nsteps=int(sys.argv[2])
rows=[(len(v)/nsteps*statistics.median(v), k, len(v), statistics.median(v), min(v), max(v)) for k,v in d.items()]
rows.sort(reverse=True)
print("%-70s %7s %11s %11s %11s %14s" % ("kernel","calls","median_us","min_us","max_us","ms_per_step"))
for tot,k,n,med,mn,mx in rows[:48]:
    print("%-70s %7d %11.1f %11.1f %11.1f %14.2f" % (k[:70], n, med, mn, mx, tot/1e3))
PY
rm -rf gpurun_out/$R/rpm_$WL
grep -v "at::native\|rocclr\|k_sig_ascii\|k_decode_sig" gpurun_out/$R/kmedian_$WL.txt | head -34 | cut -c1-140
