#!/bin/bash
# duration of every k_steps / k_resolve / k_reseed launch of the LAST step of a bench run, in launch order (the end phase of stage I):  tools/ksteps_profile.sh <workload> <tag>
WL=${1:-c3}; R=${2:-r04}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p "$ROOT/gpurun_out/$R"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/$R/rps_$WL -- python3 bench.py --workload $WL --steps 1 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/$R/ksteps_profile_$WL.err
python3 - gpurun_out/$R/rps_$WL > gpurun_out/$R/ksteps_profile_$WL.txt <<PY
import csv,sys,glob
ev=[]
for f in glob.glob(sys.argv[1]+"/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
ev.sort()
ks=[i for i,e in enumerate(ev) if "k_steps<" in e[2]]
# the last step: the last run of k_steps launches (a gap of > 20 ms between two of them separates the steps)
cut=0
for a,b in zip(ks,ks[1:]):
    if ev[b][0]-ev[a][1] > 20e6: cut=b
rows=[]; i=cut
cur=None
for e in ev[cut:]:
    if "k_steps<" in e[2]: cur=[e[0], (e[1]-e[0])/1e3, 0.0, 0.0, e[1]]; rows.append(cur)
    elif cur is not None and "k_resolve" in e[2]: cur[2]=(e[1]-e[0])/1e3
    elif cur is not None and "k_reseed" in e[2]: cur[3]=(e[1]-e[0])/1e3; cur[4]=e[1]
print("round  k_steps_us  k_resolve_us  k_reseed_us  round_us(start to start)")
for n,(r,nx) in enumerate(zip(rows, rows[1:]+[None])):
    tot=(nx[0]-r[0])/1e3 if nx else (r[4]-r[0])/1e3
    if n % 16 == 0 or n >= len(rows)-48: print("%5d %11.1f %13.1f %12.1f %10.1f" % (n, r[1], r[2], r[3], tot))
print("launches", len(rows), "sum k_steps ms %.1f" % (sum(r[1] for r in rows)/1e3), "whole chain phase ms %.1f" % ((rows[-1][4]-rows[0][0])/1e6))
PY
rm -rf gpurun_out/$R/rps_$WL
tail -60 gpurun_out/$R/ksteps_profile_$WL.txt
