#!/bin/bash
# the dispatches around one kernel in the LAST step of a bench run, with start / end relative to that kernel's start:  tools/ktimeline.sh <workload> <tag> <kernel substring> [before] [after]
WL=${1:-c4}; R=${2:-r04}; PAT=${3:-k_acc_flags}; NB=${4:-12}; NA=${5:-12}
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; mkdir -p "$ROOT/gpurun_out/$R"
cd /tmp && export TMPDIR=/tmp; cd "$ROOT"
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d gpurun_out/$R/rpt_$WL -- python3 bench.py --workload $WL --steps 1 --warmup 1 --no-cpu > /dev/null 2> gpurun_out/$R/ktimeline_$WL.err
python3 - gpurun_out/$R/rpt_$WL "$PAT" $NB $NA > gpurun_out/$R/ktimeline_$WL.txt <<PY
import csv,sys,glob
d=sys.argv[1]; pat=sys.argv[2]; nb=int(sys.argv[3]); na=int(sys.argv[4])
ev=[]
for f in glob.glob(d+"/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "K q%s %s" % (r.get("Queue_Id","?"), r["Kernel_Name"][:60])))
for f in glob.glob(d+"/*/*memory_copy_trace.csv"):
    for r in csv.DictReader(open(f)): ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), "C %s %s bytes" % (r.get("Direction","?"), r.get("Bytes", r.get("Size","?")))))
ev.sort()
idx=[i for i,e in enumerate(ev) if pat in e[2]]
if not idx: print("no such kernel"); sys.exit()
i0=idx[-1]; t0=ev[i0][0]
for e in ev[max(0,i0-nb):i0+na+1]:
    print("%10.3f ms .. %10.3f ms  (%9.3f ms)  %s" % ((e[0]-t0)/1e6, (e[1]-t0)/1e6, (e[1]-e[0])/1e6, e[2]))
PY
rm -rf gpurun_out/$R/rpt_$WL
cat gpurun_out/$R/ktimeline_$WL.txt | cut -c1-150
