#!/bin/bash
# bench lines in short form: tools/benchrep.sh <workload> [more bench.py flags]   (GPU box)
wl=$1; shift
python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'], '$*', d['value'], d['phases_ms_last_step'], d['roundtrip']['ok'], 'rounds', d['counters_last_step']['rounds'], 'seq_bases', d['counters_last_step']['seq_bases'], 'launch_us', d['roofline']['avg_launch_us'])"
