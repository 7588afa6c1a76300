#!/bin/bash
R=r04ai
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"; mkdir -p gpurun_out/$R
HARC_AMD_LEFT_CMP=2 timeout 300 python bench.py --workload c4s --steps 1 --warmup 0 --no-prime --no-cpu 2> gpurun_out/$R/err.txt | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['roundtrip'])"
grep "^\[left\]" gpurun_out/$R/err.txt | head
