run() { python bench.py --workload $1 --steps 2 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['phases_ms_last_step'], d['roofline']['avg_launch_us'], d['counters_last_step']['rounds'], d['counters_last_step']['conflicts'], d['roundtrip']['ok'], d['counters_last_step']['seq_bases'])"; }
for b in 4 8 16 32 64; do HARC_AMD_BUDGET=$b run c2r c2r_b$b; done
for b in 4 8 16 32 64; do HARC_AMD_BUDGET=$b run c2d c2d_b$b; done
