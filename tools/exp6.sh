run() { python bench.py --workload $1 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['phases_ms_last_step'], d['roofline']['avg_launch_us'], d['counters_last_step']['rounds'], d['roofline']['slots_inspected_per_read'])"; }
run c3s base
HARC_AMD_BATCHES=16,16,32,64 run c3s b16_16_32_64
HARC_AMD_BATCHES=16,32,64 run c3s b16_32_64
HARC_AMD_BATCHES=24,40,64 run c3s b24_40_64
HARC_AMD_BATCHES=32,32,64 run c3s b32_32_64
