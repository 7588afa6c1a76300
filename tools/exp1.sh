mkdir -p gpurun_out/r02b
tools/micro/gups > gpurun_out/r02b/gups.txt 2>&1
head -11 gpurun_out/r02b/gups.txt
run() { python bench.py --workload $1 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['phases_ms_last_step'], d['roofline']['avg_launch_us'], d['counters_last_step']['rounds'], d['roofline']['slots_inspected_per_read'])"; }
run c3s base
HARC_AMD_BATCHES=32,64 run c3s b32_64
HARC_AMD_BATCHES=32,32,64 run c3s b32_32_64
HARC_AMD_BATCHES=16,32,64 run c3s b16_32_64
HARC_AMD_BATCHES=48,64 run c3s b48_64
HARC_AMD_CAPMULT=8 HARC_AMD_BATCHES=32,64 run c3s cap8_b32_64
HARC_AMD_CAPMULT=6 run c3s cap6
run c2 c2base
HARC_AMD_BATCHES=32,64 run c2 c2_b32_64
