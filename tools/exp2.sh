run() { python bench.py --workload $1 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['phases_ms_last_step'], d['roofline']['avg_launch_us'], d['counters_last_step']['rounds'], d['roofline']['slots_inspected_per_read'])"; }
export HARC_AMD_BATCHES=32,64
run c3s w4
HARC_AMD_LIB=$PWD/harc_amd/libharc_w5.so run c3s w5
HARC_AMD_LIB=$PWD/harc_amd/libharc_w6.so run c3s w6
HARC_AMD_CAPMULT=8 HARC_AMD_LIB=$PWD/harc_amd/libharc_w5.so run c3s w5cap8
HARC_AMD_CAPMULT=8 HARC_AMD_LIB=$PWD/harc_amd/libharc_w6.so run c3s w6cap8
