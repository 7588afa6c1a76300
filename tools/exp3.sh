run() { python bench.py --workload $1 --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', d['value'], d['phases_ms_last_step'], d['roofline']['avg_launch_us'], d['counters_last_step']['rounds'], d['roofline']['slots_inspected_per_read'])"; }
run c2 c2
HARC_AMD_CAPMULT=8 run c2 c2cap8
HARC_AMD_CAPMULT=2 run c2 c2cap2
HARC_AMD_QUAD=0 run c2 c2pair
HARC_AMD_QUAD=0 HARC_AMD_LIB=$PWD/harc_amd/libharc_w5.so run c2 c2pair_w5
run c1 c1
HARC_AMD_CAPMULT=8 run c1 c1cap8
HARC_AMD_QUAD=0 run c1 c1pair
run c2r c2r
run c2d c2d
run c4s c4s
HARC_AMD_BATCHES=32,64 HARC_AMD_CAPMULT=8 HARC_AMD_LIB=$PWD/harc_amd/libharc_w5.so run c4s c4s_best
