run() { echo "== $*"; env "$@" python tools_perf_probe.py $ARGS 2>&1 | grep -A1 "iter 1" | cut -c1-330; }
ARGS="20000000 100 180000000 0.0 19531 8 0 16"
run X=1
run HARC_AMD_CAPMULT=6
run HARC_AMD_CAPMULT=8
ARGS="3300000 100 6300000 0.005 1421 8 0 16"
run X=1
run HARC_AMD_CAPMULT=8
ARGS="1000000 100 35000000 0.0 0 8 0 16"
run X=1
