for w in ${1:-c2d c2r}; do HARC_AMD_TRACE=1 python bench.py --workload $w --steps 1 --warmup 0 --no-cpu 2>&1 >/dev/null | grep "k_steps\]" | tail -2 | sed "s/^/$w /"; done
