for w in c2d c2r c2 c3s; do HARC_AMD_TRACE=1 python bench.py --workload $w --steps 1 --warmup 0 --no-cpu 2>&1 >/dev/null | grep "k_steps\]" | tail -1 | sed "s/^/$w /"; done
