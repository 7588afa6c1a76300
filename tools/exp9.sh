#!/bin/bash
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -E "passed|failed|Error" | tail -3
timeout 900 python tools/fuzz_parity.py 40 17 2>&1 | tail -1
for wl in c3 c4s; do
python bench.py --workload $wl --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('$wl', d['value'], d['phases_ms_last_step'], d['roundtrip']['ok'], d['counters_last_step']['singletons_aligned'])"
done
HARC_AMD_BLOOM4_HASHED=1 python bench.py --workload c4s --steps 3 --warmup 1 --no-cpu 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print('c4s hashed', d['value'], d['phases_ms_last_step'], d['roundtrip']['ok'], d['counters_last_step']['singletons_aligned'])"
bash tools/kstats.sh c3 x9 2 | grep -E "k_consensus|k_realign|k_bloom4" | head
