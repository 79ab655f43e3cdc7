#!/bin/bash
# rows of at most 384 entries: the window pass with 3 x 128 entries in flight (one trip per row): tests, then the water configuration
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_parity.py -q -m gpu -k "window or tight_tolerance or full_size_properties_water or bitwise or one_pass" 2>&1 | grep -E "passed|failed" | tail -2
for rep in 1 2; do
python3 bench.py --workload water --no-cpu-baseline --steps 8 --warmup 2 --no-alt 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']
print('water', 'ms/step', round(d['ms_per_step'],2), r['kernel'], round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), 'iters', round(d['qeq_iters_per_step'],1), 'placement', round(r['placement_search']['pass_ms_first_placement'],4), round(r['placement_search']['pass_ms_kept_placement'],4))"
done
