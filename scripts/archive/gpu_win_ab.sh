#!/bin/bash
# window pass on/off: parity tests with it on, then the default bench alternating
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
if [ "$1" != "noparity" ]; then timeout 1500 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -5; fi
for rep in 1 2; do
for w in 1 0; do
  RXMD_SPMV_WIN=$w python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('win=$w', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3), 'iters', round(d['qeq_iters_per_step'],1), 'lists', round(b['ms_lists'],2), 'force', round(b['ms_force'],2), 'winbuild', round(b['ms_k_winbuild'],3), 'PE/atom', d['energy_per_atom']['PE'])"
done
done
