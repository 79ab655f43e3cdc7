#!/bin/bash
# row sums of the CG start vector inside the list sweep (default) or as one more matrix pass (RXMD_QEQ_NO_PREPASS=1)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2; do for p in 0 1; do
  if [ $p = 1 ]; then export RXMD_QEQ_NO_PREPASS=1; else unset RXMD_QEQ_NO_PREPASS; fi
  python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; k={x['name']:x['ms'] for x in d['roofline']['kernels']}
print('no_prepass=$p', 'ms/step', round(d['ms_per_step'],2), 'pass', round(d['roofline']['avg_launch_ms'],4), 'passes/step', d['roofline']['spmv_launches_per_step'], 'iters', round(d['qeq_iters_per_step'],1), 'lists', round(b['ms_lists'],2), 'k_list10', round(k['k_list10'],3))"
done; done
