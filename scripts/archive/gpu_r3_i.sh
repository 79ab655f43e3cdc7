#!/bin/bash
# round 3: run-ahead CG loop -- full suite, then A/B against the blocking loop
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log
for rep in 1 2 3; do for v in 1 0; do
RXMD_CG_NO_RUNAHEAD=$v python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels'][-1]
print('no_runahead=$v', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'iters', round(d['qeq_iters_per_step'],2), 'other per iter us', round(1e3*k['ms'],1), 'PE', d['energy_per_atom']['PE'])"
done; done
