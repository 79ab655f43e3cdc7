#!/bin/bash
# the default bench (CPU baseline first, as the driver runs it) with the blocking and with the polling wait
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for v in 1 0 1 0; do
RXMD_SPIN_WAIT=$v python3 bench.py --no-other-configs --no-alt 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels'][-1]
print('spin=$v', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'iters', round(d['qeq_iters_per_step'],1), 'other per iter us', round(1e3*k['ms'],1), 'cpu', round(d['cpu_baseline']['atom_steps_per_s']))"
done
