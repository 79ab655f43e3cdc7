#!/bin/bash
# blocking against polling wait for Est in the CG loop, alternating on one box; prints the HIP runtime of the box (the pool is not uniform)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
python3 -c "import torch; print('torch hip', torch.version.hip)"; /opt/rocm/bin/hipconfig --version 2>/dev/null
for rep in 1 2 3; do for v in 0 1; do
RXMD_SPIN_WAIT=$v python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels'][-1]
print('spin=$v', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'iters', round(d['qeq_iters_per_step'],1), 'other per iter us', round(1e3*k['ms'],1))"
done; done
