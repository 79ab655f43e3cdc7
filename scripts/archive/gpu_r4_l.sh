#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -q -x > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -3
for v in "X=1" "RXMD_EHB_ONE_KERNEL=1"; do
env $v python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs --no-steady > $O/bench.log 2>&1; grep '^{"metric' $O/bench.log > $O/bench.json
python3 -c "
import json; d=json.load(open('$O/bench.json')); print('$v', 'ms/step', round(d['ms_per_step'],2), 'pass', round(d['roofline']['avg_launch_ms'],4), 'iters', d['qeq_iters_per_step']); b=d['breakdown_ms_per_step']; print({k: round(v,2) for k,v in b.items() if v}); print([(k['name'][:12], round(k['ms'],3)) for k in d['roofline']['kernels']])"
done
