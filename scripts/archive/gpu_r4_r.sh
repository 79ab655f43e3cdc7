#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py tests/test_gpu_multirank.py -q -x -k "step0 or 36k_against or 36k_through or self_loop or md_trajectory or one_pass or window_pass_and or benchmark_tol or vprocs_parity" > $O/pytest.log 2>&1; grep -n "passed\|failed" $O/pytest.log | tail -2
python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs --no-steady > $O/bench.log 2>&1; grep '^{"metric' $O/bench.log > $O/bench.json
python3 -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step', round(d['ms_per_step'],2), 'pass', round(d['roofline']['avg_launch_ms'],4), 'iters', d['qeq_iters_per_step'], 'qeq/iter', round(d['ms_qeq_per_iter'],4)); b=d['breakdown_ms_per_step']; print({k: round(v,2) for k,v in b.items() if v}); print([(k['name'][:12], round(k['ms'],3)) for k in d['roofline']['kernels']])"
