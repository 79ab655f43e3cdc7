#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_multirank.py -q -x -k "36k_against or 36k_through or self_loop or bench_launches" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for rep in 1 2; do
python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs --no-steady > $O/bench$rep.log 2>&1; grep '^{"metric' $O/bench$rep.log > $O/bench$rep.json
python3 -c "
import json; d=json.load(open('$O/bench$rep.json')); print('ms/step', round(d['ms_per_step'],2), 'pass', round(d['roofline']['avg_launch_ms'],4), 'iters', d['qeq_iters_per_step'], 'qeq/iter', round(d['ms_qeq_per_iter'],4)); b=d['breakdown_ms_per_step']; print({k: round(v,2) for k,v in b.items() if v}); print([(k['name'][:12], round(k['ms'],3)) for k in d['roofline']['kernels']])"
done
