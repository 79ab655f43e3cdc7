#!/bin/bash
# round 3: ring matrix pass -- parity with the ring forced on every system size, then the bench with and without it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_RING_MIN_ROWS=0 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x -k "tight_tolerance_parity_vs_oracle or one_pass or perturbed or pqeq_step0 or trajectory_tight" > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
tail -5 $O/pytest.log; cat $O/rc.txt
for ring in 1 0; do
RXMD_SPMV_RING=$ring timeout 600 python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline 2>$O/bench_$ring.err | grep '^{"metric' > $O/bench_$ring.json
python3 -c "
import json,sys; d=json.load(open('$O/bench_$ring.json')); b=d['breakdown_ms_per_step']
print('ring=$ring', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3), 'iters', round(d['qeq_iters_per_step'],1))"
done
