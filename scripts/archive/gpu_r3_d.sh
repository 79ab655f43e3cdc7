#!/bin/bash
# round 3: ring matrix pass -- bench variants by environment knobs.  usage: gpu_r3_d.sh <tag> "<env settings>" ...
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
i=0
for v in "$@"; do
i=$((i+1))
env $v timeout 600 python3 bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline 2>$O/bench_$i.err | grep '^{"metric' > $O/bench_$i.json
python3 -c "
import json,sys; d=json.load(open('$O/bench_$i.json')); b=d['breakdown_ms_per_step']
print('$v', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3), 'iters', round(d['qeq_iters_per_step'],1))" || tail -3 $O/bench_$i.err
done
