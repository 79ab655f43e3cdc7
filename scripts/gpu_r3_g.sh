#!/bin/bash
# round 3: wave-per-row DMA pass -- correctness on small systems, then bench variants
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
RXMD_SPMV_DMA=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tight_tolerance_parity_vs_oracle or one_pass or pqeq_step0" > $O/pytest.log 2>&1
echo "pytest(DMA pass) rc=$?"; tail -3 $O/pytest.log
i=0
for v in "$@"; do
i=$((i+1))
env $v timeout 600 python3 bench.py --steps 10 --warmup 3 --no-alt --no-cpu-baseline 2>$O/bench_$i.err | grep '^{"metric' > $O/bench_$i.json
python3 -c "
import json,sys; d=json.load(open('$O/bench_$i.json')); b=d['breakdown_ms_per_step']
print('$v', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3), 'iters', round(d['qeq_iters_per_step'],1))" || tail -3 $O/bench_$i.err
done
