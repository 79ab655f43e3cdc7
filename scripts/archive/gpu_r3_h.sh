#!/bin/bash
# round 3: direct halo / mdmode 0,6 / CG scatter tests, scatter A/B, per-kernel PMC traffic
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_multirank.py tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -q -k "direct or self_loop or multi_rank_path or random_velocities or one_pass or tight_tolerance_parity_vs_oracle or minimis" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.log
for v in "RXMD_CG_NO_SCATTER=1" "RXMD_X=0" "RXMD_CG_NO_SCATTER=1" "RXMD_X=0"; do
env $v timeout 600 python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline --no-other-configs 2>/dev/null | grep '^{"metric' > $O/b.json
python3 -c "
import json; d=json.load(open('$O/b.json')); k=d['roofline']['kernels'][-1]
print('$v', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'iters', round(d['qeq_iters_per_step'],1), 'cg other per iter ms', round(k['ms'],4))"
done
bash scripts/gpu_pmc_kernels.sh $1/pmc
