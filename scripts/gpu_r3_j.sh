#!/bin/bash
# round 3: two rows per wavefront side by side (k_spmv2): parity, then A/B over builds with 8 / 6 / 4 batches of 32 per trip
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
RXMD_SPMV2=1 timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x -k "tight_tolerance_parity_vs_oracle or one_pass or pqeq_step0 or perturbed_rdx_36k_against or multi_rank_path" 2>&1 | tail -2
for rep in 1 2; do
for t in "base 0" "base 1" "u6 1" "u4 1"; do set -- $t
  L=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_$1.so; [ "$1" = "base" ] && L=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip.so
  RXMD_SPMV2=$2 RXMD_HIP_LIB=$L python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$1 spmv2=$2', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3), 'iters', round(d['qeq_iters_per_step'],1))"
done; done
