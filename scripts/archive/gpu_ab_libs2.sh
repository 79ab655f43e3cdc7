#!/bin/bash
# A/B of library builds on the default bench, alternating, in one box.  usage: gpu_ab_libs2.sh "<tags>" [reps]   (tag "base" = rxmd_amd/librxmd_hip.so)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in $(seq 1 ${2:-2}); do
for t in $1; do
  L=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_$t.so; [ "$t" = "base" ] && L=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip.so
  RXMD_HIP_LIB=$L python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('$t', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'frac', round(d['roofline']['frac'],3), 'iters', round(d['qeq_iters_per_step'],1), 'lists', round(b['ms_lists'],2), 'force', round(b['ms_force'],2))"
done
done
