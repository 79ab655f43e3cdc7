#!/bin/bash
# A/B of environment switches on ONE library on one box, alternating fresh processes: usage  [LIB=exp] bash scripts/gpu_ab_env.sh <reps> "ENV_A=1" "ENV_B=1" ...   ("-" = no switch)
cd $GRAFT_REPO_ROOT
[ -n "$LIB" ] && export RXMD_HIP_LIB=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_$LIB.so
reps=$1; shift
for rep in $(seq 1 $reps); do
for v in "$@"; do
  e="$v"; [ "$v" = "-" ] && e="RXMD_X=0"
  env $e timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-alt --no-steady --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); b=d['breakdown_ms_per_step']; p=d['roofline']['placement_search']
K=d['qeq_iters_per_step']
print('%-28s ms/step %.2f  K %.2f  ms_qeq/iter %.4f  pass in loop %.4f (kept %.4f)  CG outside the pass per iter %.1f us  lists %.2f force %.2f' % ('$v', d['ms_per_step'], K, d['ms_qeq_per_iter'], d['roofline']['avg_launch_ms'], p['pass_ms_kept_placement'], 1e3 * (b['ms_qeq'] - b['ms_qeq_spmv'] - b['ms_lists']) / K, b['ms_lists'], b['ms_force']))"
done
done
