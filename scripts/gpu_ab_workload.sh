#!/bin/bash
# A/B of two builds of the library on another workload (water | sicnp), alternating fresh processes: usage bash scripts/gpu_ab_workload.sh <workload> <reps> <tagA|default> <tagB|default> ...
cd $GRAFT_REPO_ROOT
w=$1; reps=$2; shift; shift
for rep in $(seq 1 $reps); do
for t in "$@"; do
  e="RXMD_X=0"; tag=$t
  case "$t" in *=*) e="$t"; tag=default;; esac
  if [ "$tag" = "default" ]; then unset RXMD_HIP_LIB; else export RXMD_HIP_LIB=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_$tag.so; fi
  env $e timeout -k 10 400 python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-alt --no-steady 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); b=d['breakdown_ms_per_step']; p=d['roofline']['placement_search']
print('%-26s %-6s ms/step %.2f  K %.2f  ms_qeq/iter %.4f  pass in loop %.4f  kept %.4f (first %.4f, %d draws)  frac %.3f' % ('$t', '$w', d['ms_per_step'], d['qeq_iters_per_step'], d['ms_qeq_per_iter'], d['roofline']['avg_launch_ms'], p['pass_ms_kept_placement'], p['pass_ms_first_placement'], p['draws'], d['roofline']['frac']))"
done
done
