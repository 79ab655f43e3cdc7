#!/bin/bash
# A/B of two builds of the library on one box, alternating fresh processes: pass time on the kept placement (back to back) and inside the CG loop, ms per step
# usage: bash scripts/gpu_ab_pass.sh <tagA|default> <tagB|default> [reps]
cd $GRAFT_REPO_ROOT
for rep in $(seq 1 ${3:-3}); do
for t in $1 $2; do
  if [ "$t" = "default" ]; then unset RXMD_HIP_LIB; else export RXMD_HIP_LIB=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_$t.so; fi
  timeout -k 10 300 python3 bench.py --no-cpu-baseline --no-alt --no-steady --no-other-configs 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); b=d['breakdown_ms_per_step']; p=d['roofline']['placement_search']
print('%-10s ms/step %.2f  pass in the loop %.4f  kept placement %.4f (first %.4f, %d draws)  loop - kept %.4f  non-pass %.2f  lists %.2f force %.2f' % ('$t', d['ms_per_step'], d['roofline']['avg_launch_ms'], p['pass_ms_kept_placement'], p['pass_ms_first_placement'], p['draws'], d['roofline']['avg_launch_ms'] - p['pass_ms_kept_placement'], d['ms_per_step'] - b['ms_qeq_spmv'], b['ms_lists'], b['ms_force']))"
done
done
