#!/bin/bash
# self-loop multi-rank path (one rank through the staged exchange + RCCL to itself): A/B of env switches, alternating fresh processes
# usage: bash scripts/gpu_selfloop_ab.sh <reps> "ENV=1" "-" ...
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
reps=$1; shift
for rep in $(seq 1 $reps); do
for v in "$@"; do
  e="$v"; [ "$v" = "-" ] && e="RXMD_X=0"
  env $e RXMD_BENCH_FORCE_DIST=1 RXMD_FORCE_STAGED=1 RXMD_FORCE_REMOTE=1 timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-other-configs --no-steady 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); b=d['breakdown_ms_per_step']; K=d['qeq_iters_per_step']
print('%-26s self loop ms/step %.2f  K %.2f  pass %.4f  CG per iter %.4f  lists %.2f force %.2f  halo %.2f exposed %.2f allreduce %.2f' % ('$v', d['ms_per_step'], K, d['roofline']['avg_launch_ms'], (b['ms_qeq'] - b['ms_lists']) / K, b['ms_lists'], b['ms_force'], b['ms_halo'], b['ms_halo_exposed'], b['ms_allreduce']))"
done
done
