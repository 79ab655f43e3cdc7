#!/bin/bash
# The multi-rank code path of bench.py on ONE GPU at 979,776 atoms: a single rank through the staged six-stage exchange, every message through
# RCCL send/recv to itself, all-reduces on the device -- staged / direct vector halo x overlap on / off -- next to the single-rank fast path.
# One JSON line per variant (with per_rank exchange timers) -> gpurun_out/<tag>/selfloop_<variant>.json
# usage: bash scripts/gpu_selfloop_bench.sh <tag> [steps]
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-selfloop}; S=${2:-10}; mkdir -p $O
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
run() { tag=$1; shift
  env "$@" timeout -k 10 600 python3 bench.py --gpus 1 --steps $S --warmup 3 --no-cpu-baseline --no-alt --no-other-configs --no-steady 2>$O/selfloop_$tag.err | grep '^{"metric' > $O/selfloop_$tag.json
  python3 -c "
import json; d=json.load(open('$O/selfloop_$tag.json')); p=d.get('per_rank',{}); b=d['breakdown_ms_per_step']
print('%-24s ms/step %.2f  iters %.1f  pass %.4f  lists %.2f force %.2f  ghost %.2f migrate %.2f halo %.2f exposed %.2f allreduce %.2f fold %.2f' % ('$tag', d['ms_per_step'], d['qeq_iters_per_step'], d['roofline']['avg_launch_ms'], b['ms_lists'], b['ms_force'], b['ms_ghost_build'], b['ms_migrate'], b['ms_halo'], b['ms_halo_exposed'], b['ms_allreduce'], b['ms_fold']))" || tail -3 $O/selfloop_$tag.err; }
run single_rank RXMD_X=0
SL="RXMD_BENCH_FORCE_DIST=1 RXMD_FORCE_STAGED=1 RXMD_FORCE_REMOTE=1"
run staged_overlap $SL
run staged_nooverlap $SL RXMD_NO_HALO_OVERLAP=1
run direct_overlap $SL RXMD_HALO_DIRECT=1
run direct_nooverlap $SL RXMD_HALO_DIRECT=1 RXMD_NO_HALO_OVERLAP=1
run single_rank_again RXMD_X=0
