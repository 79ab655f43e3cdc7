#!/bin/bash
# For the first lease on an 8-GPU node: the weak-scaling bench at N = 8 under the four halo variants, then N = 1, 2, 4, 8 as the driver runs it.
# Every line carries per_rank exchange timers (ms_halo, ms_halo_exposed, ms_allreduce, ms_ghost_build, ms_migrate, ms_fold per step).
# usage: bash scripts/gpu8_ab.sh <outdir> [steps]
cd ${GRAFT_REPO_ROOT:-.}; O=${1:-gpurun_out/gpu8}; S=${2:-20}; mkdir -p $O
run() { tag=$1; n=$2; shift 2
  env "$@" HSA_ENABLE_IPC_MODE_LEGACY=0 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus $n --steps $S --warmup 5 --no-cpu-baseline 2>$O/$tag.err | grep '^{"metric' > $O/$tag.json
  python3 -c "
import json; d=json.load(open('$O/$tag.json')); p=d.get('per_rank',{})
print('$tag', 'steps/s', round(d['value'],2), 'ms/step', round(d['ms_per_step'],2), 'iters', round(d['qeq_iters_per_step'],1), {k:[round(x,2) for x in v] for k,v in p.items() if k.startswith('ms_halo') or k.startswith('ms_allreduce')})" || tail -3 $O/$tag.err; }
run n8_staged_overlap 8 RXMD_X=0
run n8_staged_nooverlap 8 RXMD_NO_HALO_OVERLAP=1
run n8_direct_overlap 8 RXMD_HALO_DIRECT=1
run n8_direct_nooverlap 8 RXMD_HALO_DIRECT=1 RXMD_NO_HALO_OVERLAP=1
for n in 2 4; do run n${n}_staged_overlap $n RXMD_X=0; done
python3 bench.py --steps $S --warmup 5 --no-cpu-baseline --no-other-configs --no-alt 2>/dev/null | grep '^{"metric' > $O/n1.json
