#!/bin/bash
# bench.py's N > 1 code path end to end on ONE GPU: 2 and 8 ranks share device 0, messages host-staged over gloo (functional check of per_rank + timers)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for N in 2 8; do for d in 0 1; do
RXMD_HALO_DIRECT=$d RXMD_SINGLE_STREAM=1 RXMD_BENCH_BACKEND=gloo RXMD_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N --cells 4 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mr$N.log 2>&1
grep '^{"metric' gpurun_out/bench_mr$N.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['per_rank']
print('N=$N direct=$d', 'ms/step', round(d['ms_per_step'],1), 'natoms', p['natoms'], 'boundary', p['boundary_rows'][:3], 'halo', p['ms_halo_per_step'][:3], 'exposed', p['ms_halo_exposed_per_step'][:3], 'allreduce', p['ms_allreduce_per_step'][:3])" || tail -5 gpurun_out/bench_mr$N.log
done; done
