#!/bin/bash
# bench.py's N = 2 code path on ONE GPU (two ranks share device 0, messages host-staged over gloo), window pass on and off: same energies
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for w in 1 0; do
RXMD_SPMV_WIN=$w RXMD_BENCH_BACKEND=gloo RXMD_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --cells 6 --steps 3 --warmup 1 --no-cpu-baseline --no-alt --no-other-configs > gpurun_out/bench_mr2_$w.log 2>&1
grep '^{"metric' gpurun_out/bench_mr2_$w.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); p=d['per_rank']; r=d['roofline']
print('win=$w', 'ms/step', round(d['ms_per_step'],1), r['kernel'], round(r['avg_launch_ms'],4), 'iters', d['qeq_iters_per_step'], 'PE/atom', d['energy_per_atom']['PE'], 'boundary', p['boundary_rows'], 'halo', p['ms_halo_per_step'], 'exposed', p['ms_halo_exposed_per_step'])" || tail -5 gpurun_out/bench_mr2_$w.log
done
