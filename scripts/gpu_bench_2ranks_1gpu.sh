# bench.py's N > 1 code path with the callback transport: 2 (or 8) ranks share GPU 0, messages host-staged over gloo
cd $GRAFT_REPO_ROOT
N=${1:-2}
RXMD_BENCH_BACKEND=gloo RXMD_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $N --cells ${2:-6} --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_mr$N.log 2>&1
grep '^{"metric' gpurun_out/bench_mr$N.log | cut -c1-700 || tail -20 gpurun_out/bench_mr$N.log
