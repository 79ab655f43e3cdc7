# the multi-GPU code path of bench.py on ONE GPU: a single rank through the staged exchange, every message through RCCL self send/recv
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
export RXMD_BENCH_FORCE_DIST=1 RXMD_FORCE_STAGED=1 RXMD_FORCE_REMOTE=1
timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/selfloop.log 2>&1
tail -5 gpurun_out/selfloop.log | cut -c1-1500
