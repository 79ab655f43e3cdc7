#!/bin/bash
# round 2: the N > 1 code paths of bench.py on ONE GPU -- (a) 2 and 8 ranks over the callback transport (gloo, host-staged), (b) the native
# RCCL transport as a self loop at full size, (c) the same with the fallback TorchTransport device buffers sized by the new bound
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for N in 2 8; do
  RXMD_BENCH_BACKEND=gloo RXMD_BENCH_DEVICE=0 RXMD_SINGLE_STREAM=1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2953$N bench.py --gpus $N --cells 6 --steps 3 --warmup 1 --no-cpu-baseline > $O/bench_mr$N.log 2>&1
  echo "mr$N rc=$?" >> $O/rc.txt
  grep '^{"metric' $O/bench_mr$N.log | cut -c1-1200 || tail -20 $O/bench_mr$N.log
done
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
RXMD_BENCH_FORCE_DIST=1 RXMD_FORCE_STAGED=1 RXMD_FORCE_REMOTE=1 timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-alt > $O/selfloop.log 2>&1
echo "selfloop rc=$?" >> $O/rc.txt
tail -3 $O/selfloop.log | cut -c1-1500
RXMD_BENCH_FORCE_DIST=1 RXMD_FORCE_STAGED=1 RXMD_BENCH_TRANSPORT=torch timeout 900 python bench.py --gpus 1 --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $O/selfloop_torch.log 2>&1
echo "selfloop_torch rc=$?" >> $O/rc.txt
tail -3 $O/selfloop_torch.log | cut -c1-600
cat $O/rc.txt
