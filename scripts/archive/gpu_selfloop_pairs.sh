# self-loop bench (multi-rank code path on one GPU, every message through RCCL send/recv to self): paired stages vs one round per stage
cd $GRAFT_REPO_ROOT
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
export RXMD_BENCH_FORCE_DIST=1 RXMD_FORCE_STAGED=1 RXMD_FORCE_REMOTE=1
python -m pytest tests/test_gpu_multirank.py -x -q 2>&1 | tail -3
for rep in 1 2; do
for v in pairs nopairs; do
  if [ $v = nopairs ]; then export RXMD_NO_STAGE_PAIRS=1; else unset RXMD_NO_STAGE_PAIRS; fi
  timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/selfloop_$v.log 2>&1
  echo "$v: $(grep '^{"metric' gpurun_out/selfloop_$v.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['qeq_iters_per_step'])")"
done; done
