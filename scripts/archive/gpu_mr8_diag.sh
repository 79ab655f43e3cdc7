# why are 8 ranks on ONE GPU slow?  (artificial configuration: the driver's N=8 run has one GPU per rank)
cd $GRAFT_REPO_ROOT
run() { tag=$1; shift
  env "$@" RXMD_BENCH_BACKEND=gloo RXMD_BENCH_DEVICE=0 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 8 --cells 4 --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > gpurun_out/mr8_$tag.log 2>&1
  echo "$tag: $(grep '^{"metric' gpurun_out/mr8_$tag.log | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['breakdown_ms_per_step']['ms_qeq'])")"; }
run single_stream RXMD_SINGLE_STREAM=1 RXMD_NO_HALO_OVERLAP=1
EXTRA="--qeq-mode 0" run single_stream_mode0 RXMD_SINGLE_STREAM=1 RXMD_NO_HALO_OVERLAP=1
run default X=1
