# BASELINE configs[2] and [4] on one GPU, next to the headline
cd $GRAFT_REPO_ROOT
for w in water sicnp; do
  timeout 900 python bench.py --workload $w --steps 5 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/bench_$w.log 2>&1
  tail -1 gpurun_out/bench_$w.log | cut -c1-2500
  echo
done
