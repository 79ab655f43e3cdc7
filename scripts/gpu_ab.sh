# generic A/B of env toggles on the default bench: usage  bash scripts/gpu_ab.sh "ENV1=1" "ENV2=1" ...   (each run twice, interleaved)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for v in "$@"; do
  env $v python bench.py --no-cpu-baseline --steps 8 --warmup 2 --no-alt 2>&1 | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('$v', 'ms/step', round(d['ms_per_step'],2), 'spmv ms', round(d['roofline']['avg_launch_ms'],4), 'iters', d['qeq_iters_per_step'], 'lists', round(b['ms_lists'],2), 'nonbond', round(b['ms_nonbond'],2), 'bonded', round(b['ms_bonded'],2))"
done
done
