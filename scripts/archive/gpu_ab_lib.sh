# A/B of differently built libraries (rxmd_amd/librxmd_hip_<tag>.so) on the default bench
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for t in "$@"; do
  RXMD_HIP_LIB=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_$t.so python bench.py --no-cpu-baseline --steps 6 --warmup 2 --no-alt 2>&1 | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('$t', 'ms/step', round(d['ms_per_step'],2), 'spmv', round(d['roofline']['avg_launch_ms'],4), 'iters', round(d['qeq_iters_per_step'],1), 'lists', round(b['ms_lists'],2), 'nonbond', round(b['ms_nonbond'],2), 'bo', round(b['ms_bo'],2), 'bonded', round(b['ms_bonded'],2))"
done
done
