# A/B of the QEq matrix-pass variants on the bench workload (prints avg launch ms per variant)
cd $GRAFT_REPO_ROOT
for v in "RXMD_SPMV_BS=1024" "RXMD_SPMV_BS=512" "RXMD_SPMV_BS=256" "RXMD_SPMV_ROWS=1"; do
  env $v python bench.py --no-cpu-baseline --steps 6 --warmup 1 --no-alt 2>&1 | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read())
print('$v', 'ms/step', round(d['ms_per_step'],2), 'spmv ms', round(d['roofline']['avg_launch_ms'],4), 'lists', round(d['breakdown_ms_per_step']['ms_lists'],2), 'iters', d['qeq_iters_per_step'])"
done
