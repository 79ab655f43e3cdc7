# generic A/B of env toggles on the default bench: usage  [BENCH_ARGS="..."] bash scripts/gpu_ab.sh "ENV1=1" "ENV2=1 ENV3=0" ...   (each run twice, interleaved, under a time limit)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
ARGS=${BENCH_ARGS:---no-cpu-baseline --steps 8 --warmup 2 --no-alt}
for rep in 1 2; do
for v in "$@"; do
  env $v timeout -k 10 600 python3 bench.py $ARGS 2>&1 | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']; l=next((x for x in d.get('legs',[]) if x['leg']=='isQEq 2'),{})
print('$v', 'ms/step', round(d['ms_per_step'],2), 'spmv ms', round(d['roofline']['avg_launch_ms'],4), 'iters', d['qeq_iters_per_step'], 'lists', round(b['ms_lists'],2), 'force', round(b['ms_force'],2), 'nonbond', round(b['ms_nonbond'],2), 'bonded', round(b['ms_bonded'],2), 'non-pass', round(d['ms_per_step']-b['ms_qeq_spmv'],2), '| lex', l.get('ms_per_step') and round(l['ms_per_step'],2))"
done
done
