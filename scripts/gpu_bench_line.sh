#!/bin/bash
# The default bench line with every leg (CPU baseline: the small sample unless FULL=1) and a one-screen summary.   usage: bash scripts/gpu_bench_line.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
if [ "$FULL" = "1" ]; then timeout -k 10 1200 python3 bench.py > $O/bench.log 2>&1; else timeout -k 10 800 python3 bench.py --cpu-baseline-sample small > $O/bench.log 2>&1; fi
grep '^{"metric' $O/bench.log > $O/bench.json
python3 -c "
import json; d=json.load(open('$O/bench.json')); r=d['roofline']
print('steps/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],2), 'pass', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), 'iters', d['qeq_iters_per_step'], 'bond_overlap', d.get('bond_overlap'))
print('steady', round(d['steady']['ms_per_step'],2), d['steady']['qeq_iters_per_step'], 'lex', round(d['alt_lex']['ms_per_step'],2), 'one_stream', round(d['alt_one_stream']['ms_per_step'],2), 'noplace', round(d['alt_no_placement_search']['ms_per_step'],2), 'alt', round(d['alt']['ms_per_step'],2))
print({k: round(v,2) for k,v in d['breakdown_ms_per_step'].items() if v})
print([(k['name'][:16], round(k['ms'],3), k.get('bound'), k.get('frac_of_bound') and round(k['frac_of_bound'],2), k.get('timed')) for k in r['kernels']])
print('other', [(o.get('workload','')[:12], round(o.get('ms_per_step',0),2), round(o.get('roofline',{}).get('frac',0),3)) for o in d.get('other_configs',[])], 'cpu', d.get('cpu_baseline',{}).get('value'), 'wall', d['bench_wall_s'])"
tail -2 $O/bench.log | cut -c1-200
