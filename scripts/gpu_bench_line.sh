#!/bin/bash
# The default bench line with every leg (CPU baseline: the small sample unless FULL=1) and a one-screen summary.   usage: bash scripts/gpu_bench_line.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
if [ "$FULL" = "1" ]; then timeout -k 10 1200 python3 bench.py > $O/bench.log 2>&1; else timeout -k 10 800 python3 bench.py --cpu-baseline-sample small > $O/bench.log 2>&1; fi
grep '^{"metric' $O/bench.log > $O/bench.json
python3 -c "
import json; d=json.load(open('$O/bench.json'))
for l in d.get('legs', []): print(l)
print({k: round(v,2) for k,v in d['breakdown_ms_per_step'].items() if v})
print('cpu', d.get('cpu_baseline',{}).get('value'), 'wall', d.get('bench_wall_s'))"
tail -2 $O/bench.log | cut -c1-200
