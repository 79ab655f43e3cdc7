#!/bin/bash
# one default bench line of the head on whatever box the call lands on
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python3 bench.py > $O/bench_default.log 2>&1; grep '^{"metric' $O/bench_default.log > $O/bench_default.json
python3 -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']
print('steps/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],2), 'pass', round(r['avg_launch_ms'],4), 'place', round(r['placement_search']['pass_ms_first_placement'],4), round(r['placement_search']['pass_ms_kept_placement'],4), 'steady', round(d['steady']['steps_per_s'],2), 'noplace', round(d['alt_no_placement_search']['ms_per_step'],2), 'water', round(d['other_configs'][0]['ms_per_step'],2), 'sicnp', round(d['other_configs'][1]['ms_per_step'],2), 'lex', round(d['alt_lex']['ms_per_step'],2))"
