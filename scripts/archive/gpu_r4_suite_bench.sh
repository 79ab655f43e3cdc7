#!/bin/bash
# whole GPU suite + smoke + the default bench line
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
RXMD_PLACE_VERBOSE=1 python3 bench.py > $O/bench_default.log 2> $O/bench_default.err; grep '^{"metric' $O/bench_default.log > $O/bench_default.json
grep "placement draw" $O/bench_default.err | sed 's/  hess.*//' | tr '\n' ';'; echo
python3 -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']
print('steps/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],2), 'pass', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), 'place', r['placement_search']['pass_ms_first_placement'], r['placement_search']['pass_ms_kept_placement'], 'wall', d['bench_wall_s'])
print('steady', d['steady']['ms_per_step'], d['steady']['steps_per_s'], 'noplace', d['alt_no_placement_search']['ms_per_step'], 'lex', d['alt_lex']['ms_per_step'])
print('other', [(o['workload'][:20], round(o['ms_per_step'],2), round(o['roofline']['avg_launch_ms'],3), round(o['roofline']['frac'],3)) for o in d['other_configs']])"
