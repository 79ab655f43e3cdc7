#!/bin/bash
# placement search of the pass's streams on / off, fresh processes alternating
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for rep in 1 2 3 4; do for t in 1 6; do
  RXMD_PLACE_TRIES=$t python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; p=r['placement_search']
print('tries=$t', 'ms/step', round(d['ms_per_step'],2), 'pass', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), 'first/kept', round(p['pass_ms_first_placement'],4), round(p['pass_ms_kept_placement'],4))"
done; done
