#!/bin/bash
# does the pass time of the timed window depend on how long the GPU has been busy before it?  fresh processes, warm-up 3 / 30 / 3 / 30 steps, then after 60 s of idling
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
run() { python3 bench.py --no-cpu-baseline --steps 20 --warmup $1 --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('warmup $1 $2', 'ms/step', round(d['ms_per_step'],2), 'pass', round(d['roofline']['avg_launch_ms'],4), 'iters', round(d['qeq_iters_per_step'],1))"; }
run 3; run 30; run 3; run 30; sleep 60; run 3 after-idle; sleep 60; run 30 after-idle
