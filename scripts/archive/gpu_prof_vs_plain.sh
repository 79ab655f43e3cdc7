#!/bin/bash
# the same bench leg plain / with a blocking wait / without run-ahead / under rocprofv3: is the pass time a property of the box or of the host loop?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
show() { grep '^{"metric' $1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$2', 'ms/step', round(d['ms_per_step'],2), 'pass', round(d['roofline']['avg_launch_ms'],4), 'iters', round(d['qeq_iters_per_step'],1))"; }
A="--no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs"
python3 bench.py $A > /tmp/b1.log 2>/dev/null; show /tmp/b1.log plain
RXMD_SPIN_WAIT=0 python3 bench.py $A > /tmp/b2.log 2>/dev/null; show /tmp/b2.log blocking-wait
RXMD_CG_NO_RUNAHEAD=1 python3 bench.py $A > /tmp/b3.log 2>/dev/null; show /tmp/b3.log no-runahead
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python3 bench.py $A > /tmp/b4.log 2>&1; show /tmp/b4.log under-rocprofv3
python3 bench.py $A > /tmp/b5.log 2>/dev/null; show /tmp/b5.log plain-again
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -6
