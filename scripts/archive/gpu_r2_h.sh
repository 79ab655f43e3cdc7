#!/bin/bash
# round 2, call h: torsions once (mailbox) -- parity subset, A/B against the two-visit kernel; minimiser probe
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python3 scripts/gpu_min_debug.py > $O/min_debug.log 2>&1
python -m pytest tests/test_gpu_parity.py -q -x -k "tight_tolerance or md_trajectory_tight or stress or bitwise or other_force or skewed or migration or injected" > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
B="python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline"
$B > $O/b_once.json 2> $O/b_once.err
RXMD_E4B_TWO_VISITS=1 $B > $O/b_two.json 2>/dev/null
$B > $O/b_once2.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/prof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<12: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
PY
tail -6 $O/pytest.log; cat $O/min_debug.log | tail -12
