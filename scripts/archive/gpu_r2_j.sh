#!/bin/bash
# round 2, call j: DPP wavefront sums -- full suite, bench, kernel statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline > $O/bench.json 2>/dev/null
python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline > $O/bench2.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/prof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<14: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
PY
tail -6 $O/pytest.log
