#!/bin/bash
# round 2, call d: full GPU suite on the pipelined matrix pass + A/B against the plain loop, kernel statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
B="python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline"
$B > $O/b_default.json 2> $O/b_default.err
RXMD_SPMV_NO_PIPE=1 $B > $O/b_nopipe.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/prof_default.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof_default/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<14: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
PY
tail -8 $O/pytest.log
