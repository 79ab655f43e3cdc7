#!/bin/bash
# round 2, call e: skewed box, checkpoint bytes, E3b work order, matrix-pass pipeline levels (same box)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -q -k "output or skewed or other_force or software_pipeline or bitwise or tight_tolerance or stress" > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
B="python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline"
RXMD_SPMV_PIPE=1 $B > $O/b_pipe1.json 2> $O/b_pipe1.err
RXMD_SPMV_PIPE=2 $B > $O/b_pipe2.json 2>/dev/null
RXMD_SPMV_PIPE=0 $B > $O/b_pipe0.json 2>/dev/null
RXMD_SPMV_PIPE=1 $B > $O/b_pipe1_e3batom.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/prof_default.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof_default/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<16: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
PY
tail -8 $O/pytest.log
