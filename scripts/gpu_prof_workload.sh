#!/bin/bash
# kernel statistics of one bench workload: gpu_prof_workload.sh <tag> <workload>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --workload $2 --steps 4 --warmup 1 --no-cpu-baseline --no-alt > $O/prof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<16: print("  ",r["Name"].replace("void ","")[:50],r["Calls"],"%.1f us"%(float(r["AverageNs"])/1e3),r["Percentage"])
PY
