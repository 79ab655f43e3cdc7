#!/bin/bash
# round 3, call a: parity at benchmark scale (tests/test_gpu_scale.py), tap checks, baseline bench + kernel statistics of the round-2 kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_gpu_scale.py tests/test_gpu_parity.py -m gpu -q -x > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline > $O/bench.json 2>$O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/prof.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<24: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
PY
tail -15 $O/pytest.log; cat $O/rc.txt; cut -c1-400 $O/bench.json
