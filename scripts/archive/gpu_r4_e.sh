#!/bin/bash
# round 4: full GPU suite on the current tree, then the default bench under the kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -25 $O/pytest.log | cut -c1-300
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs --no-steady > $O/bench_prof.log 2>&1
grep '^{"metric' $O/bench_prof.log > $O/bench_prof.json
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; python3 - <<PY
import csv
rows=list(csv.DictReader(open("$O/kernel_stats.csv")))
for r in rows[:24]: print("%-60s n=%-5s avg %.1f us  %s%%" % (r["Name"].replace("void ","").replace("rxmd::","")[:60], r["Calls"], float(r["AverageNs"])/1e3, r["Percentage"]))
import json; d=json.load(open("$O/bench_prof.json")); print('ms/step', d['ms_per_step'], 'pass', d['roofline']['avg_launch_ms'], 'iters', d['qeq_iters_per_step']); print(d['breakdown_ms_per_step'])
PY
find $O/prof -name '*.csv' ! -name '*stats*' -delete; find $O/prof -name '*.db' -delete
