#!/bin/bash
# round 3, final code: smoke, default bench line, kernel statistics, per-kernel PMC traffic
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py > $O/bench_default.log 2>$O/bench_default.err; grep '^{"metric' $O/bench_default.log > $O/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-configs > $O/prof_bench.log 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; grep '^{"metric' $O/prof_bench.log > $O/bench_under_profiler.json
bash scripts/gpu_pmc_kernels.sh $1/pmc > $O/pmc_summary.txt 2>&1; head -12 $O/pmc_summary.txt
python3 - <<PY
import json,csv
d=json.load(open("$O/bench_default.json")); u=json.load(open("$O/bench_under_profiler.json"))
print("default: steps/s", round(d["value"],2), "ms/step", round(d["ms_per_step"],2), "spmv", round(d["roofline"]["avg_launch_ms"],4), "frac", round(d["roofline"]["frac"],3), "noop", d["roofline"]["launches_that_returned_at_once"], "launches", d["roofline"]["launches"])
print("under profiler: spmv avg", round(u["roofline"]["avg_launch_ms"],4), "launches", u["roofline"]["launches"], "noop", u["roofline"]["launches_that_returned_at_once"])
for i,r in enumerate(csv.DictReader(open("$O/kernel_stats.csv"))):
    if i<6: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
PY
