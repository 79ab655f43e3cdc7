#!/bin/bash
# window pass: full GPU suite, smoke, default bench line, kernel statistics, per-kernel PMC traffic
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 3000 python3 -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -6
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
python3 bench.py > $O/bench_default.log 2>$O/bench_default.err; grep '^{"metric' $O/bench_default.log > $O/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-configs > $O/prof_bench.log 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; grep '^{"metric' $O/prof_bench.log > $O/bench_under_profiler.json
bash scripts/gpu_pmc_kernels.sh $1/pmc > $O/pmc_summary.txt 2>&1; head -8 $O/pmc_summary.txt
python3 - <<PY
import json,csv
d=json.load(open("$O/bench_default.json")); u=json.load(open("$O/bench_under_profiler.json"))
r=d["roofline"]
print("default: steps/s", round(d["value"],2), "ms/step", round(d["ms_per_step"],2), "pass", r["kernel"], round(r["avg_launch_ms"],4), "frac", round(r["frac"],3), "streamed GB/s", round(r["streamed_GBs"]), "iters", d["qeq_iters_per_step"], "launches", r["launches"], "noop", r["launches_that_returned_at_once"])
print(" breakdown", {k: round(v,2) for k,v in d["breakdown_ms_per_step"].items()})
print(" alt_lex", d.get("alt_lex",{}).get("ms_per_step"), [ (o["workload"][:20], round(o["ms_per_step"],2), o["roofline"]["kernel"], round(o["roofline"]["avg_launch_ms"],3), round(o["roofline"]["frac"],3)) for o in d.get("other_configs",[])])
print("under profiler: spmv avg", round(u["roofline"]["avg_launch_ms"],4))
for i,r in enumerate(csv.DictReader(open("$O/kernel_stats.csv"))):
    if i<8: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
PY
