#!/bin/bash
# round 3, end: full GPU suite; the opt-in matrix-pass kernels on the parity subset; the default bench line as the driver runs it; kernel statistics
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $O/pytest.log | tail -2
for v in RXMD_SPMV2=1 RXMD_SPMV_DMA=1 "RXMD_SPMV_RING=1 RXMD_RING_MIN_ROWS=0"; do
env $v timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -k "tight_tolerance_parity_vs_oracle or one_pass or pqeq_step0 or perturbed_rdx_36k_against" > $O/pytest_opt.log 2>&1; echo "$v rc=$?"; grep -E "passed|failed" $O/pytest_opt.log | tail -1
done
python3 bench.py > $O/bench_default.log 2>$O/bench_default.err; grep '^{"metric' $O/bench_default.log > $O/bench_default.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-configs > $O/prof_bench.log 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv; grep '^{"metric' $O/prof_bench.log > $O/bench_under_profiler.json
python3 - <<PY
import json,csv
d=json.load(open("$O/bench_default.json"))
print("default: steps/s", round(d["value"],2), "ms/step", round(d["ms_per_step"],2), "spmv", round(d["roofline"]["avg_launch_ms"],4), "frac", round(d["roofline"]["frac"],3), "iters", d["qeq_iters_per_step"], "cpu", round(d["cpu_baseline"]["atom_steps_per_s"]), d["cpu_baseline"]["cores"])
print("alt", d["alt"]["ms_per_step"], "alt_lex", d["alt_lex"]["ms_per_step"], [ (o["workload"][:20], round(o["steps_per_s"],2), round(o["roofline"]["frac"],3)) for o in d["other_configs"]])
for k in d["roofline"]["kernels"]: print("  ", k["name"][:44], round(k["ms"],4), "frac", round(k["frac"],3), "traffic", k["traffic"])
for i,r in enumerate(csv.DictReader(open("$O/kernel_stats.csv"))):
    if i<8: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
PY
