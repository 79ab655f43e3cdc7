#!/bin/bash
# round 2, call g: minimiser + full suite, profile set r02 (kernel stats, PMC traffic of the matrix pass), other workloads
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
python3 bench.py --steps 20 --warmup 5 > $O/bench_default.json 2> $O/bench_default.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt > $O/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_spmv" --output-format csv -d $O/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-alt > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_spmv" --output-format csv -d $O/pmc_write -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-alt > $O/pmc_write.log 2>&1
python3 bench.py --workload water --steps 10 --warmup 2 --no-alt --no-cpu-baseline > $O/bench_water.json 2>/dev/null
python3 bench.py --workload sicnp --steps 20 --warmup 5 --no-alt --no-cpu-baseline > $O/bench_sicnp.json 2>/dev/null
python3 - <<PY
import csv,glob
f=glob.glob("$O/prof/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<10: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
for nm in ("fetch","write"):
    f=glob.glob("$O/pmc_%s/**/*counter_collection.csv"%nm,recursive=True)[0]
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_spmv<0" in r["Kernel_Name"]]
    print(nm, len(v), sum(v)/max(len(v),1))
PY
tail -6 $O/pytest.log
