#!/bin/bash
# round 2, call c: new tests (PQEq cluster golden, output path, MPI drop-in, lex drop-in), kernel statistics of the new sweep, speculative-batch A/B
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_gpu_output.py tests/test_gpu_dropin.py tests/test_gpu_parity.py -q -k "output or dropin or hip_library or pqeq or variants or benchmark_tolerance" > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
B="python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline"
$B > $O/b_default.json 2> $O/b_default.err
RXMD_SPMV_SPEC=1 $B > $O/b_spec.json 2>/dev/null
RXMD_SPMV_SPEC=1 RXMD_SPMV_IDX32=1 $B > $O/b_spec_idx32.json 2>/dev/null
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_default -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/prof_default.log 2>&1
RXMD_LIST_NO_FP32=1 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_nofp32 -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > $O/prof_nofp32.log 2>&1
python3 - <<PY
import csv,glob
for t in ("default","nofp32"):
    f=glob.glob("$O/prof_%s/**/*kernel_stats.csv"%t,recursive=True)[0]
    print(t)
    for i,r in enumerate(csv.DictReader(open(f))):
        if i<14: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
PY
tail -15 $O/pytest.log
