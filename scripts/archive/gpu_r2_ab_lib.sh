#!/bin/bash
# A/B of two builds of the library in one box: default vs rxmd_amd/librxmd_hip_alt.so (RXMD_HIP_LIB); parity tests with the alternative
# build first, then kernel times from rocprofv3 (workload: $2, default rdx)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
W=${2:-rdx}
RXMD_HIP_LIB=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_alt.so python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or rdx222 or other or stress" > $O/pytest.log 2>&1; grep -E "passed|failed" $O/pytest.log | tail -1
for v in "NONE=1" "RXMD_HIP_LIB=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_alt.so" "NONE=2"; do
  t=$(echo $v | tr '/=' '__' | cut -c1-16)
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$t -- python3 bench.py --workload $W --steps 4 --warmup 1 --no-cpu-baseline --no-alt > $O/prof_$t.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$O/prof_$t/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k_e4b","k_e3b","k_ehb","k_nonbond","k_list10")): print("$t", r["Name"].replace("void ","")[:22], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
done
