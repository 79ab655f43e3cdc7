#!/bin/bash
# kernel times of several builds of the library in one box: the default and every rxmd_amd/librxmd_hip_<tag>.so ($2 = kernel name filter)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
F=${2:-k_nonbond}
for lib in default $(ls rxmd_amd/librxmd_hip_*.so 2>/dev/null) default; do
  t=$(basename $lib .so)
  if [ "$lib" = "default" ]; then unset RXMD_HIP_LIB; else export RXMD_HIP_LIB=$GRAFT_REPO_ROOT/$lib; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$t -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-alt > $O/prof_$t.log 2>&1
  python3 - <<PY
import csv,glob
f=sorted(glob.glob("$O/prof_$t/**/*kernel_stats.csv",recursive=True))[-1]
for r in csv.DictReader(open(f)):
    if "$F" in r["Name"]: print("$t", r["Name"].replace("void ","")[:24], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
done
