#!/bin/bash
# timing probe of k_ehb: RXMD_EHB_PROBE=1 leaves out the three atomic additions to the acceptor's force (results are garbage)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for v in "RXMD_EHB_PROBE=1" "NONE=1"; do
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $O/prof_$v.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$O/prof_$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_ehb" in r["Name"] or "k_nonbond" in r["Name"]: print("$v", r["Name"][:20], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
done
