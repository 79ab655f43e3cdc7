#!/bin/bash
# timing probes of k_ehb (RXMD_EHB_PROBE): low byte 1 = without the three atomic additions to the acceptor's force (garbage results);
# bits 8.. = atoms per wavefront (experiment; default 64)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for v in "NONE=1" "RXMD_EHB_PROBE=8192" "RXMD_EHB_PROBE=4096" "RXMD_EHB_PROBE=2048" "RXMD_EHB_PROBE=1024" "RXMD_EHB_PROBE=1"; do
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $O/prof_$v.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$O/prof_$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "k_ehb" in r["Name"]: print("$v", r["Name"][:20], r["Calls"], "%.1f us" % (float(r["AverageNs"])/1e3))
PY
done
