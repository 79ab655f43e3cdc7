#!/bin/bash
# timing probes of k_e4b (RXMD_E4B_PROBE, see bonded.hip) + batch statistics; results are garbage in probe modes
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_E4B_PROBE=5 python3 scripts/gpu_e4b_count.py > $O/count.log 2>&1; cat $O/count.log | tail -3
for v in "RXMD_E4B_PROBE=1" "RXMD_E4B_PROBE=2" "NONE=1"; do
  env $v rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$v -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt > $O/prof_$v.log 2>&1
  python3 - <<PY
import csv,glob
f=glob.glob("$O/prof_$v/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if "e4b" in r["Name"] or "e3b" in r["Name"]: print("$v", r["Name"][:40], r["Calls"], r["AverageNs"])
PY
done
