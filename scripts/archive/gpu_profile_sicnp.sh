# rocprofv3 kernel statistics of the SiC-NP + PQEq workload (BASELINE configs[4]) on one GPU
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_sicnp -- python3 bench.py --workload sicnp --steps 5 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/prof_sicnp.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_sicnp/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<18: print(r["Name"][:70],r["Calls"],r["AverageNs"],r["Percentage"])
PY
