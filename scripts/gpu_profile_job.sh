# rocprofv3 kernel statistics of the default bench (qeq_mode 1) + PMC HBM traffic of the matrix pass; outputs under gpurun_out/
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-x}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --alt-steps 0 > gpurun_out/prof_$TAG.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$TAG/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<24: print(r["Name"][:64],r["Calls"],r["AverageNs"],r["Percentage"])
PY
grep '^{"metric' gpurun_out/prof_$TAG.log | cut -c1-200
