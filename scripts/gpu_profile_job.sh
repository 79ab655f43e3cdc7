# rocprofv3 kernel statistics of the default bench (qeq_mode 1) + PMC HBM traffic of the matrix pass; outputs under gpurun_out/
# usage: bash scripts/gpu_profile_job.sh <tag>
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
TAG=${1:-x}
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-alt > gpurun_out/prof_$TAG.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_spmv" --output-format csv -d gpurun_out/pmc_fetch_$TAG -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-alt > gpurun_out/pmc_fetch_$TAG.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_spmv" --output-format csv -d gpurun_out/pmc_write_$TAG -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-alt > gpurun_out/pmc_write_$TAG.log 2>&1
python3 - <<PY
import csv,glob
f=glob.glob("gpurun_out/prof_$TAG/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<16: print(r["Name"][:64],r["Calls"],r["AverageNs"],r["Percentage"])
for nm in ("fetch","write"):
    f=glob.glob("gpurun_out/pmc_%s_$TAG/**/*counter_collection.csv"%nm,recursive=True)[0]
    v=[float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_spmv<0" in r["Kernel_Name"]]
    print(nm, len(v), sum(v)/len(v))
PY
grep '^{"metric' gpurun_out/prof_$TAG.log | cut -c1-200
