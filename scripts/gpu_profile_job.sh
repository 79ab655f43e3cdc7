cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
python bench.py > gpurun_out/bench_d.log 2>&1
tail -1 gpurun_out/bench_d.log | cut -c1-600
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --alt-steps 0 > gpurun_out/prof_d.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-include-regex "k_spmv" --output-format csv -d gpurun_out/pmc_fetch -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --alt-steps 0 > gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-include-regex "k_spmv" --output-format csv -d gpurun_out/pmc_write -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --alt-steps 0 > gpurun_out/pmc_write.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/prof_d/**/*kernel_stats.csv",recursive=True)[0]
for i,r in enumerate(csv.DictReader(open(f))):
    if i<40: print(r["Name"][:70],r["Calls"],r["AverageNs"],r["Percentage"])
PY
