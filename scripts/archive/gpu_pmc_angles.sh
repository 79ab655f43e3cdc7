# instruction mix of the angle kernels (k_e3b, k_e4b) and the list/nonbond kernels: one --pmc pass per counter group
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-include-regex "k_e4b|k_e3b|k_list10|k_nonbond|k_ehb" --output-format csv -d gpurun_out/pmc_ang_$tag -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-alt > gpurun_out/pmc_ang_$tag.log 2>&1
done
python3 - <<'PY'
import csv,glob,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("gpurun_out/pmc_ang_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("rxmd::","")[:24]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in acc.items():
    print(k, {c: "%.4g"%(sum(x)/len(x)) for c,x in sorted(v.items())})
PY
