#!/bin/bash
# instruction mix and wait cycles of the matrix pass kernels (SQ counters, one --pmc pass per group; cycle counters count quad-cycles)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for grp in "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_WAVES SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_ANY" "GRBM_GUI_ACTIVE SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_LDS SQ_INSTS_FLAT" "SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_UNALIGNED_STALL"; do
  tag=$(echo $grp | cut -d' ' -f1)
  rocprofv3 --pmc $grp --kernel-include-regex "k_spmv" --output-format csv -d $O/pmc_$tag -- python3 bench.py --no-cpu-baseline --steps 1 --warmup 0 --no-alt --no-other-configs > $O/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv",recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void ","").replace("rxmd::","")[:32]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out={k:{c: sum(x)/len(x) for c,x in sorted(v.items())} for k,v in acc.items()}
json.dump(out,open("$O/spmv_mix.json","w"),indent=1)
for k,v in out.items(): print(k, {c: "%.4g"%x for c,x in v.items()})
PY
