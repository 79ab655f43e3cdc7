#!/bin/bash
# What the wavefronts of the non-CG kernels wait for: SQ / TCP counters per kernel, one rocprofv3 --pmc pass per counter set (kernel trace only).
# usage: bash scripts/gpu_pmc_sq.sh <tag> [kernel regex] [bench args]      -> gpurun_out/<tag>/sq_per_kernel.txt (+ counters_available.txt)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
RX=${1:-"k_ehb|k_e4b|k_e3b|k_list10|k_nonbond_win|k_bonded_list|k_bond_csr|k_win_columns"}; shift
ARGS="--steps 2 --warmup 1 --no-alt --no-cpu-baseline --no-other-configs --no-steady $@"
export RXMD_PLACE_TRIES=1
rocprofv3 -L > $O/counters_available.txt 2>&1
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT" \
           "TCP_TCC_ATOMIC_WITH_RET_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_WAVES_EQ_64 SQ_LEVEL_WAVES SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT" \
           "TCC_EA0_ATOMIC_sum TCC_ATOMIC_sum" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  i=$((i+1))
  timeout 600 rocprofv3 --pmc $set --kernel-include-regex "$RX" --output-format csv -d $O/pmc_$i -- python3 bench.py $ARGS > $O/pmc_$i.log 2>&1 || echo "pass $i ($set) failed: $(tail -2 $O/pmc_$i.log | cut -c1-200)"
done
python3 - <<PY
import csv, glob, collections, re
out = collections.defaultdict(dict); calls = {}
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rxmd::", "")
        d[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, c), v in d.items():
        out[name][c] = sum(v) / len(v); calls[name] = len(v)
with open("$O/sq_per_kernel.txt", "w") as fo:
    for name, c in sorted(out.items()):
        fo.write("%s  (launches averaged: %d)\n" % (name, calls[name]))
        for k in sorted(c): fo.write("    %-40s %.4g\n" % (k, c[k]))
        wc = c.get("SQ_WAVE_CYCLES")
        if wc:
            fo.write("    -> of the wave cycles: parked (WAIT_ANY) %.2f, issue stall (WAIT_INST_ANY) %.2f, issuing (ACTIVE_INST_ANY) %.2f, VALU %.2f\n" % (
                c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_VALU", 0) / wc))
print(open("$O/sq_per_kernel.txt").read()[:6000])
PY
find $O -name '*.csv' -size +2M -delete; find $O -name '*.db' -delete
