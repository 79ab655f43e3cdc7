#!/bin/bash
# round 3: HBM traffic and cache counters of the ring pass against the wavefront-per-row pass (separate --pmc passes, kernel-trace only)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
rocprofv3 --list-avail > $O/avail.txt 2>&1
grep -o "TCP_[A-Z_0-9a-z]*\|TCC_[A-Z_0-9a-z]*" $O/avail.txt | sort -u | tr '\n' ' ' | cut -c1-3000
echo
for ctr in FETCH_SIZE WRITE_SIZE "TCP_TCC_READ_REQ_sum" "TCC_HIT_sum TCC_MISS_sum" "TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $ctr | tr ' ' '_')
  RXMD_RING_CYCLIC=0 rocprofv3 --pmc $ctr --kernel-include-regex "k_spmv" --output-format csv -d $O/pmc_$tag -- python3 scripts/gpu_ring_probe.py RXMD_RING_PROBE=0,RXMD_RING_CYCLIC=0 RXMD_RING_PROBE=1,RXMD_RING_CYCLIC=0 > $O/pmc_$tag.log 2>&1
  python3 - <<PY
import csv,glob,collections
fs=glob.glob("$O/pmc_$tag/**/*counter_collection.csv",recursive=True)
if not fs: print("$ctr: no output"); raise SystemExit
d=collections.defaultdict(list)
for r in csv.DictReader(open(fs[0])):
    d[(r["Kernel_Name"][:40],r["Counter_Name"])].append(float(r["Counter_Value"]))
for k,v in d.items(): print(k, len(v), "first10 avg %.4g"%(sum(v[:10])/max(1,len(v[:10]))), "last10 avg %.4g"%(sum(v[-10:])/len(v[-10:])))
PY
done
