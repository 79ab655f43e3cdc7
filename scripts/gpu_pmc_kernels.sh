#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE) and L1->L2 request counters of every kernel of a bench step: one rocprofv3 --pmc pass per counter
# set (kernel-trace only), averaged per kernel name -> <out>/kernel_traffic.json (copied to profiles/ by hand when it is to be judged).
# usage: COMMIT=<short hash of the commit being measured> bash scripts/gpu_pmc_kernels.sh <tag> [bench args]
# SQ_INSTS_VALU (wave instructions per launch) gives bench.py the issue floor of the vector unit per kernel (x 4 cycles / 1,024 SIMDs / clock).
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
ARGS="--steps 2 --warmup 1 --no-alt --no-cpu-baseline --no-other-configs --no-steady $@"
for ctr in FETCH_SIZE WRITE_SIZE TCP_TCC_READ_REQ_sum "TCC_HIT_sum TCC_MISS_sum" SQ_INSTS_VALU TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum; do   # (each pass under its own time limit)
  tag=$(echo $ctr | tr ' ' '_')
  RXMD_PLACE_TRIES=1 timeout -k 10 300 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_$tag -- python3 bench.py $ARGS > $O/pmc_$tag.log 2>&1
done
python3 - <<PY
import csv, glob, json, collections, re
out = collections.defaultdict(dict); calls = {}
for f in glob.glob("$O/pmc_*/**/*counter_collection.csv", recursive=True):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        name = re.sub(r"\(.*", "", r["Kernel_Name"]).replace("void ", "").replace("rxmd::", "")
        d[(name, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (name, c), v in d.items():
        out[name][c] = sum(v) / len(v); calls[name] = len(v)
natoms = None
for l in open("$O/pmc_FETCH_SIZE.log"):
    if l.startswith('{"metric'):
        j = json.loads(l); natoms = j["config"]["atoms_total"]
res = {"natoms": natoms, "commit": "${COMMIT:-unknown}", "source": "rocprofv3 --pmc, one pass per counter set, bench.py $ARGS; FETCH_SIZE / WRITE_SIZE in KB per launch (raw); "
       "hbm_bytes_per_launch = FETCH_SIZE x 2 (gfx950: a wide coalesced read is tallied at half its bytes, MI355X_MICROARCH.md -- calibrated for 16-byte-per-lane "
       "streams only, an upper bound for gather-dominated kernels) + WRITE_SIZE", "kernels": {}, "hbm_bytes_per_launch": {}}
for name, c in sorted(out.items(), key=lambda kv: -kv[1].get("FETCH_SIZE", 0) * calls[kv[0]]):
    rec = {"launches": calls[name]}; rec.update({k: round(v, 1) for k, v in c.items()})
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        rec["hbm_bytes"] = (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
        res["hbm_bytes_per_launch"][name] = rec["hbm_bytes"]
    if "TCP_TCC_READ_REQ_sum" in c: rec["l1_to_l2_read_bytes_at_128B"] = c["TCP_TCC_READ_REQ_sum"] * 128.0
    if c.get("TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum", 0) > 0: rec["atomic_requests"] = c["TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum"]
    res["kernels"][name] = rec
json.dump(res, open("$O/kernel_traffic.json", "w"), indent=1)
for name, rec in list(res["kernels"].items())[:24]:
    print("%-60s n=%-5d hbm %.3f GB  L1<-L2 %.3f GB  L2 hit/miss %s/%s" % (name[:60], rec["launches"], rec.get("hbm_bytes", 0) / 1e9, rec.get("l1_to_l2_read_bytes_at_128B", 0) / 1e9,
          rec.get("TCC_HIT_sum"), rec.get("TCC_MISS_sum")))
PY
