#!/bin/bash
# Is the placement lottery of the matrix pass a rule?  experiments build, RXMD_PLACE_SCAN=<slabs>: see Engine::tune_window_placement.
# usage: bash scripts/gpu_place_scan.sh <tag> [slabs]
cd $GRAFT_REPO_ROOT; O=gpurun_out/${1:-place_scan}; mkdir -p $O
for rep in 1 2; do
RXMD_HIP_LIB=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_exp.so RXMD_PLACE_SCAN=${2:-3} RXMD_PLACE_TRIES=2 timeout -k 10 600 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-alt --no-other-configs --no-steady > $O/scan_$rep.log 2> $O/scan_$rep.err
grep place_scan $O/scan_$rep.err > $O/place_scan_process$rep.txt; wc -l $O/place_scan_process$rep.txt
done
python3 - <<PY
import re, collections
for rep in (1, 2):
    rows = [l for l in open("$O/place_scan_process%d.txt" % rep)]
    slabs = collections.defaultdict(list)
    for l in rows:
        m = re.search(r"slab (\d+) \S+\s+oh\s+(\d+)\s+os\s+(\d+)\s+([\d.]+) ms", l)
        if m: slabs[int(m.group(1))].append((int(m.group(2)), int(m.group(3)), float(m.group(4))))
    print("process", rep, rows[0].strip() if rows else "")
    for sidx, v in sorted(slabs.items()):
        t = [x[2] for x in v]
        print("  slab %d: min %.4f max %.4f mean %.4f | os scan %s | oh scan %s" % (sidx, min(t), max(t), sum(t) / len(t), " ".join("%.3f" % x[2] for x in v if x[0] == 0), " ".join("%.3f" % x[2] for x in v if x[0] != 0)))
    for l in rows:
        if "separate draw" in l or "its value stream" in l or "sequential read" in l or "refused" in l or "failed" in l or "no VMM" in l: print("  " + l.strip().replace("[place_scan] ", ""))
PY
