#!/bin/bash
# Does the pass time bench.py reports (HIP event pairs on every 8th launch inside the CG loop) agree with the kernel trace of the same command?
# The placement search is switched off (RXMD_PLACE_TRIES=1) so that every launch of the instance in the trace is a launch of the CG loop; launches that
# returned at once (the iteration queued ahead of the exit decision, < 100 us) are left out of the trace average, as bench.py leaves them out of its own.
# usage: bash scripts/gpu_trace_agree.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_PLACE_TRIES=1 timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-other-configs --no-alt --full-line $O/bench_full.json > $O/bench.log 2>&1
python3 - <<PY
import csv, glob, json
d = json.load(open("$O/bench_full.json")); r = d["roofline"]
f = sorted(glob.glob("$O/prof/**/*kernel_trace.csv", recursive=True))[-1]
dur = [ (int(x["End_Timestamp"]) - int(x["Start_Timestamp"])) / 1e3 for x in csv.DictReader(open(f)) if "k_spmv_win<0, true, false" in x["Kernel_Name"] ]
real = [t for t in dur if t > 100.0]
s = d.get("steady", {})
n_head = r["launches"]; n_steady = int(round(s.get("spmv_launches_per_step", 0) * 100))
txt = ("kernel trace: %d launches of the dispatched instance, %d of them returned at once (< 100 us); average of the others %.1f us\\n"
       "bench.py, same run: headline leg %d launches, %d timed by event pairs, avg %.1f us; steady leg %d launches, avg %.1f us; launch-weighted %.1f us\\n"
       % (len(dur), len(dur) - len(real), sum(real) / len(real), n_head, r.get("launches_timed", 0), 1e3 * r["avg_launch_ms"], n_steady, 1e3 * s.get("avg_pass_ms", 0),
          (n_head * 1e3 * r["avg_launch_ms"] + n_steady * 1e3 * s.get("avg_pass_ms", 0)) / max(n_head + n_steady, 1)))
print(txt); open("$O/trace_agreement.txt", "w").write(txt)
PY
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; head -6 $O/kernel_stats.csv | cut -c1-60,300-420
find $O/prof -name '*.csv' ! -name '*stats*' -delete; find $O/prof -name '*.db' -delete
