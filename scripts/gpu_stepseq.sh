#!/bin/bash
# the kernel sequence of one step with start times and idle gaps (kernel trace of 6 steps; prints the 4th step)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-other-configs --no-alt --no-steady --steps 6 > $O/bench.log 2>&1
f=$(find $O/prof -name '*kernel_trace.csv' | head -1); m=$(find $O/prof -name '*memory_copy_trace.csv' | head -1)
python3 - "$f" "$m" <<'PY' > $O/stepseq.txt
import csv, sys
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', '').replace('rxmd::', '')[:48]))
try:
    with open(sys.argv[2]) as fh:
        for r in csv.DictReader(fh): rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '')[:30]))
except Exception as e: print('no copy trace', e)
rows.sort()
starts = [k for k, r in enumerate(rows) if 'k_kick_drift' in r[2]]
print('steps seen', len(starts))
a, b = starts[-3], starts[-2]
t0 = rows[a][0]; last_end = rows[a - 1][1]; npass = 0
for s, e, n in rows[a:b]:
    if 'k_spmv_win' in n: npass += 1
    mid = ('k_spmv_win' in n or 'k_cg_' in n or 'k_reduce_fused' in n) and 3 < npass < 30
    if not mid: print(f'{(s - t0) / 1e3:10.1f} us  dur {(e - s) / 1e3:8.1f}  gap {(s - last_end) / 1e3:8.1f}  {n}')
    last_end = max(last_end, e)
print('step length', (rows[b][0] - t0) / 1e3)
PY
cat $O/stepseq.txt | head -400
find $O/prof -name '*.csv' -delete; find $O/prof -name '*.db' -delete
