#!/bin/bash
# where the in-loop cost of the matrix pass sits: kernel trace of 10 steps, pass duration by position in the CG loop, idle gap before each kernel of the loop
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-other-configs --no-alt --no-steady --steps 10 > $O/bench.log 2>&1
f=$(find $O/prof -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY' | tee $O/passpos.txt
import csv, sys, collections
rows = []
with open(sys.argv[1]) as fh:
    for r in csv.DictReader(fh):
        rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']))
rows.sort()
def short(n):
    n = n.split('(')[0]
    return n.replace('void ', '')[:44]
pos = None; bypos = collections.defaultdict(list); prevk = collections.defaultdict(list)
gap = collections.defaultdict(list); dur = collections.defaultdict(list)
last_end = None; last_name = None; inloop = False
for s, e, n in rows:
    k = short(n)
    if 'k_list10' in k: pos = 0; inloop = False
    if 'k_spmv_win' in k and pos is not None:
        bypos[min(pos, 40)].append((e - s) / 1e3); prevk[last_name].append((e - s) / 1e3); pos += 1; inloop = True
    if inloop and last_end is not None:
        gap[k].append((s - last_end) / 1e3); dur[k].append((e - s) / 1e3)
    if 'k_apply_q' in k: inloop = False
    last_end = e; last_name = k
print('pass duration by position in the step (us):')
for p in sorted(bypos):
    v = bypos[p]; print(f'  {p:3d} n={len(v):4d} mean={sum(v)/len(v):8.1f} min={min(v):8.1f} max={max(v):8.1f}')
print('pass duration by preceding kernel:')
for k, v in prevk.items(): print(f'  {k:46s} n={len(v):4d} mean={sum(v)/len(v):8.1f}')
print('kernels of the loop: mean duration, mean idle gap before (us), count')
tot = 0
for k in sorted(dur, key=lambda k: -sum(dur[k]) - sum(gap[k])):
    print(f'  {k:46s} dur={sum(dur[k])/len(dur[k]):8.1f} gap={sum(gap[k])/len(gap[k]):7.1f} n={len(dur[k])}')
PY
find $O/prof -name '*.csv' -delete; find $O/prof -name '*.db' -delete
