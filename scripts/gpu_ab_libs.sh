#!/bin/bash
# kernel times of several builds of the library on one box: the default and every rxmd_amd/librxmd_hip_<tag>.so (or the tags named in $3...)
# usage: bash scripts/gpu_ab_libs.sh <outdir tag> <kernel name regex> [tag ...]     each run under its own time limit (a faulting run must not hang the job)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
F=${2:-k_nonbond}; shift; shift
LIBS="$@"; [ -z "$LIBS" ] && LIBS=$(ls rxmd_amd/librxmd_hip_*.so 2>/dev/null | sed 's#.*librxmd_hip_##; s#\.so##')
for t in default $LIBS default; do
  if [ "$t" = "default" ]; then unset RXMD_HIP_LIB; else export RXMD_HIP_LIB=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_$t.so; fi
  timeout -k 10 420 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$t -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-alt --no-other-configs --no-steady > $O/prof_$t.log 2>&1 || { echo "$t: run failed"; tail -3 $O/prof_$t.log | cut -c1-200; continue; }
  python3 - <<PY
import csv,glob,re,json
f=sorted(glob.glob("$O/prof_$t/**/*kernel_stats.csv",recursive=True))[-1]
tot=0
for r in csv.DictReader(open(f)):
    if re.search(r"$F", r["Name"]): print("  $t %-44s n=%-5s %9.1f us" % (r["Name"].replace("void ","").replace("rxmd::","")[:44], r["Calls"], float(r["AverageNs"])/1e3))
for l in open("$O/prof_$t.log"):
    if l.startswith('{"metric'):
        d=json.loads(l); b=d["breakdown_ms_per_step"]
        print("  $t ms/step %.2f  lists %.2f force %.2f (bo %.2f nonbond %.2f bonded %.2f) ghost %.3f migrate %.3f  iters %.1f  pass %.4f" % (d["ms_per_step"], b["ms_lists"], b["ms_force"], b["ms_bo"], b["ms_nonbond"], b["ms_bonded"], b["ms_ghost_build"], b["ms_migrate"], d["qeq_iters_per_step"], d["roofline"]["avg_launch_ms"]))
PY
  find $O/prof_$t -name '*.csv' ! -name '*stats*' -delete; find $O/prof_$t -name '*.db' -delete
done
