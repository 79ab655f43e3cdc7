#!/bin/bash
# round 3: the profile set judged from profiles/ -- kernel statistics of the default bench, the default bench line, the other workloads,
# the self-loop run of the multi-GPU code path (exchange timers), the two opt-in matrix-pass kernels
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_scale.py -m gpu -q -k "two_ranks" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-alt --no-other-configs > $O/prof_bench.log 2>&1
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/kernel_stats.csv
grep '^{"metric' $O/prof_bench.log > $O/bench_under_profiler.json
python3 bench.py --no-cpu-baseline > $O/bench_default_nocpu.log 2>&1; grep '^{"metric' $O/bench_default_nocpu.log > $O/bench_default_nocpu.json
for w in water sicnp; do python3 bench.py --workload $w --steps 8 --warmup 2 --no-cpu-baseline --no-alt 2>/dev/null | grep '^{"metric' > $O/bench_$w.json; done
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29517 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0
for d in 0 1; do
RXMD_HALO_DIRECT=$d RXMD_BENCH_FORCE_DIST=1 RXMD_FORCE_STAGED=1 RXMD_FORCE_REMOTE=1 timeout 900 python bench.py --gpus 1 --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' > $O/bench_selfloop_direct$d.json
done
for v in "RXMD_SPMV_RING=1" "RXMD_SPMV_DMA=1 RXMD_DMA_WPB=8"; do
env $v python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' > "$O/bench_$(echo $v | tr ' =' '__').json"
done
python3 - <<PY
import json,glob,csv
for f in sorted(glob.glob("$O/bench_*.json")):
    try: d=json.load(open(f))
    except Exception as e: print(f, "unreadable"); continue
    print(f.split("/")[-1], "ms/step", round(d["ms_per_step"],2), "steps/s", round(d["value"],2), "spmv", round(d["roofline"]["avg_launch_ms"],4), "frac", round(d["roofline"]["frac"],3), "iters", round(d["qeq_iters_per_step"],1), {k:round(v,3) for k,v in d["breakdown_ms_per_step"].items() if k.startswith("ms_halo") or k in("ms_ghost_build","ms_migrate","ms_allreduce","ms_fold")} if "selfloop" in f else "")
for i,r in enumerate(csv.DictReader(open("$O/kernel_stats.csv"))):
    if i<16: print("  ",r["Name"][:60],r["Calls"],r["AverageNs"],r["Percentage"])
PY
