#!/bin/bash
# round 4, first lease: the new launch-path / 8-rank tests, then the default bench under rocprofv3 --kernel-trace --stats as a baseline
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_multirank.py::test_bench_launches_its_own_ranks "tests/test_gpu_scale.py::test_perturbed_rdx_36k_on_several_ranks_against_the_multi_rank_oracle" -q -x > $O/pytest_new.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest_new.log
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs > $O/bench_prof.log 2>&1
grep '^{"metric' $O/bench_prof.log > $O/bench_prof.json
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; head -30 $O/kernel_stats.csv
find $O/prof -name '*.csv' ! -name '*stats*' -delete; find $O/prof -name '*.db' -delete
