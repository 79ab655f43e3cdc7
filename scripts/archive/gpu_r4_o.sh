#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 300 python3 scripts/gpu_r4_small.py 2>&1 | tail -2
rocprofv3 --kernel-trace --hip-trace --stats --output-format csv -d $O/prof -- python3 scripts/gpu_r4_small.py > $O/small.log 2>&1; tail -2 $O/small.log
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); head -12 $f | cut -c1-150
f=$(find $O/prof -name '*hip_api_stats.csv' -o -name '*hip_stats.csv' | head -1); head -14 $f | cut -c1-150
find $O/prof -name '*.csv' ! -name '*stats*' -delete; find $O/prof -name '*.db' -delete
