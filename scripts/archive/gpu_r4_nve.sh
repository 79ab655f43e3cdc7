#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 600 python3 scripts/gpu_nve_drift.py 400 2>&1 | tee $O/nve.log | tail -8
