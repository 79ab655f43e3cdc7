#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_HIP_LIB=$PWD/rxmd_amd/librxmd_hip_exp.so timeout 600 python3 scripts/gpu_r4_iso.py > $O/iso.log 2>&1; tail -5 $O/iso.log | cut -c1-600
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x --durations=5 -k "md_trajectory or migration_across or stress_acc" > $O/pytest.log 2>&1; tail -10 $O/pytest.log | cut -c1-200
