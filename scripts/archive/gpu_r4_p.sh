#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_HIP_LIB=$PWD/rxmd_amd/librxmd_hip_exp.so timeout 600 python3 scripts/gpu_r4_iso.py > $O/iso.log 2>&1; tail -4 $O/iso.log | cut -c1-400
timeout 1500 python -m pytest tests/test_gpu_scale.py tests/test_gpu_multirank.py tests/test_gpu_parity.py -q -x -k "36k or self_loop or bench_launches or bond_tables_grow or window_slots or window_pass_and" > $O/pytest.log 2>&1; tail -3 $O/pytest.log | cut -c1-200
