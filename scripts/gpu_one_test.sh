#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 1200 python3 -m pytest tests/test_gpu_multirank.py -x -q -m gpu -k "test_vprocs_parity_vs_mpi_reference and example2" 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -40
