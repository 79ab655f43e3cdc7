#!/bin/bash
# round 4: (1) window-pass variants back to back (experiments library), (2) the contiguous-allocation probe, (3) the tests touched by the tolerance change
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_HIP_LIB=$PWD/rxmd_amd/librxmd_hip_exp.so timeout 900 python3 scripts/gpu_r4_iso.py > $O/iso.log 2>&1; tail -6 $O/iso.log
timeout 300 scripts/micro/contig_small > $O/contig.log 2>&1; cat $O/contig.log
timeout 1500 python -m pytest tests/test_gpu_multirank.py::test_vprocs_parity_vs_mpi_reference tests/test_gpu_output.py "tests/test_gpu_parity.py::test_window_pass_and_row_pass_are_the_same_operator" -q -x > $O/pytest_tol.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest_tol.log | cut -c1-250
