#!/bin/bash
# round 4: (1) the suite with EVERY engine buffer from hipDeviceMallocContiguous and poisoned (experiments library): does round 3's corruption come back, and as NaNs?
#          (2) default bench line with the prefetching window pass, (3) the poison test + the tolerance tests
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_HIP_LIB=$PWD/rxmd_amd/librxmd_hip_exp.so RXMD_CONTIG_ALLOC=0 RXMD_POISON_ALLOC=1 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_output.py -m gpu -q -p no:cacheprovider > $O/pytest_contig_poison.log 2>&1
echo "contig+poison pytest rc=$?"; tail -40 $O/pytest_contig_poison.log | cut -c1-250
RXMD_HIP_LIB=$PWD/rxmd_amd/librxmd_hip_exp.so RXMD_CONTIG_ALLOC=0 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_output.py -m gpu -q -p no:cacheprovider > $O/pytest_contig.log 2>&1
echo "contig pytest rc=$?"; tail -40 $O/pytest_contig.log | cut -c1-250
python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs > $O/bench.log 2>&1; grep '^{"metric' $O/bench.log > $O/bench.json
python3 -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step', d['ms_per_step'], 'pass', d['roofline']['avg_launch_ms'], 'iters', d['qeq_iters_per_step'], 'place', d['roofline']['placement_search'])"
timeout 1500 python -m pytest tests/test_gpu_parity.py::test_poisoned_allocations tests/test_gpu_output.py "tests/test_gpu_parity.py::test_window_pass_and_row_pass_are_the_same_operator" -q > $O/pytest_tol.log 2>&1
echo "pytest rc=$?"; tail -15 $O/pytest_tol.log | cut -c1-250
