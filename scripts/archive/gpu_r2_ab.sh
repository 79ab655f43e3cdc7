#!/bin/bash
# round 2 A/B: new 10 A sweep (z-trimmed columns, FP32 first test) and the 16-bit column stream of the matrix pass, same box, same run
O=gpurun_out/$1; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -x -q -k "tight_tolerance or variants or md_trajectory_tight or migration or pqeq_step0 or full_size_properties_rdx or row_stride or error_codes or bitwise" > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
B="python bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline"
$B > $O/b_default.json 2> $O/b_default.err
RXMD_LIST_NO_FP32=1 $B > $O/b_nofp32.json 2>/dev/null
RXMD_SPMV_IDX32=1 $B > $O/b_idx32.json 2>/dev/null
RXMD_SPMV_IDX32=1 RXMD_LIST_NO_FP32=1 $B > $O/b_idx32_nofp32.json 2>/dev/null
tail -3 $O/pytest.log
