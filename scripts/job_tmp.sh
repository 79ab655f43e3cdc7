cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_l; mkdir -p $O
RXMD_HIP_LIB=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_ot4.so timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -p no:cacheprovider -k "window_pass or tight_tolerance or benchmark_tolerance or full_size_properties_rdx or md_trajectory_at or perturbed_rdx_36k_against" > $O/pytest_ot4.log 2>&1
echo "pytest(ot4) rc=$?"; tail -4 $O/pytest_ot4.log | cut -c1-300
bash scripts/gpu_ab_libs.sh r05_l/ab "k_spmv_win" ot4 2>&1 | tail -12
