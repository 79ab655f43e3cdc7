cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out/r05_final
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -x -q -p no:cacheprovider -k "tight_tolerance or md_trajectory or benchmark_tolerance or one_pass or window_pass or lagrangian or charges_every or bitwise or full_size_properties_rdx or perturbed_rdx_36k_against or nve" > gpurun_out/r05_final/pytest_subset.log 2>&1
echo "pytest rc=$?"; tail -3 gpurun_out/r05_final/pytest_subset.log | cut -c1-200
COMMIT=c33d237 bash scripts/gpu_final.sh r05_final > gpurun_out/r05_final.log 2>&1; tail -70 gpurun_out/r05_final.log | cut -c1-260
