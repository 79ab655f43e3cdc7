cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_b; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "tight or injected or forces_only or md_trajectory or other_force or bitwise or full_size_properties_rdx or published" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
bash scripts/gpu_ab_libs.sh r05_b/ab k_ehb 2>&1 | tail -12
