cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_j; mkdir -p $O
timeout -k 10 400 python3 scripts/gpu_variants_check.py default nbnt > $O/variants.log 2>&1; cut -c1-200 $O/variants.log | tail -12
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "tight_tolerance or injected or forces_only or bitwise or other_force or stress or full_size_properties_rdx" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-300
bash scripts/gpu_ab_libs.sh r05_j/ab "k_e3b|k_nonbond" nbnt 2>&1 | tail -12
