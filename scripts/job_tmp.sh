cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_g; mkdir -p $O
timeout -k 10 300 python3 scripts/gpu_variants_check.py default > $O/variants.log 2>&1; cut -c1-200 $O/variants.log | tail -6
timeout -k 10 1200 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "fused_self or documented_bound or tight_tolerance or injected or forces_only or migration_across or bitwise or window_slots or other_force or skewed or bond_tables_grow or error_codes or pqeq or stress or torsion_kernel" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
bash scripts/gpu_ab_libs.sh r05_g/ab "k_ehb|k_bonded_list|k_bond_csr|k_sorted_pos|k_list10|k_nonbond" r4k 2>&1 | tail -40
