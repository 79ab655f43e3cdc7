cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_e; mkdir -p $O
timeout -k 10 300 python3 scripts/gpu_variants_check.py default ehbdbg > $O/variants.log 2>&1; cut -c1-200 $O/variants.log | tail -24
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q -p no:cacheprovider -k "fused_self or documented_bound or tight_tolerance or injected or forces_only or migration_across or bitwise or window_slots or other_force" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
bash scripts/gpu_ab_libs.sh r05_e/ab "k_ehb|k_list10|k_win_columns|k_bonded_list|k_bond_csr|k_seg_|k_slab|k_append|k_to_" r4k 2>&1 | tail -60
