cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_h; mkdir -p $O
timeout -k 10 1800 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -28 $O/pytest.log | cut -c1-220
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
bash scripts/gpu_ab_libs.sh r05_h/ab "k_ehb|k_list10" 2>&1 | grep -v "^  r4k\|^  default k_seg" | tail -12
