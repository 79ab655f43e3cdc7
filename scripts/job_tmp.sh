cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_o; mkdir -p $O
timeout -k 10 1800 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -4 $O/pytest.log | cut -c1-220
timeout -k 10 200 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
