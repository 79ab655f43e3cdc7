#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest.log | cut -c1-200
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
