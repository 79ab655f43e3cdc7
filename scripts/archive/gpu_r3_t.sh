#!/bin/bash
# round 3: a selection of GPU tests only.  usage: bash scripts/gpu_r3_t.sh <tag> <pytest args...>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O; shift
python -m pytest "$@" > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
tail -30 $O/pytest.log; cat $O/rc.txt
