#!/bin/bash
# the whole GPU suite with every device allocation filled with 0xFF (NaN doubles, index -1) and the per-step scratch re-poisoned each step
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_POISON_ALLOC=1 timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_poisoned.log 2>&1
echo "pytest rc=$?"; tail -6 $O/pytest_poisoned.log | cut -c1-200
