#!/bin/bash
# the whole GPU suite with every engine buffer poisoned (RXMD_POISON_ALLOC=1): which kernel reads what nobody wrote?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_POISON_ALLOC=1 timeout 3000 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_poison.log 2>&1
echo "pytest rc=$?"; tail -60 $O/pytest_poison.log | cut -c1-300
