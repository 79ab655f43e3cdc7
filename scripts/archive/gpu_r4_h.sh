#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for v in "X=1" "RXMD_NONBOND_WIN=0" "RXMD_SPMV_WIN=0"; do
  echo "== $v"; env $v RXMD_POISON_ALLOC=1 timeout 600 python3 tests/poison_worker.py 2>&1 | tail -6 | cut -c1-400
done
