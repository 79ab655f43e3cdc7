#!/bin/bash
# placement spread of the window pass for different row strides of the 10 A list
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for s in 640 704 768; do
  echo "row stride $s entries"
  RXMD_S10=$s RXMD_PLACE_TRIES=1 python3 scripts/gpu_iso_copies.py 2>&1 | grep "process" | head -6 | cut -c1-210
done
