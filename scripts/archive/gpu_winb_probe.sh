#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for p in 1 2 3 4 0; do
  RXMD_WINB_PROBE=$p RXMD_SPMV_WIN=0 python3 bench.py --no-cpu-baseline --steps 6 --warmup 2 --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['breakdown_ms_per_step']
print('probe=$p winbuild', round(b['ms_k_winbuild'],3))"
done
