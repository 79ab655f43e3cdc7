#!/bin/bash
# the placement search with its losers kept allocated: all ten draws of fresh processes on one box with the addresses of the streams (RXMD_PLACE_VERBOSE, RXMD_PLACE_ALL)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for r in 1 2 3 4; do
  RXMD_PLACE_VERBOSE=1 RXMD_PLACE_ALL=1 python3 bench.py --no-cpu-baseline --no-other-configs --no-alt --no-steady --steps 4 --warmup 2 > $O/bench$r.log 2> $O/err$r.log
  echo "process $r"; grep 'placement draw' $O/err$r.log | sed 's/.*draw //'
done
