#!/bin/bash
# round 3: isolated ring-pass probes over library variants.  usage: gpu_r3_f.sh "<lib tags>" <specs...>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
tags=$1; shift
for t in $tags; do
  L=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip_$t.so; [ "$t" = "base" ] && L=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip.so
  echo "== $t"; RXMD_HIP_LIB=$L timeout 300 python scripts/gpu_ring_probe.py "$@" 2>&1 | tail -$#
done
