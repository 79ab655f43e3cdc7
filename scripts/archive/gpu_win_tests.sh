#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
timeout 3200 python3 -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl" | tail -30
