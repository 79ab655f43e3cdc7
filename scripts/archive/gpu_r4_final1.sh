#!/bin/bash
# round 4 final, part 1: the whole GPU suite, then the PMC passes of the head commit (685572c)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 2700 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=6 > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -14 $O/pytest.log | cut -c1-200
COMMIT=685572c bash scripts/gpu_pmc_kernels.sh $1/pmc > $O/pmc.log 2>&1; tail -26 $O/pmc.log | cut -c1-200
find $O/pmc -name '*.csv' -size +2M -delete; find $O/pmc -name '*.db' -delete
