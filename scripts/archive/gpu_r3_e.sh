#!/bin/bash
# round 3: ring matrix pass -- correctness on small systems with the ring forced, then the isolated probes (each in its own process)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_RING_MIN_ROWS=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tight_tolerance_parity_vs_oracle or one_pass" > $O/pytest.log 2>&1
echo "pytest(small, ring forced) rc=$?"; tail -3 $O/pytest.log
RXMD_RING_MIN_ROWS=0 RXMD_RING_CYCLIC=0 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "tight_tolerance_parity_vs_oracle" > $O/pytest2.log 2>&1
echo "pytest(small, ring forced, contiguous) rc=$?"; tail -3 $O/pytest2.log
shift
for spec in "$@"; do timeout 300 python scripts/gpu_ring_probe.py $spec 2>&1 | tail -2; done
