#!/bin/bash
# round 2, call f: ghost-column sums only on boundary rows (A/B), row stride of the 10 A list (A/B), full GPU suite
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
B="python3 bench.py --steps 20 --warmup 5 --no-alt --no-cpu-baseline"
$B > $O/b_warm.json 2> $O/b_warm.err
$B > $O/b_default.json 2>/dev/null
RXMD_SPMV_ALL_ROWS_GHOST=1 $B > $O/b_allghost.json 2>/dev/null
RXMD_S10=448 $B > $O/b_s448.json 2>/dev/null
RXMD_S10=512 $B > $O/b_s512.json 2>/dev/null
$B > $O/b_default2.json 2>/dev/null
python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
tail -8 $O/pytest.log
