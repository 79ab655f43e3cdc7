#!/bin/bash
# The charge-free part of FORCE on its own stream (engine.h: bond_stream) against the one-stream order (RXMD_NO_BOND_OVERLAP=1): the tests that
# exercise it, then the default bench (headline + isQEq 2 leg) with and without it, interleaved.        usage: bash scripts/gpu_overlap_check.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout -k 10 1200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x \
  -k "bonded_chain_on_its_own or tight_tolerance_parity_vs_oracle or md_trajectory or bitwise_reproducible or migration_across or stress_accumulators or charges_every_third or poisoned or bond_tables_grow or row_stride" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -8 $O/pytest.log | cut -c1-220
for rep in 1 2; do
for v in "RXMD_X=0" "RXMD_NO_BOND_OVERLAP=1"; do
  env $v timeout -k 10 400 python3 bench.py --no-cpu-baseline --no-alt --no-steady --steps 20 --warmup 3 2>&1 | grep '^{"metric' > $O/bench_${v%%=*}_$rep.json
  python3 -c "
import json; d=json.load(open('$O/bench_${v%%=*}_$rep.json')); b=d['breakdown_ms_per_step']; l=d.get('alt_lex') or {}
print('$v', 'ms/step', round(d['ms_per_step'],2), 'pass ms', round(d['roofline']['avg_launch_ms'],4), 'iters', d['qeq_iters_per_step'], 'qeq', round(b['ms_qeq'],2), 'force(main stream)', round(b['ms_force'],2), 'bonded', round(b['ms_bonded'],2), 'nonbond', round(b['ms_nonbond'],2), '| lex ms/step', l.get('ms_per_step'), [(o.get('workload','')[:12], round(o.get('ms_per_step',0),2)) for o in d.get('other_configs',[])])"
done
done
