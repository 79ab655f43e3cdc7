#!/bin/bash
# A/B of two builds on one box: rxmd_amd/librxmd_hip_old.so against rxmd_amd/librxmd_hip.so, bench main leg only, alternating
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_scale.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log | cut -c1-200
for v in _old "" _old ""; do
  export RXMD_HIP_LIB=$GRAFT_REPO_ROOT/rxmd_amd/librxmd_hip$v.so
  python3 bench.py --no-cpu-baseline --no-other-configs --no-alt --no-steady > $O/bench$v.log 2>&1
  grep '^{"metric' $O/bench$v.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; b=d['breakdown_ms_per_step']
print('lib$v', 'ms/step', round(d['ms_per_step'],3), 'pass', round(r['avg_launch_ms'],4), 'iters', d['qeq_iters_per_step'], {k['name'][:24]: round(k['ms'],3) for k in r['kernels'] if 'list' in k['name'] or 'nonbond' in k['name']}, 'pe', d['energy_per_atom'])"
done
