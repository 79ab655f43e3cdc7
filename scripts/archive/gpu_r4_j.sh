#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
RXMD_HIP_LIB=$PWD/rxmd_amd/librxmd_hip_exp.so timeout 900 python3 scripts/gpu_r4_listprobe.py 2>&1 | grep probe
python3 bench.py --no-cpu-baseline --steps 20 --warmup 5 --no-alt --no-other-configs --no-steady > $O/bench.log 2>&1; grep '^{"metric' $O/bench.log > $O/bench.json
python3 -c "
import json; d=json.load(open('$O/bench.json')); print('ms/step', d['ms_per_step'], 'pass', d['roofline']['avg_launch_ms'], 'iters', d['qeq_iters_per_step']); print(d['breakdown_ms_per_step']); print([(k['name'][:20], round(k['ms'],3)) for k in d['roofline']['kernels']])"
timeout 900 python -m pytest tests/test_gpu_parity.py -q -x -k "step0 or 36k or poison or energies or e3b or angle" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
