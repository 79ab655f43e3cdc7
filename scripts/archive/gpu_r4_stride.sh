#!/bin/bash
# the window pass against the row stride of the 10 A list (experiments library, RXMD_S10), three processes per stride, strides interleaved: is the placement effect an aliasing of the stride?
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
for rep in 1 2 3; do for s10 in 448 512 576 640 704 768 832 896 960 1024 1088; do
RXMD_HIP_LIB=$PWD/rxmd_amd/librxmd_hip_exp.so RXMD_S10=$s10 python3 bench.py --no-cpu-baseline --steps 10 --warmup 3 --no-alt --no-other-configs --no-steady 2>/dev/null | grep '^{"metric' | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; p=r['placement_search']
print('S10 $s10 rep $rep: loop %.4f first %.4f kept %.4f ms/step %.2f nonbond %.3f list10 %.3f' % (r['avg_launch_ms'], p['pass_ms_first_placement'], p['pass_ms_kept_placement'], d['ms_per_step'], d['breakdown_ms_per_step']['ms_nonbond'], r['kernels'][0]['ms']))"
done; done | tee $O/stride.log
