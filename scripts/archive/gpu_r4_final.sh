#!/bin/bash
# round 4 final: PMC passes of the head commit (0b94490) -> profiles/kernel_traffic.json, then the default bench line (CPU baseline and all legs) and the same GPU legs under the kernel trace
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
COMMIT=0b94490 bash scripts/gpu_pmc_kernels.sh $1/pmc > $O/pmc.log 2>&1; tail -30 $O/pmc.log | cut -c1-170
find $O/pmc -name '*.csv' -size +2M -delete; find $O/pmc -name '*.db' -delete
cp $O/pmc/kernel_traffic.json profiles/kernel_traffic.json
python3 bench.py > $O/bench_default.log 2>&1; grep '^{"metric' $O/bench_default.log > $O/bench_default.json
python3 -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']
print('steps/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],2), 'pass', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), 'frac_real', r.get('frac_real_traffic'), 'frac_streamed', r.get('frac_streamed'), 'valu_floor', r.get('valu_floor_ms'), 'iters', d['qeq_iters_per_step'], 'traffic', r.get('traffic'))
print('steady', {k: (round(v,3) if isinstance(v,float) else v) for k,v in d.get('steady',{}).items() if k in ('ms_per_step','steps_per_s','qeq_iters_per_step','avg_pass_ms')})
print('alt', d.get('alt',{}).get('ms_per_step'), 'lex', d.get('alt_lex',{}).get('ms_per_step'), 'noplace', d.get('alt_no_placement_search',{}).get('ms_per_step'))
print('other', [(o.get('workload','')[:30], o.get('ms_per_step'), o.get('roofline',{}).get('avg_launch_ms'), o.get('roofline',{}).get('frac')) for o in d.get('other_configs',[])])
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))
print({k: round(v,2) for k,v in d['breakdown_ms_per_step'].items() if v}); print([(k['name'][:14], round(k['ms'],3), k.get('valu_floor_ms') and round(k['valu_floor_ms'],3), k.get('traffic') and round(k['traffic']/1e9,2)) for k in r['kernels']])"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-other-configs > $O/bench_prof.log 2>&1
grep '^{"metric' $O/bench_prof.log > $O/bench_prof.json
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; head -6 $O/kernel_stats.csv | cut -c1-120
find $O/prof -name '*.csv' ! -name '*stats*' -delete; find $O/prof -name '*.db' -delete
