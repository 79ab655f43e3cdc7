#!/bin/bash
# What a round leaves under profiles/: PMC passes of the head (-> kernel_traffic.json), the default bench line with every leg and the CPU baseline,
# the same GPU legs under the kernel trace, and kernel statistics + FETCH / WRITE of the matrix pass for the water and the SiC-NP (PQEq) workloads.
# usage: COMMIT=<short hash> bash scripts/gpu_final.sh <tag>        every run under its own time limit
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
COMMIT=${COMMIT:-unknown} timeout -k 10 1500 bash scripts/gpu_pmc_kernels.sh $1/pmc > $O/pmc.log 2>&1; tail -26 $O/pmc.log | cut -c1-170
find $O/pmc -name '*.csv' -size +2M -delete; find $O/pmc -name '*.db' -delete
[ -s $O/pmc/kernel_traffic.json ] && cp $O/pmc/kernel_traffic.json profiles/kernel_traffic.json
timeout -k 10 900 python3 bench.py --full-line $O/bench_default_full.json > $O/bench_default.log 2>&1; grep '^{"metric' $O/bench_default.log > $O/bench_default_line.json; cp $O/bench_default_full.json $O/bench_default.json
echo "stdout line: $(wc -c < $O/bench_default_line.json) bytes"
python3 -c "
import json; d=json.load(open('$O/bench_default.json')); r=d['roofline']
print('steps/s', round(d['value'],3), 'ms/step', round(d['ms_per_step'],2), 'pass', round(r['avg_launch_ms'],4), 'frac', round(r['frac'],3), 'frac_real', r.get('frac_real_traffic'), 'iters', d['qeq_iters_per_step'])
print('steady', {k: (round(v,3) if isinstance(v,float) else v) for k,v in d.get('steady',{}).items() if k in ('ms_per_step','steps_per_s','qeq_iters_per_step','avg_pass_ms')})
print('alt', d.get('alt',{}).get('ms_per_step'), 'lex', d.get('alt_lex',{}).get('ms_per_step'), 'bond_streams', d.get('alt_bond_streams',{}).get('ms_per_step'), 'noplace', d.get('alt_no_placement_search',{}).get('ms_per_step'), 'placement', r.get('placement_search'))
print('other', [(o.get('workload','')[:30], o.get('ms_per_step'), o.get('roofline',{}).get('avg_launch_ms'), o.get('roofline',{}).get('frac')) for o in d.get('other_configs',[])])
print('cpu', d.get('cpu_baseline',{}).get('value'), d.get('cpu_baseline',{}).get('cores'))
print({k: round(v,2) for k,v in d['breakdown_ms_per_step'].items() if v}); print([(k['name'][:14], round(k['ms'],3), k.get('bound'), k.get('frac_of_bound') and round(k['frac_of_bound'],2)) for k in r['kernels']])"
timeout -k 10 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-other-configs --full-line $O/bench_prof.json > $O/bench_prof.log 2>&1
f=$(find $O/prof -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats.csv; head -24 $O/kernel_stats.csv | cut -c1-130
find $O/prof -name '*.csv' ! -name '*stats*' -delete; find $O/prof -name '*.db' -delete
for w in water sicnp; do
  timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$w -- python3 bench.py --workload $w --steps 6 --warmup 2 --no-cpu-baseline --no-alt --no-steady --full-line $O/bench_$w.json > $O/bench_$w.log 2>&1
  f=$(find $O/prof_$w -name '*kernel_stats.csv' | head -1); cp $f $O/kernel_stats_$w.csv; head -8 $O/kernel_stats_$w.csv | cut -c1-130
  find $O/prof_$w -name '*.csv' ! -name '*stats*' -delete; find $O/prof_$w -name '*.db' -delete
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout -k 10 400 rocprofv3 --pmc $ctr --kernel-include-regex "k_spmv" --output-format csv -d $O/pmc_${w}_$ctr -- python3 bench.py --workload $w --steps 1 --warmup 0 --no-cpu-baseline --no-alt --no-steady > $O/pmc_${w}_$ctr.log 2>&1
  done
  python3 - <<PY
import csv, glob, collections, json
res = {}
for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
    d = collections.defaultdict(list)
    for f in glob.glob("$O/pmc_${w}_%s/**/*counter_collection.csv" % ctr, recursive=True):
        for r in csv.DictReader(open(f)): d[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rxmd::", "")].append(float(r["Counter_Value"]))
    for k, v in d.items(): res.setdefault(k, {})[ctr] = sum(v) / len(v); res[k]["launches"] = len(v)
for k, v in res.items():
    if "FETCH_SIZE" in v and "WRITE_SIZE" in v: v["hbm_bytes_per_launch"] = (2.0 * v["FETCH_SIZE"] + v["WRITE_SIZE"]) * 1024.0
json.dump(res, open("$O/pass_traffic_$w.json", "w"), indent=1); print("$w", {k: round(v.get("hbm_bytes_per_launch", 0) / 1e9, 3) for k, v in res.items()})
PY
  find $O/pmc_${w}_* -name '*.csv' -delete; find $O/pmc_${w}_* -name '*.db' -delete
done
