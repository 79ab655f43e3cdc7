cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/r05_n; mkdir -p $O
timeout -k 10 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_output.py tests/test_gpu_dropin.py tests/test_gpu_multirank.py -x -q -p no:cacheprovider -k "md_ or trajectory or thermo or velocity or poison or nve or bitwise or output or xyz or bnd or rxff or minimiser or dropin or reference_driver or example or self_loop or lagrangian or charges_every or tight_tolerance or published or reference_shaped or error_codes or bench_launches_its" > $O/pytest.log 2>&1
echo "pytest rc=$?"; tail -5 $O/pytest.log | cut -c1-300
timeout -k 10 600 bash scripts/gpu_stepseq.sh r05_n/seq > $O/stepseq.log 2>&1; grep -c . $O/seq/stepseq.txt; grep "step length" $O/seq/stepseq.txt
timeout -k 10 400 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-alt --no-other-configs 2>/dev/null | grep '^{"metric' > $O/bench.json; python3 -c "
import json; d=json.load(open('$O/bench.json')); b=d['breakdown_ms_per_step']
print('ms/step %.2f iters %.1f pass %.4f steady %.2f (K %.1f)' % (d['ms_per_step'], d['qeq_iters_per_step'], d['roofline']['avg_launch_ms'], d['steady']['ms_per_step'], d['steady']['qeq_iters_per_step']), {k: round(v,2) for k,v in b.items() if v})"
