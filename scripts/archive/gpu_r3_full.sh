#!/bin/bash
# round 3: the full GPU suite, then the default bench line as the driver runs it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q > $O/pytest.log 2>&1
echo "pytest rc=$?" > $O/rc.txt
tail -8 $O/pytest.log; cat $O/rc.txt
( time python3 bench.py > $O/bench.json 2>$O/bench.err ) 2>&1 | tail -3
python3 - <<PY
import json
d=json.loads([l for l in open("$O/bench.json") if l.startswith('{"metric')][-1])
print("value", d["value"], "ms/step", d["ms_per_step"], "frac", d["roofline"]["frac"], "iters", d["qeq_iters_per_step"])
for k in d["roofline"]["kernels"]: print("  ", k["name"][:40], round(k["ms"],4), "ms frac", round(k["frac"],3))
print("cpu", {k:v for k,v in d.get("cpu_baseline",{}).items() if k!="other_samples"})
print("alt", d.get("alt")); print("alt_lex", d.get("alt_lex"))
for o in d.get("other_configs",[]): print("other", o.get("workload"), o.get("steps_per_s"), o.get("roofline",{}).get("frac"), o.get("error"))
PY
