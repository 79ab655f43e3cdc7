#!/bin/bash
# end of round 3: full GPU suite, smoke, the default bench line as the driver runs it
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
timeout 3000 python3 -m pytest tests -q -m gpu 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Hostname\|^Librccl\|amdgpu.ids" | tail -4
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
python3 bench.py > $O/bench_default.log 2>$O/bench_default.err; grep '^{"metric' $O/bench_default.log > $O/bench_default.json
python3 - <<PY
import json
d=json.load(open("$O/bench_default.json")); r=d["roofline"]
print("default: steps/s", round(d["value"],2), "ms/step", round(d["ms_per_step"],2), r["kernel"], round(r["avg_launch_ms"],4), "frac", round(r["frac"],3), "placement first/kept", round(r["placement_search"]["pass_ms_first_placement"],4), round(r["placement_search"]["pass_ms_kept_placement"],4))
print(" alt_lex", round(d["alt_lex"]["steps_per_s"],1), [(o["workload"][:12], round(o["steps_per_s"],1), round(o["roofline"]["frac"],3)) for o in d["other_configs"]], "cpu", round(d["cpu_baseline"]["atom_steps_per_s"]))
PY
