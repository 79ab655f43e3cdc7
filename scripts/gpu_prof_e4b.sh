#!/bin/bash
# kernel trace of FORCE with the torsions visited once / twice (RDX 979,776 atoms, one stream): usage  bash scripts/gpu_prof_e4b.sh <tag>
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
cat > /tmp/e4b_force.py <<'PY'
import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rxmd_amd
from rxmd_amd import system
INP = os.path.join(ROOT, "tests", "golden", "inputs")
names, frac, lat = system.read_xyz(os.path.join(INP, "rdx.xyz"))
ff = os.path.join(INP, "ffield_rdx")
lat_s, rec = system.geninit(ff, names, frac, lat, mc=(18, 18, 18))
e = rxmd_amd.RxmdEngine(ff, lat_s, QEq_tol=1e-7, NMAXQEq=500, device=0, qeq_mode=1)
e.set_atoms_rxff(rec); e.QEq()
for _ in range(6): e.FORCE()
e.close()
PY
for once in 0 1; do
  export RXMD_E4B_ONCE=$once RXMD_NO_BOND_OVERLAP=1 RXMD_PLACE_TRIES=1
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/once$once -- python3 /tmp/e4b_force.py > $O/once$once.log 2>&1
  f=$(find $O/once$once -name "*kernel_stats.csv" | head -1)
  echo "== RXMD_E4B_ONCE=$once"
  [ -n "$f" ] && { grep -E "k_e4b|k_e3b|k_ehb_sweep|k_bond_force|k_ccbnd" "$f" | python3 -c "
import sys,csv
for r in csv.reader(sys.stdin): print(r[0][:48].ljust(48), *r[1:5])"; cp "$f" $O/kernel_stats_once$once.csv; } < /dev/null
done
