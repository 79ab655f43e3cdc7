"""PCIe-inclusive rate of drop-in level 1 (host arrays in the reference's shapes on every call: rxmd_hip_QEq + rxmd_hip_FORCE) at the
bench workload, next to the device-resident rate that bench.py reports.  Prints one JSON line."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rxmd_amd
from rxmd_amd import system
INP = os.path.join(ROOT, "tests", "golden", "inputs")
ff = os.path.join(INP, "ffield_rdx")
names, frac, lat = system.read_xyz(os.path.join(INP, "rdx.xyz"))
lat3, rec = system.geninit(ff, names, frac, lat, mc=(18, 18, 18))
e = rxmd_amd.RxmdEngine(ff, lat3, qeq_mode=1)
e.set_atoms_rxff(rec)
e.QEq(); e.FORCE()
a = e.atoms()
n = len(a["q"]); nbuf = n + 1000
atype = np.zeros(nbuf); atype[:n] = a["type"] + a["gid"] * 1e-13
pos = np.zeros((nbuf, 3)); pos[:n] = a["pos"]
q = np.zeros(nbuf); q[:n] = a["q"]
t = []
for rep in range(3):
    pos[:n] += 1e-6 * (rep + 1)          # the driver moved the atoms since the last call
    t0 = time.perf_counter()
    e.QEq_arrays(atype, pos, q, n)
    f, pe = e.FORCE_arrays(atype, pos, q, n)
    t.append(time.perf_counter() - t0)
print(json.dumps({"workload": "RDX 18x18x18 = %d atoms" % n, "level1_ms_per_QEq_plus_FORCE": [round(1e3 * x, 1) for x in t],
                  "note": "host arrays up and down on every call; FORCE right after QEq re-uses ghosts and lists"}))
