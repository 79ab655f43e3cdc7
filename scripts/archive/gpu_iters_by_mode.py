"""CG iterations per MD step, qeq_mode 0 vs 1 vs the oracle (reference algebra on CPU), same system and tolerance (1e-7)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_api as oa
import rxmd_amd
from rxmd_amd import system

mc = tuple(int(x) for x in (sys.argv[1:4] or (3, 3, 3)))
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 12
ff, names, frac, lat = oa.make_system("rdx168")
lat3, rec = system.geninit(ff, names, frac, lat, mc=mc)
res = {}
for mode in (0, 1):
    e = rxmd_amd.RxmdEngine(ff, lat3, qeq_mode=mode)
    e.set_atoms_rxff(rec)
    it = [e.QEq()[0]]; e.FORCE()
    for s in range(nsteps):
        e.step(1); it.append(e.stats()["qeq_iters_last"])
    res[mode] = it
    print("mode", mode, it, "mean", np.mean(it[1:]))
    e.close()
if np.prod(mc) <= 27:
    lat2, ranks = oa.geninit(names, frac, lat, oa.ffield_names(ff), mc=mc)
    o = oa.Oracle(ff, lat2, ranks)
    it = [o.qeq()]; o.force()
    for s in range(nsteps):
        o.step(1); it.append(o.L.rxo_qeq_iters(o.w))
    print("oracle", it, "mean", np.mean(it[1:]))
