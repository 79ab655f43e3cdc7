import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import rxmd_amd
from rxmd_amd import system
import oracle_api as oa
for case, mc in (("rdx168", (18, 18, 18)), ("rdx168", (5, 5, 5)), ("ice644", (6, 4, 4)), ("example1", (2, 3, 5)), ("pbt2272", (1, 1, 1)), ("fes576", (1, 1, 1))):
    try:
        ff, names, frac, lat = oa.make_system(case)
        lat3, rec = system.geninit(ff, names, frac, lat, mc=mc)
        e = rxmd_amd.RxmdEngine(ff, lat3, QEq_tol=1e-7, NMAXQEq=500); e.set_atoms_rxff(rec); e.QEq(); e.FORCE(); e.step(10)
        st = e.stats(); print(case, mc, "atoms", st["natoms"], "max bond list", st["max_nb"], "mean", st["nbonds"] / (st["natoms"] + st["nghost_force"]), flush=True); e.close()
    except Exception as ex: print(case, "failed", ex)
