import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as oa
from test_gpu_parity import _engine
import rxmd_amd
kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
e = _engine("rdx168", (1, 1, 1), **kw)
e.QEq(); pe = e.FORCE(); a = e.atoms()
E0 = pe[0]; f = a["f"].copy(); pos0 = a["pos"].copy()
print("E0", E0, "pdf", (f * f).sum())
rec0 = e.get_atoms_rxff()
L = np.array(e.lattice[:3])
s = 1e-2 / 168
for it in range(4):
    s *= 2
    rec = rec0.copy(); rec[:, 0:3] = np.mod((pos0 + s * f) / L, 1.0)
    e2 = rxmd_amd.RxmdEngine(oa.make_system("rdx168")[0], e.lattice, **kw)
    e2.set_atoms_rxff(rec); e2.QEq(); p2 = e2.FORCE()
    print(it, s, p2[0], p2[0] - E0)
    e2.close()
loops, pef, ev = e.minimise(ftol=1e-5)
print("minimise: loops", loops, "pe", pef, "evals", ev, "dE", pef - E0)
a2 = e.atoms()
o = np.argsort(a2["gid"])
print("max displacement", np.abs(a2["pos"][o] - pos0[np.argsort(a["gid"])]).max())
