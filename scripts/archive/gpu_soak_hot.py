"""Hot RDX (Maxwell velocities, default 2500 K) for many steps, engine against the CPU oracle at tight QEq tolerance: bonds break and
form, the bond lists change length (the torsion kernel switches instance when a list passes 15), atoms migrate through the box faces.
usage: gpu_soak_hot.py [steps=120] [T=2500] [qeq_mode=1]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_api as oa
import rxmd_amd
from rxmd_amd import system
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
T = float(sys.argv[2]) if len(sys.argv) > 2 else 2500.0
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 1
kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
ff, names, frac, lat = oa.make_system("rdx222")
ffn = oa.ffield_names(ff)
lat2, ranks = oa.geninit(names, frac, lat, ffn, mc=(2, 2, 2))
n = len(ranks[0]["type"])
mass = {"C": 12.0, "H": 1.008, "O": 15.999, "N": 14.0}
m = np.array([mass[ffn[t - 1]] for t in ranks[0]["type"]])
rng = np.random.default_rng(7)
kB = 0.0019872041                                   # kcal/mol/K
# velocities in the engine's units: the oracle and the engine take the same numbers (rxff.bin velocity columns); scale so that
# sum m v^2 / (3 N kB) = T with the reference's KE convention 0.5 m v^2 (init.F90: hmas) -- only the ratio matters for a soak
v = rng.normal(0.0, 1.0, (n, 3)) * np.sqrt(kB * T / m)[:, None] * 1e-2
v -= (m[:, None] * v).sum(0) / m.sum()
o = oa.Oracle(ff, lat2, ranks, v0=[v], **kw)
lat3, rec = system.geninit(ff, names, frac, lat, mc=(2, 2, 2))
rec = np.array(rec).reshape(-1, 10); rec[:, 3:6] = v
e = rxmd_amd.RxmdEngine(ff, lat3, qeq_mode=mode, **kw)
e.set_atoms_rxff(rec)
o.qeq(); o.force(); e.QEq(); e.FORCE()
worst = 0.0
for s0 in range(0, steps, 20):
    k = min(20, steps - s0)
    o.step(k); e.step(k)
    a = e.atoms()
    og = o.gids(); order_ok = (a["gid"] == og).all()
    if not order_ok:                                   # compare by gid if the local order ever differs
        ia = np.argsort(a["gid"]); io = np.argsort(og)
    else:
        ia = io = np.arange(len(og))
    dx = np.abs(a["pos"][ia] - o.pos()[io]); L = np.array(lat3[:3]); dx = np.minimum(dx, np.abs(dx - L)).max()
    fo = o.forces()[io]; fe = a["f"][ia]
    frms = np.sqrt((fo ** 2).mean()); df = (np.abs(fe - fo).max(axis=1) / np.maximum(np.abs(fo).max(axis=1), frms)).max()
    dq = np.abs(a["q"][ia] - o.charges()[io]).max()
    nb, _ = o.bonds(); cnt = (nb[:len(og)] > 0).sum(axis=1)
    print("step %4d  order %s  max|dx| %.2e A  f_err %.2e  max|dq| %.2e  KE %.5f / %.5f  longest bond list %d" %
          (s0 + k, "same" if order_ok else "DIFFERS", dx, df, dq, e.energy()["KE"], o.kinetic(), cnt.max()), flush=True)
    worst = max(worst, df)
print("worst f_err %.2e" % worst)
