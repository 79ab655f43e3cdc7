"""NVE energy conservation at the headline size: RDX 979,776 atoms, dt 0.25 fs, QEq tol 1e-7, qeq_mode 1; prints TE/atom every 20 steps"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_api as oa
import rxmd_amd
from rxmd_amd import system

nsteps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
ff, names, frac, lat = oa.make_system("rdx168")
lat3, rec = system.geninit(ff, names, frac, lat, mc=(18, 18, 18))
rng = np.random.default_rng(1)
e = rxmd_amd.RxmdEngine(ff, lat3, qeq_mode=1)
# 300 K Maxwell velocities (units of the engine's rec10: as DAT/rxff.bin holds them) are not needed for a drift check: start cold, the
# crystal is off its ReaxFF minimum and heats up by itself
e.set_atoms_rxff(rec)
e.QEq(); e.FORCE()
n = len(rec)
t0 = time.time()
rows = []
for s in range(0, nsteps + 1, 20):
    en = e.energy()
    rows.append((s, (en["PE"][0] + en["KE"]) / n, en["PE"][0] / n, en["KE"] / n))
    print("step %4d  TE/atom %.8f  PE/atom %.8f  KE/atom %.8f kcal/mol" % rows[-1], flush=True)
    if s < nsteps:
        e.step(20)
dt = time.time() - t0
te = np.array([r[1] for r in rows])
print("steps %d in %.1f s (%.1f ms/step incl. energy reads); TE drift %.3e kcal/mol/atom over the run, max |TE - TE0| %.3e" % (nsteps, dt, 1e3 * dt / nsteps, te[-1] - te[0], np.abs(te - te[0]).max()))
