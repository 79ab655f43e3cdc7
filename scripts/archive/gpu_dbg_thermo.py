import os, sys, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle_api as oa, rxmd_amd, tempfile
for win in ("1", "0", "1"):
    os.environ["RXMD_SPMV_WIN"] = win
    for mode, kw in [(4, dict(vsfact=0.9)), (5, dict(treq=300.0))]:
        g = np.load(os.path.join(oa.GOLD, "rdx168_thermo%d.npz" % mode))
        ff = oa.make_system("rdx168")[0]
        lat, vp, step0, recs = oa.parse_rxff(g["restart_rxff"])
        e = rxmd_amd.RxmdEngine(ff, lat, QEq_tol=1e-12, NMAXQEq=2000)
        e.set_atoms_rxff(recs[0])
        its = [e.QEq()[0]]; e.FORCE()
        for nstep in range(7):
            if nstep % 3 == 0: e.thermostat(mode, **kw)
            e.step(1); its.append(e.stats()["qeq_iters_last"])
        p = tempfile.mktemp(); e.write_rxff(p, current_step=step0 + 7); st = e.stats(); e.close()
        _, _, s1, r1 = oa.parse_rxff(np.frombuffer(open(p, "rb").read(), np.uint8)); _, _, s2, r2 = oa.parse_rxff(g["final_rxff"])
        a, b = r1[0], r2[0]
        print("win", win, "mode", mode, "in use", st["win_in_use"], "iters", its, "pos", np.abs(a[:, 0:3] - b[:, 0:3]).max(), "vel rel", np.abs(a[:, 3:6] - b[:, 3:6]).max() / np.abs(b[:, 3:6]).max(), "q", np.abs(a[:, 6] - b[:, 6]).max())
