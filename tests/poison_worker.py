"""Run by test_gpu_parity.py::test_poisoned_allocations in a process of its own (RXMD_POISON_ALLOC is read once per process): every engine
buffer starts as 0xFF bytes -- NaN for doubles, -1 for indices -- and the per-step scratch is filled with the pattern again before every rebuild.
A kernel that read an element nobody wrote this step would turn charges, forces or energies into NaN or trap on an index."""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import oracle_api as oa
from test_gpu_parity import _engine, _oracle, q_err, f_err, e_err

assert os.environ.get("RXMD_POISON_ALLOC") == "1"
kw = dict(QEq_tol=1e-12, NMAXQEq=2000)
for case, mc, extra, steps in (("rdx222", (2, 2, 2), {}, 3), ("sicnp", (1, 1, 1), dict(pqeq=oa.PQEQ_SICNP), 2)):
    o = _oracle(case, mc, **kw, **extra)                        # (the oracle once per case: most of this test's time is its PQEq solve on the host)
    if extra:
        o.set_pqeq_clean(1)
    o.qeq(); o.force()
    io = np.argsort(o.gids())
    q0, f0, pe0 = o.charges()[io].copy(), o.forces()[io].copy(), o.energy().copy()
    o.step(steps)
    io1 = np.argsort(o.gids())
    pos1, q1, f1 = o.pos()[io1].copy(), o.charges()[io1].copy(), o.forces()[io1].copy()
    for qeq_mode in (1, 0):
        e = _engine(case, mc, qeq_mode=qeq_mode, **kw, **extra)
        e.QEq(); pe = e.FORCE()
        tap = e.debug(14, cap=2)
        assert tap[0] == 1.0 and np.isnan(tap[1]), tap          # the pattern is on, and an element no kernel writes still holds it
        a = e.atoms()
        assert np.isfinite(a["q"]).all() and np.isfinite(a["f"]).all() and np.isfinite(pe).all()
        ie = np.argsort(a["gid"])
        errs = (q_err(a["q"][ie], q0), f_err(a["f"][ie], f0), e_err(pe, pe0))
        print(case, "qeq_mode", qeq_mode, "step 0: q %.2e f %.2e E %.2e" % errs, "PE", ["%.6g" % (x - y) for x, y in zip(pe, pe0)], flush=True)
        assert errs[0] <= 1e-6 and errs[1] <= 1e-6 and errs[2] <= (5e-9 if extra else 1e-9), errs      # (PQEq: PE(12) and PE(13) are large sums of opposite sign, their 1e-7 CG noise cancels in PE(0) only)
        e.step(steps)
        a = e.atoms(); en = e.energy()
        ie = np.argsort(a["gid"])
        assert np.isfinite(a["q"]).all() and np.isfinite(a["f"]).all() and np.isfinite(en["PE"]).all() and np.isfinite(en["KE"])
        assert np.abs(a["pos"][ie] - pos1).max() <= 1e-9
        assert q_err(a["q"][ie], q1) <= 1e-6 and f_err(a["f"][ie], f1) <= 1e-6
        e.close()
print("POISON-OK")
