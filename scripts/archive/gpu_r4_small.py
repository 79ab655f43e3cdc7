"""where does a 168-atom MD step spend its time? (run under rocprofv3 --kernel-trace --stats)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_parity import _engine
e = _engine("rdx168", (1, 1, 1), QEq_tol=1e-12, NMAXQEq=2000)
t0 = time.time(); e.QEq(); e.FORCE(); t1 = time.time(); e.step(3); t2 = time.time()
st = e.stats()
print("QEq+FORCE %.2f s, 3 steps %.2f s; iterations %d; ms_qeq %.1f ms_lists %.1f ms_force %.1f" % (t1 - t0, t2 - t1, st["qeq_iters_total"], st["ms_qeq"], st["ms_lists"], st["ms_force"]), flush=True)
e.close()
