"""window pass variants back to back in one process (experiments library, debug tap 104): default / row pass / 384 in flight / second batch
requested before the barrier / default again"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests")); sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from test_gpu_parity import _engine
e = _engine("rdx168", (18, 18, 18), qeq_mode=1)
e.QEq(); e.FORCE(); e.step(2)
os.environ["RXMD_ISO_REPS"] = "100"
for rep in range(4):
    iso = e.debug(104, cap=20)
    print("window %.4f  row %.4f  win384 %.4f  win_prefetch %.4f  window again %.4f | win2 (sorted rows, early issue) %.4f  window %.4f ms | max |row-sum difference| %.3e of %.3e" % tuple(iso[:9]), flush=True)
e.close()
