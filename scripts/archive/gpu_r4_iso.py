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
    print("window(2,0) %.4f  row %.4f  win(3,0) %.4f  win(2,pre) %.4f  window again %.4f || lean interior groups + pre %.4f  win(2,pre) %.4f  lean without pre %.4f" % (iso[0], iso[1], iso[2], iso[3], iso[4], iso[5], iso[6], iso[7]), flush=True)
e.close()
