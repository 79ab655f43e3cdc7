"""does the pass time depend on where its streams lie?  several processes, in each: the pass on the engine's arrays and on two fresh copies"""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.join(here, "..")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
    from test_gpu_parity import _engine
    e = _engine("rdx168", (18, 18, 18), qeq_mode=1)
    e.QEq(); e.FORCE()
    os.environ["RXMD_ISO_REPS"] = "50"; os.environ["RXMD_ISO_COPIES"] = "1"
    for rep in range(2):
        iso = e.debug(104, cap=12); print("process %s: window pass %.4f (hess %x sl10 %x) | fresh copy %.4f (%x %x) | another fresh copy %.4f (%x %x) | row pass %.4f ms" % (sys.argv[2], iso[0], int(iso[4]), int(iso[5]), iso[2], int(iso[6]), int(iso[7]), iso[3], int(iso[8]), int(iso[9]), iso[1]), flush=True)
    e.close(); sys.exit(0)
for k in range(5):
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(k)])
