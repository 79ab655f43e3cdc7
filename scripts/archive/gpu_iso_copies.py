"""does the pass time depend on where its streams lie?  several processes; in each: the pass on the engine's arrays, on four copies held at the
same time (twice round: a property of the buffer repeats, a drift in time does not), on a copy with its rows in cell-sorted order"""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.join(here, "..")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
    from test_gpu_parity import _engine
    e = _engine("rdx168", (18, 18, 18), qeq_mode=1)
    e.QEq(); e.FORCE()
    os.environ["RXMD_ISO_REPS"] = "50"; os.environ["RXMD_ISO_COPIES"] = "1"
    for rep in range(2):
        iso = e.debug(104, cap=24)
        print("process %s: engine's arrays %.4f | copies A B (plain) C D (contiguous) %.4f %.4f %.4f %.4f | again %.4f %.4f %.4f %.4f | D in cell-sorted row order %.4f | engine's arrays again %.4f | row pass %.4f ms\n           value array at offsets 0 / 4 KB / 64 KB / 1 MB / 2 MB + 4 KB / 16 MB / 37 MB + 8 KB / 64 MB inside ONE allocation: %.4f %.4f %.4f %.4f %.4f %.4f %.4f %.4f"
              % ((sys.argv[2], iso[0]) + tuple(iso[2:10]) + (iso[10], iso[11], iso[1]) + tuple(iso[12:20])), flush=True)
    e.close(); sys.exit(0)
for k in range(int(os.environ.get("ISO_PROCS", "3"))):
    subprocess.run([sys.executable, os.path.abspath(__file__), "--child", str(k)], stderr=subprocess.STDOUT)
