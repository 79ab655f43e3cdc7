"""A/B of library builds: the real window pass / row pass back to back (tap 104) and inside the CG loop, one engine per library in ONE process
usage: gpu_iso_libs.py tag1 tag2 ...   (tag "base" = rxmd_amd/librxmd_hip.so); each library in a child process, alternating twice"""
import os, sys, subprocess
here = os.path.dirname(os.path.abspath(__file__)); root = os.path.join(here, "..")
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, os.path.join(root, "tests")); sys.path.insert(0, root)
    from test_gpu_parity import _engine
    e = _engine("rdx168", (18, 18, 18), qeq_mode=1)
    e.QEq(); e.FORCE()
    os.environ["RXMD_ISO_REPS"] = "100"
    iso = e.debug(104, cap=4)
    s0 = e.stats(); e.step(4); s1 = e.stats(); nl = s1["spmv_launches"] - s0["spmv_launches"]
    print("%-6s back to back: window %.4f row %.4f | in the CG loop %.4f ms (%d passes) | winbuild %.3f" % (sys.argv[2], iso[0], iso[1], (s1["ms_qeq_spmv"] - s0["ms_qeq_spmv"]) / nl, nl, (s1["ms_k_winbuild"] - s0["ms_k_winbuild"]) / 4), flush=True)
    e.close(); sys.exit(0)
for rep in range(int(os.environ.get("AB_REPS", "2"))):
    for t in sys.argv[1:]:
        env = dict(os.environ)
        if t != "base": env["RXMD_HIP_LIB"] = os.path.join(root, "rxmd_amd", "librxmd_hip_%s.so" % t)
        subprocess.run([sys.executable, os.path.abspath(__file__), "--child", t], env=env)
