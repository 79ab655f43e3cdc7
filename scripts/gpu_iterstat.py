"""Iteration statistics of the CG at QEq_tol 1e-7 over a long trajectory (RDX 2x2x2): iterations per step, one-iteration exits and their runs, distance of the
charges from the converged solution of the same geometry -- per matrix-pass / algebra variant (tests/test_gpu_parity.py::test_iteration_statistics_*).
usage: python scripts/gpu_iterstat.py [nsteps] [out.npz]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_api as oa
import rxmd_amd
from rxmd_amd import system
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 110


def runs(its):
    """lengths of the runs of consecutive one-iteration exits"""
    out, r = [], 0
    for x in its:
        if x <= 1: r += 1
        else:
            if r: out.append(r)
            r = 0
    if r: out.append(r)
    return out


def summary(tag, its, dqs):
    its = np.asarray(its); dqs = np.asarray(dqs); rl = runs(its[11:])
    print("%-22s mean its %.2f  one-iteration exits %.3f  runs: max %d hist %s  |dq| p50 %.2e p90 %.2e p99 %.2e max %.2e"
          % (tag, its[11:].mean(), (its[11:] <= 1).mean(), max(rl) if rl else 0, np.bincount(rl)[1:].tolist() if rl else [], np.percentile(dqs, 50), np.percentile(dqs, 90), np.percentile(dqs, 99), dqs.max()), flush=True)


if __name__ == "__main__":
    g = np.load(os.path.join(oa.GOLD, "rdx222_md1000.npz")); gd = np.load(os.path.join(oa.GOLD, "rdx222_md1000_dq.npz"))
    summary("reference", g["qeq_iters"], gd["dq_tight"])
    ff, names, frac, lat = oa.make_system("rdx222")
    res = {}
    for qeq_mode, win, seed in ((1, "1", 0), (1, "1", 1), (1, "1", 2), (0, "1", 0), (1, "0", 0), (0, "0", 0), (1, "0", 1), (0, "1", 1)):
        os.environ["RXMD_SPMV_WIN"] = win
        lat3, rec = system.geninit(ff, names, frac, lat, mc=(2, 2, 2))
        if seed:                                     # another draw of the same statistics: every atom moved by ~1e-6 A
            rec[:, 0:3] += np.random.default_rng(seed).normal(0, 1e-6, (len(rec), 3)) / np.asarray(lat3[:3])
        e = rxmd_amd.RxmdEngine(ff, lat3, qeq_mode=qeq_mode); e.set_atoms_rxff(rec)
        t = rxmd_amd.RxmdEngine(ff, lat3, QEq_tol=1e-12, NMAXQEq=2000, qeq_mode=0)
        it0, _ = e.QEq(); e.FORCE()
        its, dqs = [it0], []
        for s in range(NS):
            e.step(1)
            its.append(e.stats()["qeq_iters_last"])
            r = e.get_atoms_rxff()
            t.set_atoms_rxff(r); t.QEq()
            dqs.append(np.abs(t.atoms()["q"] - r[:, 6]).max())
        summary("mode %d win %s seed %d" % (qeq_mode, win, seed), its, dqs)
        res["its_%d_%s_%d" % (qeq_mode, win, seed)] = np.array(its); res["dq_%d_%s_%d" % (qeq_mode, win, seed)] = np.array(dqs)
        e.close(); t.close()
    if len(sys.argv) > 2:
        np.savez_compressed(sys.argv[2], **res)
