"""Diagnostic run on a GPU box: compares every intermediate of the HIP path with the oracle and
prints a table.  Not a pytest file (the gated parity tests are tests/test_gpu_parity.py)."""
import json, os, sys, time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))
import oracle_api as oa
import rxmd_amd
from rxmd_amd import system


def relerr(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))


def run(case, mc, steps=0, **kw):
    out = {"case": case, "mc": mc}
    ff, names, frac, lat = oa.make_system(case)
    lat2, ranks = oa.geninit(names, frac, lat, oa.ffield_names(ff), mc=mc)
    o = oa.Oracle(ff, lat2, ranks, **kw)
    t0 = time.time(); oit = o.qeq(); o.force(); out["oracle_s"] = time.time() - t0
    lat3, rec = system.geninit(ff, names, frac, lat, mc=mc)
    e = rxmd_amd.RxmdEngine(ff, lat3, **{k: v for k, v in kw.items() if k in ("isQEq", "NMAXQEq", "QEq_tol")})
    e.set_atoms_rxff(rec)
    t0 = time.time(); it, est = e.QEq(); out["gpu_qeq_s"] = time.time() - t0
    st = e.stats()
    out["N"] = st["natoms"]; out["G_gpu"] = st["natoms"] + st["nghost_force"]; out["G_oracle"] = int(o.L.rxo_nghost_total(o.w, 0))
    out["iters_gpu"] = it; out["iters_oracle"] = oit; out["est_gpu"] = est; out["est_oracle"] = float(o.trace()[-1, 0]) if oit or len(o.trace()) else None
    a = e.atoms()
    out["q_err"] = relerr(a["q"], o.charges())
    out["n10_equal"] = bool((e.debug(6).astype(int) == o.get(104).astype(int)).all())
    out["hess_rowsum_err"] = relerr(e.debug(7), o.get(108))
    t0 = time.time(); pe = e.FORCE(); out["gpu_force_s"] = time.time() - t0
    G = out["G_gpu"]
    if out["G_gpu"] == out["G_oracle"]:
        out["ghost_pos_err"] = float(np.abs(e.debug(3, 3) - o.get(100, 0, 3)).max())
        out["ghost_gid_equal"] = bool((e.debug(4).astype(np.int64) == o.get(106).astype(np.int64)).all())
        out["nbrcnt_equal"] = bool((e.debug(2).astype(int) == o.get(103).astype(int)).all())
        out["deltap_err"] = float(np.abs(e.debug(1) - o.get(102)).max())
        out["delta_err"] = float(np.abs(e.debug(0) - o.get(101)).max())
    a = e.atoms()
    fo = o.forces(); frms = np.sqrt((fo ** 2).mean())
    out["f_maxabs_err"] = float(np.abs(a["f"] - fo).max()); out["f_rms"] = float(frms)
    out["f_rel_err"] = float((np.abs(a["f"] - fo).max(axis=1) / np.maximum(np.abs(fo).max(axis=1), frms)).max())
    peo = o.energy()
    out["pe_gpu"] = [float(x) for x in pe]; out["pe_oracle"] = [float(x) for x in peo]
    out["pe_relerr"] = [float(abs(x - y) / max(abs(y), 1e-12)) for x, y in zip(pe, peo)]
    if steps:
        o.step(steps); e.step(steps)
        a = e.atoms()
        og = o.gids(); order_o = np.argsort(og); order_g = np.argsort(a["gid"])
        out["md_order_equal"] = bool((og == a["gid"]).all()) if len(og) == len(a["gid"]) else False
        out["md_pos_err"] = float(np.abs(a["pos"][order_g] - o.pos()[order_o]).max())
        out["md_q_err"] = relerr(a["q"][order_g], o.charges()[order_o])
        fo = o.forces()[order_o]
        out["md_f_maxabs_err"] = float(np.abs(a["f"][order_g] - fo).max())
        out["md_iters_gpu"] = e.stats()["qeq_iters_last"]; out["md_iters_oracle"] = int(o.L.rxo_qeq_iters(o.w))
    out["stats"] = e.stats()
    e.close()
    return out


if __name__ == "__main__":
    res = []
    cases = [("rdx168", (1, 1, 1), 3, dict(QEq_tol=1e-12, NMAXQEq=2000)),
             ("rdx168", (1, 1, 1), 0, dict(isQEq=0)),
             ("rdx168", (1, 1, 1), 5, dict()),
             ("rdx222", (2, 2, 2), 2, dict(QEq_tol=1e-12, NMAXQEq=2000)),
             ("ice644", (6, 4, 4), 2, dict(QEq_tol=1e-12, NMAXQEq=2000))]
    for c in cases:
        try:
            r = run(c[0], c[1], c[2], **c[3])
        except Exception as ex:
            import traceback
            r = {"case": c[0], "error": repr(ex), "tb": traceback.format_exc()}
        res.append(r)
        print(json.dumps(r, indent=1, default=str)); sys.stdout.flush()
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(os.path.dirname(HERE), "gpurun_out", "gpu_debug.json"), "w"), indent=1, default=str)
