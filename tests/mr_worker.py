"""worker functions for the multi-process tests (spawned by torch.multiprocessing)"""
import os, sys, traceback
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE); sys.path.insert(0, os.path.dirname(HERE))


def _init(rank, world, port):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    return dist


def transport_selftest(rank, world, port, out):
    try:
        dist = _init(rank, world, port)
        from rxmd_amd.comm import TorchTransport
        t = TorchTransport(mode="host")
        rc = t.selftest()
        out[rank] = (rc, t.n_exchange, t.n_allreduce, repr(t.error))
        dist.barrier(); dist.destroy_process_group()
    except Exception:
        out[rank] = (-99, 0, 0, traceback.format_exc())


def engine_rank(rank, world, port, case, vp, steps, out):
    """one rank of a vprocs run; all ranks share GPU 0, messages are staged through the host over gloo"""
    try:
        dist = _init(rank, world, port)
        import torch
        import oracle_api as oa
        import rxmd_amd
        from rxmd_amd import system
        from rxmd_amd.comm import TorchTransport
        g = np.load(os.path.join(oa.GOLD, case + ".npz"))
        mc = tuple(int(x) for x in g["mc"])
        ff, names, frac, lat = oa.make_system(case)
        lat_s, rec = system.geninit(ff, names, frac, lat, mc=mc, vprocs=vp, myid=rank)
        e = rxmd_amd.RxmdEngine(ff, lat_s, vprocs=vp, myid=rank, QEq_tol=1e-12, NMAXQEq=2000, device=0)
        tr = TorchTransport(mode="staged", device=torch.device("cuda", 0), capacity_doubles=1 << 20)
        tr.attach(e)
        e.set_atoms_rxff(rec)
        it, est = e.QEq()
        pe = e.FORCE()
        if steps:
            e.step(steps)
        a = e.atoms()
        st = e.stats()
        out[rank] = dict(gid=a["gid"], q=a["q"], f=a["f"], pos=a["pos"], iters=st["qeq_iters_last"], pe=pe, nex=tr.n_exchange, nar=tr.n_allreduce, err=repr(tr.error))
        e.close()
        dist.barrier(); dist.destroy_process_group()
    except Exception:
        out[rank] = dict(error=traceback.format_exc())


def slab_system():
    """RDX-168 squeezed into the lower 44 % of a box 1/0.35 times as long in x: at vprocs 2x1x1 rank 1 owns NO atom"""
    import oracle_api as oa
    ff, names, frac, lat = oa.make_system("rdx168")
    frac2 = frac.copy(); frac2[:, 0] = frac2[:, 0] * 0.35
    lat2 = list(lat); lat2[0] = lat[0] / 0.35
    return ff, names, frac2, lat2


def engine_rank_empty(rank, world, port, steps, qeq_mode, out):
    """2x1x1 with an EMPTY second rank: it still takes part in every exchange round and all-reduce"""
    try:
        dist = _init(rank, world, port)
        import torch
        import rxmd_amd
        from rxmd_amd import system
        from rxmd_amd.comm import TorchTransport
        ff, names, frac, lat = slab_system()
        vp = (2, 1, 1)
        lat_s, rec = system.geninit(ff, names, frac, lat, vprocs=vp, myid=rank)
        e = rxmd_amd.RxmdEngine(ff, lat_s, vprocs=vp, myid=rank, QEq_tol=1e-12, NMAXQEq=2000, device=0, qeq_mode=qeq_mode, nbuffer=20000)
        tr = TorchTransport(mode="staged", device=torch.device("cuda", 0), capacity_doubles=1 << 20)
        tr.attach(e)
        e.set_atoms_rxff(rec)
        n0 = len(rec)
        e.QEq(); pe = e.FORCE()
        if steps:
            e.step(steps)
        a = e.atoms()
        out[rank] = dict(n0=n0, gid=a["gid"], q=a["q"], f=a["f"], pos=a["pos"], pe=pe, err=repr(tr.error))
        e.close()
        dist.barrier(); dist.destroy_process_group()
    except Exception:
        out[rank] = dict(error=traceback.format_exc())


def engine_rank_rccl(rank, world, port, case, vp, steps, qeq_mode, out):
    """one rank of a vprocs run on ITS OWN GPU (device = rank) with the engine's native RCCL transport; gloo only carries the 128-byte id"""
    try:
        dist = _init(rank, world, port)
        import torch
        import oracle_api as oa
        import rxmd_amd
        from rxmd_amd import system
        g = np.load(os.path.join(oa.GOLD, case + ".npz"))
        mc = tuple(int(x) for x in g["mc"])
        ff, names, frac, lat = oa.make_system(case)
        lat_s, rec = system.geninit(ff, names, frac, lat, mc=mc, vprocs=vp, myid=rank)
        e = rxmd_amd.RxmdEngine(ff, lat_s, vprocs=vp, myid=rank, QEq_tol=1e-12, NMAXQEq=2000, device=rank, qeq_mode=qeq_mode)
        idt = torch.zeros(128, dtype=torch.uint8)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(e.rccl_unique_id()), dtype=torch.uint8))
        dist.broadcast(idt, 0)
        e.init_rccl(bytes(idt.numpy().tobytes()), rank, world)
        e.set_atoms_rxff(rec)
        it, est = e.QEq()
        pe = e.FORCE()
        if steps:
            e.step(steps)
        a = e.atoms()
        st = e.stats()
        out[rank] = dict(gid=a["gid"], q=a["q"], f=a["f"], pos=a["pos"], iters=st["qeq_iters_last"], pe=pe, nghost=st["nghost_force"])
        e.close()
        dist.barrier(); dist.destroy_process_group()
    except Exception:
        out[rank] = dict(error=traceback.format_exc())


def engine_rank_perturbed(rank, world, port, vp, steps, qeq_mode, out):
    """one rank of the 36,288-atom perturbed RDX system of tests/test_gpu_scale.py; all ranks share GPU 0, messages staged over gloo"""
    try:
        dist = _init(rank, world, port)
        import torch
        import rxmd_amd
        from rxmd_amd.comm import TorchTransport
        import test_gpu_scale as ts
        ff, lat2, ranks, vs = ts._perturbed_rdx(vp)
        r = ranks[rank]
        n = len(r["type"])
        rec = np.zeros((n, 10))
        rec[:, 0:3] = r["rnorm"]; rec[:, 3:6] = vs[rank]; rec[:, 7] = r["type"] + r["gid"] * 1e-13
        e = rxmd_amd.RxmdEngine(ff, lat2, vprocs=vp, myid=rank, device=0, qeq_mode=qeq_mode, **ts.KW)
        tr = TorchTransport(mode="staged", device=torch.device("cuda", 0), capacity_doubles=1 << 22)
        tr.attach(e)
        e.set_atoms_rxff(rec)
        e.QEq(); e.FORCE()
        if steps:
            e.step(steps)
        a = e.atoms(); st = e.stats()
        out[rank] = dict(gid=a["gid"], q=a["q"], f=a["f"], pos=a["pos"], natoms=st["natoms"], n_boundary_rows=st["n_boundary_rows"], err=repr(tr.error))
        e.close()
        dist.barrier(); dist.destroy_process_group()
    except Exception:
        out[rank] = dict(error=traceback.format_exc())


def mode_velocities(gid, seed, sigma, ntotal):
    """seeded velocities that depend only on the GLOBAL atom id (the same atom gets the same draw on every decomposition)"""
    v = np.random.default_rng(seed).normal(0.0, sigma, (ntotal + 1, 3))
    return v[gid]


def engine_rank_mode(rank, world, port, spec, out):
    """one rank of a vprocs run in any MODE of the driver (round 6: PQEq, isQEq 2, the velocity-scaling modes, the electric field) -- all
    ranks share GPU 0, messages host-staged over gloo.  spec: dict(case, mc, vp, steps, pqeq, isQEq, efield, qeq_mode, thermo=(mode, kw, every),
    restart=<golden with restart_rxff>, v=(seed, sigma, natoms of the whole system), kw=engine keywords)"""
    try:
        dist = _init(rank, world, port)
        import torch
        import oracle_api as oa
        import rxmd_amd
        from rxmd_amd import system
        from rxmd_amd.comm import TorchTransport
        vp = tuple(spec["vp"])
        kw = dict(spec["kw"]) if "kw" in spec else dict(QEq_tol=1e-12, NMAXQEq=2000)
        if spec.get("restart"):
            g = np.load(os.path.join(oa.GOLD, spec["restart"] + ".npz"))
            ff = oa.make_system(spec["case"])[0]
            lat_s, vp_file, _, recs = oa.parse_rxff(g["restart_rxff"])
            assert tuple(vp_file) == vp
            rec = recs[rank]
        else:
            ff, names, frac, lat = oa.make_system(spec["case"])
            lat_s, rec = system.geninit(ff, names, frac, lat, mc=tuple(spec["mc"]), vprocs=vp, myid=rank)
            if spec.get("v"):
                ty = np.rint(rec[:, 7]).astype(np.int64); gid = np.rint((rec[:, 7] - ty) * 1e13).astype(np.int64)
                rec[:, 3:6] = mode_velocities(gid, *spec["v"])
        e = rxmd_amd.RxmdEngine(ff, lat_s, vprocs=vp, myid=rank, device=0, qeq_mode=spec.get("qeq_mode", 1), isQEq=spec.get("isQEq", 1),
                                pqeq=oa.PQEQ_SICNP if spec.get("pqeq") else None, efield=spec.get("efield"), **kw)
        tr = TorchTransport(mode="staged", device=torch.device("cuda", 0), capacity_doubles=1 << 21)
        tr.attach(e)
        e.set_atoms_rxff(rec)
        e.QEq(); pe = e.FORCE()
        a0 = e.atoms(); sh0 = e.shells() if spec.get("pqeq") else None          # the state after the pre-loop QEq + FORCE (main.F90:27-32)
        th = spec.get("thermo")
        for n in range(spec.get("steps", 0)):
            if th and n % th[2] == 0:
                e.thermostat(th[0], **th[1])
            e.step(1)
        a = e.atoms(); st = e.stats()
        out[rank] = dict(gid0=a0["gid"], q0=a0["q"], f0=a0["f"], pos0=a0["pos"], shells0=sh0,
                         gid=a["gid"], q=a["q"], f=a["f"], pos=a["pos"], v=a["v"], pe=pe, shells=e.shells() if spec.get("pqeq") else None,
                         iters=st["qeq_iters_last"], nghost=st["nghost_force"], nex=tr.n_exchange, nar=tr.n_allreduce, err=repr(tr.error))
        e.close()
        dist.barrier(); dist.destroy_process_group()
    except Exception:
        out[rank] = dict(error=traceback.format_exc())
