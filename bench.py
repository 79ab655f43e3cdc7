#!/usr/bin/env python3
"""bench.py -- MD steps/s of the HIP ReaxFF+QEq hot path on replicated RDX at fixed atoms/GPU.

Contract (driver):  python bench.py --gpus N --steps K --warmup W      (N>1: launched by torch.distributed.run)
One "step" = one pass of the MD loop body (reference src/main.F90:64-98: kick, drift, migrate, QEq, FORCE,
kick) over the workload, inputs resident in HBM.  Workload at N=1 = BASELINE.json configs[1]: RDX unit cell
(168 atoms) replicated 18x18x18 = 979,776 atoms, QEq tol 1e-7, dt 0.25 fs, v0 = 0, q0 = 0 (the reference's
own example input, examples/1-reaxff/rxmd.in).  Prints ONE JSON line on rank 0.
"""
import argparse, json, os, resource, shutil, subprocess, sys, tempfile, time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
INP = os.path.join(ROOT, "tests", "golden", "inputs")
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md
L2_PEAK_GBS = 34500.0  # aggregate L2 -> L1 rate of the 8 XCDs, same guide
ATOMIC_REQ_PER_S = 20e9  # scattered float atomics at the memory side: 0.08 TB/s of 4-byte adds, one 64-byte request each (same guide)
SIMDS, CLOCK_HZ = 1024, 2.4e9   # 256 CUs x 4 SIMDs, peak engine clock: a wave64 vector instruction occupies its SIMD for 4 cycles
ATOMS_PER_GPU_CELLS = 18


def _run_reference(exe, gen, mc, nsteps, nthreads):
    """one timed run of the reference's OpenMP build on RDX mc^3; returns (atoms, loop seconds, wall seconds) or None"""
    tmp = tempfile.mkdtemp(prefix="rxmd_cpu_")
    try:
        os.makedirs(os.path.join(tmp, "DAT"))
        shutil.copy(os.path.join(INP, "rdx.xyz"), os.path.join(tmp, "input.xyz"))
        shutil.copy(os.path.join(INP, "ffield_rdx"), os.path.join(tmp, "ffield"))
        shutil.copy(os.path.join(INP, "rxmd.in"), os.path.join(tmp, "rxmd.in"))
        subprocess.run([gen, "-i", "input.xyz", "-f", "ffield", "-o", "DAT", "-mc", str(mc), str(mc), str(mc)], cwd=tmp, check=True, stdout=subprocess.DEVNULL)
        env = dict(os.environ, OMP_NUM_THREADS=str(nthreads), OMP_STACKSIZE="1G")
        t0 = time.time()
        def unlimited_stack():                # in the child only: the benchmark process keeps its own limits
            try:
                resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
            except Exception:
                soft, hard = resource.getrlimit(resource.RLIMIT_STACK)
                resource.setrlimit(resource.RLIMIT_STACK, (hard, hard))
        p = subprocess.run([exe, "--ntime_step", str(nsteps), "--pstep", "1000", "--fstep", "100000"], cwd=tmp, env=env, preexec_fn=unlimited_stack,
                           stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        wall = time.time() - t0
        loop = None
        for l in p.stdout.split("\n"):
            if "total (sec)" in l:
                loop = float(l.split()[2])       # the reference's own timer around its MD loop (main.F90:35,108-109)
        if loop is None or loop <= 0 or "successfully finished" not in p.stdout:
            return None
        return 168 * mc ** 3, loop, wall
    except Exception:
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def physical_cores():
    """physical cores of this host (lscpu: sockets x cores per socket); falls back to the logical count"""
    try:
        out = subprocess.run(["lscpu"], stdout=subprocess.PIPE, text=True, timeout=10).stdout
        kv = {l.split(":")[0].strip(): l.split(":")[1].strip() for l in out.split("\n") if ":" in l}
        return max(1, int(kv["Socket(s)"]) * int(kv["Core(s) per socket"])), int(kv.get("Thread(s) per core", "1"))
    except Exception:
        return os.cpu_count() or 1, 1


def cpu_baseline(sample="full"):
    """Times the REAL reference (oracle/_ref: Fortran + OpenMP, built by oracle/Makefile from the sources where they lie) on the PHYSICAL
    host cores of this box (OMP_NUM_THREADS = sockets x cores per socket, lscpu), on bounded samples of the same workload, same rxmd.in:
      * rxmd_omp_huge: the reference with ONLY its compiled-in capacity NBUFFER raised to 600,000 (module.F90:80), RDX 12x12x12 = 290,304
        atoms x 5 MD steps -- the sample BASELINE.md 3 plans, and the headline comparison when it ran;
      * rxmd_omp_big (NBUFFER 150,000): RDX 6x6x6 = 36,288 atoms x 10 MD steps;
      * rxmd_omp: the unmodified reference on the largest cube its NBUFFER = 30000 holds, RDX 3x3x3 = 4,536 atoms x 60 steps
        (there the NBUFFER-sized overheads dominate).
    Runs BEFORE torch / HIP are initialised in this process (fork of a GPU-initialised, multi-threaded parent is unsafe).  The
    reference keeps NBUFFER-sized automatic arrays on the stack: the stack limit is raised in the CHILD only (preexec_fn)."""
    ref = os.path.join(ROOT, "oracle", "_ref")
    gen = os.path.join(ref, "geninit")
    if not os.path.exists(gen):
        return None
    cores, smt = physical_cores()
    full = 168 * ATOMS_PER_GPU_CELLS ** 3
    out, samples = None, []
    try:
        mem_gb = os.sysconf("SC_PAGE_SIZE") * os.sysconf("SC_PHYS_PAGES") / 2 ** 30
    except Exception:
        mem_gb = 0
    plan = []
    if sample == "small":                  # --cpu-baseline-sample small (tests of the hand-off): only the unmodified reference on its 4,536-atom cube
        mem_gb = 0
    if mem_gb >= 40:                       # NBUFFER 600,000: nbplist + hessian alone are 10.8 GB
        plan.append((os.path.join(ref, "rxmd_omp_huge"), 12, 5, "USCCACS/RXMD Fortran+OpenMP with NBUFFER raised to 600000 (oracle/_ref/rxmd_omp_huge)"))
    if sample != "small":
        plan.append((os.path.join(ref, "rxmd_omp_big"), 6, 10, "USCCACS/RXMD Fortran+OpenMP with NBUFFER raised to 150000 (oracle/_ref/rxmd_omp_big)"))
    plan.append((os.path.join(ref, "rxmd_omp"), 3, 60, "USCCACS/RXMD Fortran+OpenMP unmodified (oracle/_ref/rxmd_omp)"))
    for exe, mc, nsteps, what in plan:
        if not os.path.exists(exe):
            continue
        r = _run_reference(exe, gen, mc, nsteps, cores)
        if r is None:
            continue
        atoms, loop, wall = r
        rate = atoms * nsteps / loop
        rec = {"sample": "%s, RDX %dx%dx%d = %d atoms, %d MD steps in %.2f s loop time (%.0f atom-steps/s), scaled to steps/s at %d atoms"
                         % (what, mc, mc, mc, atoms, nsteps, loop, rate, full),
               "atoms": atoms, "steps": nsteps, "loop_s": loop, "wall_s": wall, "atom_steps_per_s": rate, "value": rate / full}
        samples.append(rec)
        if out is None:
            out = {"value": rate / full, "unit": "steps/s", "cores": cores, "threads_per_core_on_host": smt, "kind": "reference", "sample": rec["sample"],
                   "atom_steps_per_s": rate, "wall_s": wall}
    if out is not None and len(samples) > 1:
        out["other_samples"] = samples[1:]
    return out


def make_workload(workload, ncells):
    """(ffield, names, fractional coordinates, lattice, cells per edge per GPU, description, pqeq parameter file) of a BASELINE configuration"""
    from rxmd_amd import system
    pqeq = None
    if workload == "rdx":
        ff = os.path.join(INP, "ffield_rdx")
        names, frac, lat = system.read_xyz(os.path.join(INP, "rdx.xyz"))
        cells = (ncells,) * 3
        wname = "RDX %dx%dx%d cells per GPU" % cells
    elif workload == "water":
        # BASELINE configs[2] / SURVEY 8d C3: conf/init.water/ice-1h.xyz holds REAL coordinates with exactly collinear
        # O-H...O triples (NaN in the reference): Gaussian kick sigma 0.02 A, numpy default_rng(12345)
        import numpy as np
        ff = os.path.join(INP, "ffield_water")
        lines = open(os.path.join(INP, "ice-1h_real.xyz")).read().split("\n")
        n0 = int(lines[0].split()[0]); lat = [float(x) for x in lines[1].split()[:6]]
        rng = np.random.default_rng(12345)
        names, frac = [], []
        for l in lines[2:2 + n0]:
            e, x, y, z = l.split()[:4]
            r = np.array([float(x), float(y), float(z)]) + rng.normal(0.0, 0.02, 3)
            names.append(e); frac.append(r / np.array(lat[:3]))
        frac = np.array(frac)
        cells = (60, 35, 40) if ncells == ATOMS_PER_GPU_CELLS else (ncells,) * 3
        wname = "perturbed ice Ih %dx%dx%d cells per GPU" % cells
    else:
        ff = os.path.join(INP, "ffield_sicnp")
        names, frac, lat = system.read_xyz(os.path.join(INP, "sicnp.xyz"))
        pqeq = os.path.join(INP, "pqeq_sicnp.in")
        c = 12 if ncells == ATOMS_PER_GPU_CELLS else ncells
        cells = (c, c, c)
        wname = "SiC nanoparticle + O2 (conf/init.sicnp) %dx%dx%d cells per GPU, PQEq" % cells
    return ff, names, frac, lat, cells, wname, pqeq


def pass_bytes(st, pq):
    """algorithmic bytes of ONE matrix pass: value f64 + column i32 per entry (PQEq: + the shell-core value f64 of pqeq.F90:381-411,
    streamed by the pass that produces Est) + ~7 vector words per row.  This is the figure `roofline.achieved` is computed from whatever the
    pass streams: the survey says a scheme that beats the formula is still reported against it."""
    return st["nnz10"] * (20.0 if pq else 12.0) + st["natoms"] * 56.0


def pass_info(st, pq):
    """which matrix pass ran, and the bytes it really streams per launch: the window pass (k_spmv_win) reads a 16-bit window slot instead of the
    4-byte entry (10 / 18 bytes per entry) plus the window's vector entries once per group of 16 rows"""
    win = bool(st.get("win_in_use", 0))
    bpe = (18.0 if pq else 10.0) if win else (20.0 if pq else 12.0)
    return {"pass": "window (k_spmv_win: 16 cell-sorted rows per workgroup, partners staged in LDS, 16-bit slots)" if win else "row (k_spmv: one wavefront per row, 16-byte gather per entry)",
            "bytes_per_entry_streamed": bpe, "streamed_bytes_per_launch": st["nnz10"] * bpe + st["natoms"] * 56.0,
            "window_groups": st.get("win_groups", 0), "largest_window_slots": 8 * st.get("win_max_units", 0)}


def compact_leg(workload, ncells, steps, warmup, device, **kw):
    """one more configuration in the same process, single rank: set-up, QEq + FORCE, warm-up, `steps` timed steps -> a compact record"""
    import torch, rxmd_amd
    from rxmd_amd import system
    ff, names, frac, lat, cells, wname, pqeq = make_workload(workload, ncells)
    cfg = system.parse_rxmd_in(os.path.join(INP, "rxmd.in"))
    lat_super, rec = system.geninit(ff, names, frac, lat, mc=cells)
    ekw = dict(isQEq=cfg["isQEq"], NMAXQEq=cfg["NMAXQEq"], QEq_tol=cfg["QEq_tol"], qstep=cfg["qstep"], dt_fs=cfg["dt"], device=device, qeq_mode=1, pqeq=pqeq)
    ekw.update(kw)
    eng = rxmd_amd.RxmdEngine(ff, lat_super, **ekw)
    try:
        eng.set_atoms_rxff(rec)
        eng.QEq(); eng.FORCE(); eng.step(warmup); eng.reset_timers()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        eng.step(steps)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
        st = eng.stats()
        launches = max(st["spmv_launches"], 1)
        ms_spmv = st["ms_qeq_spmv"] / launches
        bp = pass_bytes(st, pqeq is not None)
        ach = bp / (ms_spmv * 1e-3) / 1e9 if ms_spmv > 0 else 0.0
        return {"workload": "%s = %d atoms, QEq tol %g, dt %g fs" % (wname, len(rec), ekw["QEq_tol"], ekw["dt_fs"]), "isQEq": ekw["isQEq"], "qeq_mode": ekw["qeq_mode"],
                "steps": steps, "warmup": warmup, "ms_per_step": 1e3 * dt / steps, "steps_per_s": steps / dt, "ns_per_day": steps / dt * ekw["dt_fs"] * 86400e-6,
                "qeq_iters_per_step": st["qeq_iters_total"] / max(st["qeq_calls"], 1), "spmv_launches_per_step": st["spmv_launches"] / steps,
                "roofline": dict({"bound": "hbm", "kernel": "k_spmv_win" if st.get("win_in_use") else "k_spmv", "bytes_per_entry": 20 if pqeq else 12, "algorithmic_bytes_per_launch": bp, "avg_launch_ms": ms_spmv, "achieved": ach,
                             "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}, **pass_info(st, pqeq is not None)),
                "breakdown_ms_per_step": {k: st[k] / steps for k in ("ms_qeq", "ms_qeq_spmv", "ms_lists", "ms_force", "ms_bo", "ms_nonbond", "ms_bonded", "ms_k_winbuild", "ms_bond_exposed")},
                "bond_overlap": st.get("bond_overlap", 0),
                "kernel_ms_per_step": {k: st[k] / steps for k in st if k.startswith("ms_k_")}}
    finally:
        eng.close()


def spawn_ranks(n):
    """start `n` ranks of this script under torch.distributed.run as a child process; returns its exit code"""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    last_json = None
    for line in p.stdout:                      # relay; the ranks' JSON line is printed once more as the LAST line of this process
        if line.startswith("{") and '"metric"' in line:
            last_json = line.rstrip("\n")
        else:
            sys.stdout.write(line)
    rc = p.wait()
    sys.stdout.flush()
    if last_json is not None:
        print(last_json, flush=True)
    elif rc == 0:
        sys.stderr.write("bench.py: the %d ranks printed no result line\n" % n)
        rc = 1
    return rc


def vprocs_for(n):
    """the rank grid of n ranks: 1, 2, 4, 8 -> the grids SURVEY 8d names (2 1 1 / 2 2 1 / 2 2 2); any other n -> its prime factors dealt to
    the currently smallest axis, x first (6 -> 3 2 1)"""
    vp = [1, 1, 1]
    f, m = 2, n
    fac = []
    while m > 1:
        while m % f == 0:
            fac.append(f); m //= f
        f += 1
    for p in sorted(fac, reverse=True):
        vp[vp.index(min(vp))] *= p
    return tuple(sorted(vp, reverse=True))


# ---- what the first run between real peers is to be held against ----------------------------------------------------------------
# The multi-rank code path has only ever run on ONE GPU (a rank pushed through the staged exchange with RCCL send/recv to itself:
# profiles/r06_selfloop_*.json, RXMD_FORCE_STAGED=1 RXMD_FORCE_REMOTE=1).  That measurement holds every cost of N ranks except the wire: the
# packing and unpacking kernels, the RCCL launches, the host's size messages, the split matrix pass.  The prediction adds the wire:
#   bytes that cross a face of a split axis / the rate of ONE xGMI link (each stage talks to one face neighbour over its own link; the two stages of
#   an axis with two ranks go to the same peer), for the exchanges nothing hides (ghost build 48 B, charge halo 8 B, force fold 24 B per ghost);
#   the (hs,ht) halo of every CG iteration (16 B per ghost) runs under the interior window groups of the matrix pass and counts only beyond them;
#   two small all-reduces per CG iteration at the latency of N peers instead of one.
PREDICT = {
    "source": "profiles/r06_final2_selfloop_staged_overlap.json (one MI355X through the multi-rank code path: 49.02 ms per step at K = 36.55, pass 0.8569 ms in the loop) + assumed wire figures",
    "selfloop_fixed_ms": 14.37,          # per step: exchange + lists + FORCE + kicks of the self-loop run (its ms per step minus its CG iterations)
    "selfloop_cg_ms_beyond_the_pass": 0.091,   # per CG iteration beyond the matrix pass itself: vector kernels, the second launch of the split pass, the exposed part of the halo to itself, two all-reduces to itself
    "single_rank_fixed_ms": 13.55, "single_rank_cg_ms_beyond_the_pass": 0.068,   # the same two figures of the single-rank fast path (profiles/r06_final2_selfloop_single_rank*.json): the N = 1 row of DESIGN.md 6
    "allreduce_self_us": 5.5,            # an 8-double ncclAllReduce with one rank: its launch
    "allreduce_us": {1: 5.5, 2: 10.0, 4: 15.0, 8: 20.0},   # ASSUMED small-message latency of RCCL over xGMI (no two-GPU lease inside a round to measure it)
    "xgmi_link_GBs_per_direction": 60.0,  # ASSUMED achieved rate for MB-sized messages (7 links x ~153 GB/s bidirectional per GPU = 76.8 GB/s per direction and link at peak)
    "interior_fraction_of_pass": 0.75,   # window groups without a ghost partner at 979,776 atoms per rank
}


def predict_ms_per_step(vp, K, nghost, pass_ms):
    """predicted wall time of one MD step on the rank grid vp at K CG iterations per step and the (slowest rank's) matrix-pass time of THIS run
    (weak scaling, the per-rank size of the self-loop run: 979,776 atoms, 400,262 ghosts)"""
    nsplit = sum(1 for v in vp if v > 1)
    world = vp[0] * vp[1] * vp[2]
    link = PREDICT["xgmi_link_GBs_per_direction"] * 1e9
    per_axis_ghosts = nghost / 3.0
    wire_fixed_ms = nsplit * per_axis_ghosts * 80.0 / link * 1e3
    halo_ms = nsplit * (per_axis_ghosts * 16.0 / link * 1e3 + 0.03)            # per CG iteration: the stages of the split axes one after the other, ~30 us of pack / launch / unpack each
    halo_exposed_ms = max(0.0, halo_ms - PREDICT["interior_fraction_of_pass"] * pass_ms)
    ar = PREDICT["allreduce_us"].get(world, 20.0 + 2.0 * max(world - 8, 0) ** 0.5)
    allreduce_ms = 2.0 * K * (ar - PREDICT["allreduce_self_us"]) * 1e-3
    self_loop = PREDICT["selfloop_fixed_ms"] + K * (pass_ms + PREDICT["selfloop_cg_ms_beyond_the_pass"])
    total = self_loop + wire_fixed_ms + K * halo_exposed_ms + allreduce_ms
    return {"ms_per_step": round(total, 2), "at_cg_iterations_per_step": round(K, 2), "at_pass_ms": round(pass_ms, 4), "self_loop_part_ms": round(self_loop, 2),
            "wire_ms": round(wire_fixed_ms, 3), "vector_halo_ms_per_iteration": round(halo_ms, 3), "vector_halo_exposed_ms_per_iteration": round(halo_exposed_ms, 3),
            "allreduce_extra_ms": round(allreduce_ms, 3), "assumptions": PREDICT["source"] + "; link %g GB/s per direction, all-reduce %g us" % (PREDICT["xgmi_link_GBs_per_direction"], ar)}


def leg_summary(name, rec):
    """ms per step, CG iterations per step and the matrix pass's roofline fraction of one leg, as short as it gets"""
    if not rec:
        return None
    if "error" in rec:
        return {"leg": name, "error": str(rec["error"])[:120]}
    r = rec.get("roofline") or {}
    out = {"leg": name, "ms_per_step": round(rec["ms_per_step"], 3), "K": round(rec.get("qeq_iters_per_step", 0.0), 2)}
    if r.get("avg_launch_ms"):
        out["pass_ms"] = round(r["avg_launch_ms"], 4); out["frac"] = round(r["frac"], 4)
    elif rec.get("avg_pass_ms"):
        out["pass_ms"] = round(rec["avg_pass_ms"], 4)
    if "ms_qeq_per_iter" in rec:
        out["ms_qeq_per_iter"] = round(rec["ms_qeq_per_iter"], 4)
    return out


def compact_line(full, limit=7600):
    """the line for stdout: every key of the driver's contract, `roofline` and `cpu_baseline` without their tables, the per-rank record of an N > 1
    run, and -- LAST, so that a recorded tail holds it -- `legs`: one short record per leg (steady, isQEq 2, the other algebra, water, SiC-NP + PQEq ...)."""
    keep = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config",
            "ns_per_day", "atom_steps_per_s", "qeq_iters_per_step", "ms_qeq_per_iter", "n10", "nb", "bond_overlap", "timer_pairs_dropped", "energy_per_atom", "bench_wall_s", "full_record")
    out = {k: full[k] for k in keep if k in full}
    r = dict(full["roofline"])
    rk = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "frac_real_traffic", "frac_streamed", "algorithmic_bytes_per_launch", "bytes_per_entry",
          "bytes_per_entry_streamed", "avg_launch_ms", "launches", "launches_timed", "launches_that_returned_at_once", "measured_read_stream_GBs", "frac_of_measured_read_stream",
          "spmv_launches_per_step", "step_frac_of_hbm_roofline")
    out["roofline"] = {k: r[k] for k in rk if k in r}
    if r.get("traffic_source"):
        out["roofline"]["traffic_source"] = {k: r["traffic_source"].get(k) for k in ("file", "measured_at_commit", "kernel_signature")}
    ps = r.get("placement_search") or {}
    out["roofline"]["placement_search"] = {k: ps.get(k) for k in ("pass_ms_first_placement", "pass_ms_kept_placement", "draws", "total_ms", "in_timed_region")}
    out["roofline"]["kernels_ms"] = {k["name"].split("+")[0].split(" ")[0]: [round(k["ms"], 4), k.get("bound"), None if k.get("frac_of_bound") is None else round(k["frac_of_bound"], 3)] for k in r.get("kernels", [])}
    out["breakdown_ms_per_step"] = {k: round(v, 4) for k, v in full.get("breakdown_ms_per_step", {}).items()}
    if "per_rank" in full:
        out["per_rank"] = full["per_rank"]
    if "predicted" in full:
        out["predicted"] = full["predicted"]
    if "cpu_baseline" in full:
        cb = full["cpu_baseline"]
        out["cpu_baseline"] = {k: cb[k] for k in ("value", "unit", "cores", "kind", "sample", "atom_steps_per_s", "measured_by") if k in cb}
    legs = [leg_summary("headline", dict(full, roofline=r))]
    for name, key in (("steady (100 steps after 10)", "steady"), ("isQEq 2", "alt_lex"), ("reference algebra (qeq_mode %d)" % (1 - full["config"].get("qeq_mode", 1)), "alt"),
                      ("no placement search", "alt_no_placement_search"), ("bonded chain on the other stream" if not full.get("bond_overlap") else "one stream", "alt_bond_streams")):
        if full.get(key):
            legs.append(leg_summary(name, full[key]))
    for rec in full.get("other_configs") or []:
        legs.append(leg_summary(str(rec.get("workload", "?")).split(" = ")[0].split(",")[0][:48], rec))
    out["legs"] = [l for l in legs if l]
    line = json.dumps(out)
    for drop in ("energy_per_atom", "breakdown_ms_per_step", "per_rank"):        # never over the limit: the least important records go first
        if len(line) <= limit:
            break
        if drop == "per_rank" and "per_rank" in out:
            out["per_rank"] = {k: v for k, v in out["per_rank"].items() if k in ("natoms", "nghost", "qeq_iters_total", "ms_halo_exposed_per_step", "pass_ms_in_the_loop")}
        else:
            out.pop(drop, None)
        legs_ = out.pop("legs"); out["legs"] = legs_                              # keep `legs` last
        line = json.dumps(out)
    return line


def main():
    t_bench0 = time.time()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--cells", type=int, default=ATOMS_PER_GPU_CELLS, help="RDX unit cells per edge per GPU (18 -> 979,776 atoms)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--qeq-mode", type=int, default=1, help="1: one matrix pass per CG iteration (gradient by recurrence; default), 0: the reference's two passes")
    ap.add_argument("--no-alt", action="store_true", help="skip the second leg (the other qeq_mode over the SAME trajectory window, reported under \"alt\")")
    ap.add_argument("--workload", default="rdx", choices=["rdx", "water", "sicnp"],
                    help="rdx: BASELINE configs[1]/[3] (headline); water: configs[2], perturbed ice Ih 60x35x40 = 2,016,000 atoms; "
                         "sicnp: configs[4], SiC nanoparticle + O2 with PQEq, 547-atom cell replicated --cells (default 12) per edge")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent periodic replicas instead of one decomposed box")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the compact legs of BASELINE configs[2] (water) and configs[4] (SiC nanoparticle, PQEq) and the isQEq 2 leg")
    ap.add_argument("--no-steady", action="store_true", help="skip the steady-state leg (SURVEY 8d: 100 timed steps after 10 warm-up steps from the same cold start)")
    ap.add_argument("--cpu-baseline-sample", default="full", choices=["full", "small"], help="small: only the unmodified reference on RDX 3x3x3 (seconds; the tests of the hand-off)")
    ap.add_argument("--full-line", default=os.environ.get("RXMD_BENCH_FULL_LINE"), metavar="FILE",
                    help="write the FULL record (per-kernel roofline table, every leg with its breakdown) to FILE; the line on stdout is its compact form, "
                         "kept under 8 KB so that a driver that records the tail of stdout sees all of it (default: the full record goes to stderr)")
    a = ap.parse_args()

    # `python bench.py --gpus N` without a launcher (the reference is started as `mpirun -np N rxmd`, examples/2-reaxff-dc/Makefile): this
    # process touches neither torch nor HIP; it starts the N ranks as a CHILD (torch.distributed.run, one rank per GPU), relays what they
    # print and leaves with their exit code.  Launched by a launcher (WORLD_SIZE set) --gpus must name the same N: never a silent 1-rank run.
    if "WORLD_SIZE" not in os.environ and a.gpus > 1:
        # the parent touches no GPU: it times the reference on the host cores BEFORE the ranks exist (nothing competes for the cores) and hands
        # the record to rank 0 through a file, so that an N > 1 line carries `cpu_baseline` like the N = 1 line does (the reference prints its
        # timer table on every rank count, main.F90:135-182)
        tmpf = None
        if not a.no_cpu_baseline and a.workload == "rdx":
            cbp = cpu_baseline(a.cpu_baseline_sample)
            if cbp is not None:
                fd, tmpf = tempfile.mkstemp(prefix="rxmd_cpu_baseline_", suffix=".json")
                with os.fdopen(fd, "w") as fh:
                    json.dump(cbp, fh)
                os.environ["RXMD_BENCH_CPU_BASELINE_FILE"] = tmpf
        try:
            rc = spawn_ranks(a.gpus)
        finally:
            if tmpf:
                os.unlink(tmpf)
        raise SystemExit(rc)
    if int(os.environ.get("WORLD_SIZE", "1")) != a.gpus and "RXMD_BENCH_FORCE_DIST" not in os.environ:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%s: the launcher and the flag must agree" % (a.gpus, os.environ.get("WORLD_SIZE", "1")))
    probes = [k for k in os.environ if k.startswith("RXMD_") and k.endswith("_PROBE")]
    if probes:
        raise SystemExit("bench.py: refusing to run with work-skipping switches set: %s" % ", ".join(sorted(probes)))

    # the CPU baseline first: rank 0 of a 1-GPU run, before torch / HIP exist in this process
    cb = None
    if not a.no_cpu_baseline and int(os.environ.get("WORLD_SIZE", "1")) == 1 and a.workload == "rdx" and "RXMD_BENCH_FORCE_DIST" not in os.environ:
        cb = cpu_baseline(a.cpu_baseline_sample)
    elif os.environ.get("RXMD_BENCH_CPU_BASELINE_FILE") and int(os.environ.get("RANK", "0")) == 0:
        try:                                  # measured by the parent of a self-spawned N > 1 run (above)
            cb = json.load(open(os.environ["RXMD_BENCH_CPU_BASELINE_FILE"]))
            cb["measured_by"] = "the parent process of bench.py --gpus N, before it started the ranks"
        except Exception:
            cb = None

    import torch
    import rxmd_amd
    from rxmd_amd import system

    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1")); local = int(os.environ.get("LOCAL_RANK", "0"))
    use_dist = world > 1 or "RXMD_BENCH_FORCE_DIST" in os.environ     # the latter: drive the N > 1 code path with one rank (tests)
    backend = os.environ.get("RXMD_BENCH_BACKEND", "nccl")       # "gloo": host-staged messages (several ranks on one GPU, debugging)
    if "RXMD_BENCH_DEVICE" in os.environ:
        local = int(os.environ["RXMD_BENCH_DEVICE"])
    if use_dist:
        import torch.distributed as dist
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    vp = vprocs_for(world)

    ff, names, frac, lat, cells, wname, pqeq = make_workload(a.workload, a.cells)
    mc = tuple(cells[i] * vp[i] for i in range(3))
    cfg = system.parse_rxmd_in(os.path.join(INP, "rxmd.in"))
    # weak scaling (BASELINE configs[3]): the box is cells*vprocs unit cells per edge, every rank owns one domain of the
    # reference's vprocs grid; ghost atoms, QEq vector halos, force return and migration travel through the engine's
    # six-stage exchange, scalars through small all-reduces.
    transport_mode = None
    if use_dist and not a.replicas:
        mc_local, vp_local, myid = mc, vp, rank
    else:
        mc_local, vp_local, myid = cells, (1, 1, 1), 0
    lat_super, rec = system.geninit(ff, names, frac, lat, mc=mc_local, vprocs=vp_local, myid=myid)
    natoms = len(rec)
    eng = rxmd_amd.RxmdEngine(ff, lat_super, vprocs=vp_local, myid=myid, isQEq=cfg["isQEq"], NMAXQEq=cfg["NMAXQEq"], QEq_tol=cfg["QEq_tol"],
                              qstep=cfg["qstep"], dt_fs=cfg["dt"], device=local, qeq_mode=a.qeq_mode, pqeq=pqeq)
    if use_dist and not a.replicas:
        from rxmd_amd.comm import TorchTransport
        dev = torch.device("cuda", local)
        # widest message pair of the six-stage exchange: the ghost build, 6 doubles per atom, both stages of an axis together.  Bound:
        # everything a rank can hold besides its residents (NBUFFER is sized 1.12 x (1 + 2 shell/L)^3 x natoms; the 13 A shell is
        # what the reference's NMINCELL x lcsize gives for RDX), i.e. 6 x 1.12 x ((1 + 26/L)^3 - 1) x natoms, plus slack
        lmin = min(lat_super[i] / vp_local[i] for i in range(3))
        cap = int(6 * 1.25 * natoms * ((1.0 + 27.0 / lmin) ** 3 - 1.0)) + (1 << 20)
        tr = None
        if backend == "nccl" and os.environ.get("RXMD_BENCH_TRANSPORT", "native") == "native":
            # native transport: the engine's own RCCL communicator (ncclSend/ncclRecv/ncclAllReduce on its stream, rccl_comm.hip);
            # torch.distributed only carries the 128-byte unique id
            ok = torch.ones(1, device=dev)
            try:
                idt = torch.zeros(128, dtype=torch.uint8, device=dev)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(eng.rccl_unique_id()), dtype=torch.uint8))
                dist.broadcast(idt, 0)
                eng.init_rccl(bytes(idt.cpu().numpy().tobytes()), rank, world)
                transport_mode = "native RCCL send/recv on the engine stream"
            except Exception as ex:
                sys.stderr.write("rank %d: native RCCL transport unavailable (%s)\n" % (rank, ex))
                ok[0] = 0
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            native_ok = ok.item() >= 1
        else:
            native_ok = False
        if native_ok:
            pass
        elif backend == "nccl":
            tr = TorchTransport(mode="device", device=dev, capacity_doubles=cap)
            ok = torch.ones(1, device=dev)
            try:                                           # one ring round on the device path before trusting it
                peer_to, peer_from = (rank + 1) % world, (rank - 1) % world
                tr.send_t[:8] = float(rank)
                n = tr._exchange(None, peer_to, 0, 8, peer_from, 0, cap)
                if n != 8 or float(tr.recv_t[0].item()) != float(peer_from):
                    ok[0] = 0
            except Exception:
                ok[0] = 0
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if ok.item() < 1:                              # fall back to host-staged messages over gloo
                gl = dist.new_group(backend="gloo")
                tr = TorchTransport(mode="staged", group=gl, device=dev, capacity_doubles=cap)
        else:
            tr = TorchTransport(mode="staged", device=dev, capacity_doubles=cap)
        if tr is not None:
            transport_mode = "torch.distributed " + tr.mode
            tr.attach(eng)
    eng.set_atoms_rxff(rec)
    eng.QEq(); eng.FORCE()                       # main.F90:27-32
    eng.step(a.warmup)
    eng.reset_timers()
    place_draws_before = eng.stats().get("place_draws", 0)      # the placement search runs inside the first QEq call that used the window pass: before this line
    import ctypes
    ctypes.CDLL(None).fflush(None)              # every rank: push out what RCCL / the runtime wrote to C stdout during set-up

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    barrier()
    t0 = time.perf_counter()
    eng.step(a.steps)
    barrier()
    dt = time.perf_counter() - t0
    if use_dist:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    st = eng.stats()
    en = eng.energy()
    per_rank = None
    if use_dist:          # every rank's residents, ghosts, CG iterations and exchange timers: a decomposition that silently lost a neighbour, or a rank
        # that waits for its halo, shows here
        tkeys = ("ms_ghost_build", "ms_migrate", "ms_halo", "ms_halo_exposed", "ms_allreduce", "ms_fold", "ms_qeq", "ms_force", "ms_lists")
        launches_r = max(st["spmv_launches"], 1)
        mine = torch.tensor([st["natoms"], st["nghost_force"], st["qeq_iters_total"], st["n_boundary_rows"]] + [st[k] / a.steps for k in tkeys]
                            + [st.get("place_ms_first", 0.0), st.get("place_ms_kept", 0.0), float(st.get("place_draws", 0)), st["ms_qeq_spmv"] / launches_r],
                            dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        allr = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank = {"natoms": [int(t[0]) for t in allr], "nghost": [int(t[1]) for t in allr], "qeq_iters_total": [int(t[2]) for t in allr],
                    "boundary_rows": [int(t[3]) for t in allr]}
        for j, k in enumerate(tkeys):
            per_rank[k + "_per_step"] = [round(float(t[4 + j]), 4) for t in allr]
        # every rank draws its own placement of the pass's streams (qeq.hip: tune_window_placement) and the slowest rank sets the step
        for j, k in enumerate(("place_ms_first", "place_ms_kept", "place_draws", "pass_ms_in_the_loop")):
            per_rank[k] = [round(float(t[4 + len(tkeys) + j]), 4) for t in allr]
    probe = None
    if world == 1:                               # plain 16-B/lane read of the matrix value array on this very box: the ceiling the pass is quoted next to
        try:
            pr = eng.debug(12, cap=16)
            rates = [pr[2 * g + 1] / (pr[2 * g] * 1e-3) / 1e9 for g in range(4) if pr[2 * g] > 0]
            probe = max(rates) if rates else None
        except Exception:
            probe = None
    alt = None
    if not a.no_alt and world == 1:
        # the other QEq algebra over the SAME window: restart from the same initial records, same warm-up, same number of steps (the
        # reference's exit test makes the iteration count jump from step to step, so only equal windows compare)
        eng.set_qeq_mode(1 - a.qeq_mode)
        eng.set_atoms_rxff(rec)
        eng.QEq(); eng.FORCE()
        eng.step(a.warmup)
        eng.reset_timers()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.step(a.steps)
        torch.cuda.synchronize(); dta = time.perf_counter() - t1
        sa = eng.stats()
        alt = {"qeq_mode": 1 - a.qeq_mode, "window": "same initial state, warm-up and step count as the headline leg", "steps": a.steps,
               "ms_per_step": 1e3 * dta / a.steps, "steps_per_s": a.steps / dta,
               "qeq_iters_per_step": sa["qeq_iters_total"] / max(sa["qeq_calls"], 1), "ms_qeq_per_iter": sa["ms_qeq"] / max(sa["qeq_iters_total"], 1),
               "spmv_launches_per_step": sa["spmv_launches"] / a.steps}
        eng.set_qeq_mode(a.qeq_mode)

    steady = None
    if not a.no_steady and world == 1:
        # SURVEY 8(d) protocol: 100 timed steps after 10 warm-up steps from the same cold start (q0 = 0, v0 = 0).  The headline window (--steps after
        # --warmup) sits in the first steps, where the charges start from zero and the CG needs ~36 iterations per step; here it has settled (K ~ 24).
        eng.set_atoms_rxff(rec)
        eng.QEq(); eng.FORCE()
        eng.step(10)
        eng.reset_timers()
        torch.cuda.synchronize(); t1 = time.perf_counter()
        eng.step(100)
        torch.cuda.synchronize(); dts = time.perf_counter() - t1
        ss = eng.stats()
        steady = {"protocol": "SURVEY 8(d): 100 timed steps after 10 warm-up steps from the cold start of the headline leg", "steps": 100, "warmup": 10,
                  "ms_per_step": 1e3 * dts / 100, "steps_per_s": 100 / dts, "ns_per_day": 100 / dts * cfg["dt"] * 86400e-6,
                  "qeq_iters_per_step": ss["qeq_iters_total"] / max(ss["qeq_calls"], 1), "spmv_launches_per_step": ss["spmv_launches"] / 100,
                  "avg_pass_ms": ss["ms_qeq_spmv"] / max(ss["spmv_launches"], 1),
                  "breakdown_ms_per_step": {k: ss[k] / 100 for k in ("ms_qeq", "ms_qeq_spmv", "ms_lists", "ms_force", "ms_bo", "ms_nonbond", "ms_bonded", "ms_k_winbuild")}}

    other, alt_lex, noplace, one_stream = None, None, None, None
    if world == 1 and not a.no_other_configs and a.workload == "rdx" and a.cells == ATOMS_PER_GPU_CELLS:
        eng.close(); eng = None                                   # the 979,776-atom engine gives its memory back first
        # isQEq = 2 (qeq.F90:51-57, main.F90:67-68,98): the reference's own production mode -- extended-Lagrangian charges, ONE CG step per MD
        # step; there the non-CG part of the step IS the step
        try:
            alt_lex = compact_leg("rdx", ATOMS_PER_GPU_CELLS, a.steps, a.warmup, local, isQEq=2)
        except Exception as ex:
            alt_lex = {"error": str(ex)}
        # the placement search of the window pass's streams (qeq.hip: tune_window_placement) switched off: what the headline gains from it
        try:
            os.environ["RXMD_PLACE_TRIES"] = "1"
            noplace = compact_leg("rdx", ATOMS_PER_GPU_CELLS, a.steps, a.warmup, local)
            noplace["note"] = "the headline configuration in a second engine of this process with RXMD_PLACE_TRIES=1 (first placement kept)"
        except Exception as ex:
            noplace = {"error": str(ex)}
        finally:
            os.environ.pop("RXMD_PLACE_TRIES", None)
        # The charge-free part of FORCE on a stream of its own next to ENbond (engine.h: bond_stream; RXMD_BOND_OVERLAP=1, the default of rounds 4-5): what
        # that would buy now.  The headline runs the one-stream order: its per-kernel times describe kernels that have the GPU to themselves.  (When
        # the headline itself is run with RXMD_BOND_OVERLAP=1 this leg is the one-stream order, and roofline.kernels[] takes its times from it.)
        headline_overlap = os.environ.get("RXMD_BOND_OVERLAP", "0") not in ("", "0")
        saved_overlap = os.environ.get("RXMD_BOND_OVERLAP")
        try:
            os.environ["RXMD_BOND_OVERLAP"] = "0" if headline_overlap else "1"
            one_stream = compact_leg("rdx", ATOMS_PER_GPU_CELLS, a.steps, a.warmup, local)
            one_stream["note"] = ("the headline configuration in another engine of this process with RXMD_BOND_OVERLAP=%s: bond orders, bonded terms and assembly %s"
                                  % (os.environ["RXMD_BOND_OVERLAP"], "queued on the main stream behind ENbond" if headline_overlap else "on a stream of their own next to ENbond"))
        except Exception as ex:
            one_stream = {"error": str(ex)}
        finally:
            if saved_overlap is None: os.environ.pop("RXMD_BOND_OVERLAP", None)
            else: os.environ["RXMD_BOND_OVERLAP"] = saved_overlap
        other = []
        for w in ("water", "sicnp"):                              # BASELINE configs[2] and configs[4] at their one-GPU sizes
            try:
                other.append(compact_leg(w, ATOMS_PER_GPU_CELLS, 8, 2, local))
            except Exception as ex:
                other.append({"workload": w, "error": str(ex)})

    if rank == 0:
        steps_per_s = a.steps / dt
        n10 = st["nnz10"] / max(st["natoms"], 1)
        nb = st["nbonds"] / max(st["natoms"] + st["nghost_force"], 1)
        iters = st["qeq_iters_total"] / max(st["qeq_calls"], 1)
        launches = max(st["spmv_launches"], 1)
        ms_spmv = st["ms_qeq_spmv"] / launches
        # algorithmic bytes of ONE matrix pass (value f64 + column i32 per entry, + ~7 vector words per row)
        bytes_pass = pass_bytes(st, pqeq is not None)
        achieved = bytes_pass / (ms_spmv * 1e-3) / 1e9 if ms_spmv > 0 else 0.0
        traffic = None
        # per-atom-step byte models with the measured n10, nb, K:
        #  (i) SURVEY 8d's formula as written: TWO matrix passes per CG iteration (the reference algebra, qeq_mode 0)
        # (ii) the bytes of the passes this run actually launched (qeq_mode 1: one pass per iteration + ~300 B of vector kernels)
        passes = st["spmv_launches"] / a.steps
        pinfo = pass_info(st, pqeq is not None)
        b_fixed = n10 * 14 + 40 + n10 * 4 + 64 + nb * 104 * 5      # the sweep writes the 16-bit window slot next to entry and value
        b_step_exec = b_fixed + passes * (n10 * pinfo["bytes_per_entry_streamed"] + 56) + iters * (300 if a.qeq_mode == 1 else 112)
        # the kernels behind the ~21 ms of a step that are not the matrix pass: HIP-event time per launch (rxmd_stats.ms_k_*), algorithmic bytes
        # per launch (SURVEY 8d / DESIGN.md 3), PMC bytes per launch where profiles/kernel_traffic.json holds them for this workload
        # PMC numbers are NOT measured in this run: they come from profiles/kernel_traffic.json (scripts/gpu_pmc_kernels.sh: one rocprofv3 --pmc pass per
        # counter set over a bench step), which records the commit and the kernels' full template signatures it was taken at.  A kernel whose
        # signature is not in the file gets null, never another instance's bytes.
        ktraffic, kvalu, kl2, katom, ksource = {}, {}, {}, {}, None
        kfile = os.path.join(ROOT, "profiles", "kernel_traffic.json")
        if os.path.exists(kfile):
            try:
                kj = json.load(open(kfile))
                if kj.get("natoms") == st["natoms"]:
                    ktraffic = kj.get("hbm_bytes_per_launch", {})
                    kvalu = {k: v.get("SQ_INSTS_VALU") for k, v in kj.get("kernels", {}).items() if v.get("SQ_INSTS_VALU") is not None}
                    kl2 = {k: v.get("l1_to_l2_read_bytes_at_128B") for k, v in kj.get("kernels", {}).items() if v.get("l1_to_l2_read_bytes_at_128B") is not None}
                    katom = {k: v.get("atomic_requests") for k, v in kj.get("kernels", {}).items() if v.get("atomic_requests")}
                    ksource = {"file": "profiles/kernel_traffic.json", "measured_at_commit": kj.get("commit"), "how": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), gfx950 correction FETCH_SIZE x 2; SQ_INSTS_VALU for the issue floor"}
            except Exception:
                pass
        def valu_floor_ms(*parts):
            """issue floor of the vector unit: SQ_INSTS_VALU (wave instructions per launch) x 4 cycles / 1,024 SIMDs / clock"""
            hit = [v for k, v in kvalu.items() if any(k == p_ or k.startswith(p_ + "<") for p_ in parts)]
            return (sum(hit) * 4.0 / SIMDS / CLOCK_HZ * 1e3) if hit else None
        def traffic_for(*parts):
            """PMC bytes per launch of the kernels whose names start with one of `parts` (template instances included), summed; None if absent"""
            hit = [v for k, v in ktraffic.items() if any(k == p_ or k.startswith(p_ + "<") for p_ in parts)]
            return sum(hit) if hit else None
        G = st["natoms"] + st["nghost_force"]
        ncg = max(st["qeq_iters_total"], 1)
        ms_cg_vec = max(st["ms_qeq"] - st["ms_qeq_spmv"] - st["ms_lists"], 0.0) / ncg       # per CG iteration: update + direction + sorted copy (+ reduction)
        kdefs = [("k_win_columns+k_list10", "ms_k_list10", st["nnz10"] * 14.0 + st["natoms"] * 40.0, a.steps, "10 A sweep: entry + value + 16-bit window slot written once (qeq.F90:183-268, main.F90:420-477)"),
                 ("k_nonbond_win+k_nonbond", "ms_k_nonbond", st["nnz10"] * 4.0 + st["natoms"] * 64.0, a.steps, "ENbond: the entry stream (pot.F90:676-781)"),
                 ("k_bo_prime+k_deltap+k_bo_full+k_delta_lp", "ms_k_bondorder", G * nb * 104.0, a.steps, "BOPRIM + BOFULL over residents and ghosts (bo.F90:28-298)"),
                 ("k_e3q+k_e3b", "ms_k_e3b", st["natoms"] * nb * 104.0, a.steps, "E3b (pot.F90:319-557): the surviving angles through a queue, 64 per wavefront batch (k_e3q; k_e3b = a thread per centre atom where a bond list is longer than 12): FP64 exp / log / acos chains, not bytes, bound it"),
                 ("k_e4b+k_e4b_deliver", "ms_k_e4b", st["natoms"] * nb * 104.0, a.steps, "E4b (pot.F90:980-1227): every torsion evaluated once on a persistent grid, its k-l side through the delivery table; latency of dependent loads and LDS, not bytes or issue, bounds it"),
                 ("k_ehb_donors+k_ehb_sweep+k_ehb", "ms_k_ehb", st["natoms"] * nb * 104.0, a.steps, "Ehb (pot.F90:559-673): bound by the 51 M memory-side FP64 atomics of the acceptor forces"),
                 ("k_bonded_list+k_bond_csr", "ms_k_blist", G * (32.0 + nb * 13.0), a.steps, "NEIGHBORLIST + nbrindx (main.F90:321-417): positions read once, partner / owner / mirror / type written per bond"),
                 ("k_cd_gather+k_ccbnd_terms+k_ccbnd_sum+k_bond_force_terms+k_bond_force_sum", "ms_k_assemble", G * nb * 104.0, a.steps, "ForceBondedTerms as gathers (pot.F90:113-144)")]
        def l2_bytes_for(*parts):
            """L1 <- L2 read bytes per launch: TCP_TCC_READ_REQ x 128 B (one request = one 128-byte line), summed over the named kernels; None if absent"""
            hit = [v for k, v in kl2.items() if any(k == p_ or k.startswith(p_ + "<") for p_ in parts)]
            return sum(hit) if hit else None
        def atomics_for(*parts):
            """memory-side atomic requests per launch (TCP_TCC_ATOMIC_WITHOUT_RET_REQ), summed over the named kernels; None if absent"""
            hit = [v for k, v in katom.items() if any(k == p_ or k.startswith(p_ + "<") for p_ in parts)]
            return sum(hit) if hit else None
        def bound_of(ms, hbm_bytes, l2_bytes, valu_ms, atomic_req=None):
            """the ceiling that binds a kernel: the largest of (HBM bytes / 8 TB/s, L1<-L2 bytes / 34.5 TB/s, vector issue floor, memory-side atomic
            requests / 20 G per s -- the rate of scattered float atomics in MI355X_MICROARCH.md: 0.08 TB/s of 4-byte adds, one request each) over its
            time; below 0.35 of every one of them the kernel waits on dependent round trips: `latency`, quoted with its best fraction"""
            if not ms or ms <= 0:
                return None, None, {}
            fr = {}
            if hbm_bytes: fr["hbm"] = hbm_bytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS
            if l2_bytes: fr["l2"] = l2_bytes / (ms * 1e-3) / 1e9 / L2_PEAK_GBS
            if valu_ms: fr["valu"] = valu_ms / ms
            if atomic_req: fr["atomics"] = atomic_req / ATOMIC_REQ_PER_S / (ms * 1e-3)
            if not fr:
                return None, None, fr
            b = max(fr, key=fr.get)
            return (b if fr[b] >= 0.35 else "latency"), fr[b], fr
        kernels = []
        # a kernel's time and ceilings are those of the kernel ALONE: when this engine ran the bonded list next to the 10 A sweep and the bonded chain next to
        # ENbond (rxmd_stats.bond_overlap), the per-kernel times come from the one-stream leg of the same configuration
        alone = (one_stream or {}).get("kernel_ms_per_step") if st.get("bond_overlap") else None
        for name, key, byts, cnt, note in kdefs:
            ms = alone[key] if (alone and key in alone) else st.get(key, 0.0) / max(cnt, 1)
            ach_k = byts / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            tr_k = traffic_for(*name.split("+"))
            vf_k = valu_floor_ms(*name.split("+"))
            bnd, fb, fr_all = bound_of(ms, tr_k, l2_bytes_for(*name.split("+")), vf_k, atomics_for(*name.split("+")))
            if st.get("bond_overlap") and not (alone and key in alone):      # the kernel shared the GPU with another chain and no one-stream leg ran: its time says nothing about a ceiling
                bnd, fb, fr_all = None, None, {}
            kernels.append({"name": name, "ms": ms, "algorithmic_bytes": byts, "achieved_GBs": ach_k, "frac": ach_k / HBM_PEAK_GBS, "traffic": tr_k,
                            "frac_real_traffic": (tr_k / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if (tr_k and ms > 0) else None, "valu_floor_ms": vf_k,
                            "bound": bnd, "frac_of_bound": fb, "frac_of_each_ceiling": fr_all, "note": note,
                            "timed": "alt_bond_streams leg (one stream)" if (alone and key in alone) else "headline leg"})
        kernels.append({"name": "CG vector kernels (k_cg_update, k_cg_direction, k_sorted_vec, k_reduce_fused)", "ms": ms_cg_vec, "algorithmic_bytes": st["natoms"] * 300.0,
                        "achieved_GBs": st["natoms"] * 300.0 / (ms_cg_vec * 1e-3) / 1e9 if ms_cg_vec > 0 else 0.0,
                        "frac": (st["natoms"] * 300.0 / (ms_cg_vec * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_cg_vec > 0 else 0.0, "traffic": traffic_for("k_cg_update", "k_cg_direction", "k_sorted_vec", "k_reduce_fused"),
                        "note": "per CG iteration, everything of qeq() that is neither the matrix pass nor the list build"})
        # the pass that ran, by its full template signature (MODE_HSH, STORE, PQ, NSTEP, VAR -- qeq.hip): only that instance's counters count
        if st.get("win_in_use"):                     # the engine reports the instance it dispatched (rxmd_stats.spmv_nstep / spmv_var, qeq.hip)
            sig = "k_spmv_win<0, true, %s, %d, %d>" % ("true" if pqeq else "false", st.get("spmv_nstep", 0), st.get("spmv_var", 0))
        else:
            sig = "k_spmv<0, true, %s, 1>" % ("true" if pqeq else "false")
        traffic = ktraffic.get(sig)                  # None when the file was taken with another instance of the kernel
        if ktraffic and traffic is None:
            sys.stderr.write("bench.py: profiles/kernel_traffic.json holds this atom count but not the dispatched instance %s: roofline.traffic stays null\n" % sig)
        out = {
            "metric": "MD steps/sec (RDX, 979,776 atoms/GPU; one step advances every GPU's domain: weak scaling, wall-clock steps/s of the whole job)" if a.workload == "rdx" and a.cells == ATOMS_PER_GPU_CELLS
                      else "MD steps/sec (%s, %d atoms/GPU; wall-clock steps/s of the whole job)" % (a.workload, natoms),
            "value": steps_per_s, "unit": "steps/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64",
            "data": "synthetic (unit cell of the reference's conf/init.%s replicated; v0=0, q0=0)" % ("rdx" if a.workload == "rdx" else a.workload),
            "config": {"workload": "%s = %d atoms/GPU, QEq tol %g, dt %g fs, mdmode 1 (NVE)" % (wname, natoms, cfg["QEq_tol"], cfg["dt"]),
                       "atoms_total": natoms * world, "parallelism": "1 GPU" if world == 1 else ("%d independent replicas" % world if a.replicas else
                                       "vprocs %dx%dx%d domain decomposition, six-stage halo, %s" % (vp[0], vp[1], vp[2], transport_mode)),
                       "qeq_mode": a.qeq_mode, "env": {k: v for k, v in sorted(os.environ.items()) if k.startswith("RXMD_")}},
            "steps_per_s_wall": steps_per_s, "ns_per_day": steps_per_s * cfg["dt"] * 86400e-6, "atom_steps_per_s": steps_per_s * natoms * world,
            "qeq_iters_per_step": iters, "ms_qeq_per_iter": st["ms_qeq"] / max(st["qeq_iters_total"], 1), "n10": n10, "nb": nb,
            "roofline": {"bound": "hbm", "kernel": ("k_spmv_win" if st.get("win_in_use") else "k_spmv") + " (QEq matrix pass, qeq.hip)", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "frac_is": "algorithmic bytes of SURVEY 8(d) (12 B per entry; 20 with PQEq) / time / peak", "traffic": traffic,
                         "traffic_source": dict(ksource, kernel_signature=sig) if (ksource and traffic is not None) else None,
                         "frac_real_traffic": (traffic / (ms_spmv * 1e-3) / 1e9 / HBM_PEAK_GBS) if (traffic and ms_spmv > 0) else None,
                         "frac_streamed": (pinfo["streamed_bytes_per_launch"] / (ms_spmv * 1e-3) / 1e9 / HBM_PEAK_GBS) if ms_spmv > 0 else None,
                         "valu_floor_ms": (kvalu[sig] * 4.0 / SIMDS / CLOCK_HZ * 1e3) if sig in kvalu else None,
                         "algorithmic_bytes_per_launch": bytes_pass,
                         "pass": pinfo["pass"], "bytes_per_entry_streamed": pinfo["bytes_per_entry_streamed"], "streamed_bytes_per_launch": pinfo["streamed_bytes_per_launch"],
                         "streamed_GBs": (pinfo["streamed_bytes_per_launch"] / (ms_spmv * 1e-3) / 1e9) if ms_spmv > 0 else 0.0,
                         "window_groups": pinfo["window_groups"], "largest_window_slots": pinfo["largest_window_slots"],
                         "placement_search": {"pass_ms_first_placement": st.get("place_ms_first", 0.0), "pass_ms_kept_placement": st.get("place_ms_kept", 0.0),
                                              "draws": st.get("place_draws", 0), "total_ms": st.get("place_total_ms", 0.0), "bytes_held": st.get("place_bytes_held", 0.0),
                                              "in_timed_region": bool(st.get("place_draws", 0) > place_draws_before),      # measured: draws counted after the timers were reset
                                              "note": "where the pass's streams lie in physical memory is worth up to 15 % of its time; the engine times a few placements once and keeps the fastest (RXMD_PLACE_TRIES)"},
                         "avg_launch_ms": ms_spmv, "launches": st["spmv_launches"], "launches_timed": st.get("spmv_launches_timed", st["spmv_launches"]),
                         "timing": "HIP event pair on the engine's stream around every %s-th launch of the timed region (RXMD_PASS_TIMING_EVERY); avg_launch_ms = their mean" % os.environ.get("RXMD_PASS_TIMING_EVERY", "8"),
                         "launches_that_returned_at_once": st.get("spmv_noop_launches", 0),
                         "measured_read_stream_GBs": probe, "frac_of_measured_read_stream": (achieved / probe) if probe else None,
                         "spmv_launches_per_step": passes,
                         "bytes_per_entry": 20 if pqeq else 12,
                         "step_bytes_per_atom_executed": b_step_exec, "step_frac_of_hbm_roofline": (b_step_exec * natoms * steps_per_s) / (HBM_PEAK_GBS * 1e9),
                         "kernels": kernels},
            "breakdown_ms_per_step": {k: st.get(k, 0.0) / a.steps for k in ("ms_qeq", "ms_qeq_spmv", "ms_lists", "ms_force", "ms_bo", "ms_nonbond", "ms_bonded", "ms_k_blist",
                                                                    "ms_ghost_build", "ms_migrate", "ms_halo", "ms_halo_exposed", "ms_allreduce", "ms_fold", "ms_k_winbuild", "ms_bond_exposed")},
            "bond_overlap": st.get("bond_overlap", 0),
            "timer_pairs_dropped": st.get("timer_pairs_dropped", 0),      # > 0: the ms_* breakdown of the timed region under-counts (rxmd_stats)
            "energy_per_atom": {"PE": en["PE"][0] / natoms, "KE": en["KE"] / natoms, "qsum": en["qsum"]},
        }
        if per_rank:
            out["per_rank"] = per_rank
            out["config"]["ranks_in_communicator"] = world      # rccl_init checks ncclCommCount against the vprocs grid
            if not a.replicas and a.workload == "rdx" and a.cells == ATOMS_PER_GPU_CELLS:
                # what this run is to be held against: the self-loop measurement of the same code path + the wire
                out["predicted"] = predict_ms_per_step(vp, iters, max(per_rank["nghost"]), max(per_rank["pass_ms_in_the_loop"]))
                out["predicted"]["measured_over_predicted"] = round(1e3 * dt / a.steps / out["predicted"]["ms_per_step"], 3)
        if alt:
            out["alt"] = alt
        if alt_lex:
            out["alt_lex"] = alt_lex
        if steady:
            out["steady"] = steady
        if noplace:
            out["alt_no_placement_search"] = noplace
        if one_stream:
            out["alt_bond_streams"] = one_stream
        if other:
            out["other_configs"] = other
        if cb:
            out["cpu_baseline"] = cb
        out["bench_wall_s"] = round(time.time() - t_bench0, 1)      # the whole run: every leg, the CPU baseline, set-up (not part of `value`)
        import ctypes
        ctypes.CDLL(None).fflush(None)          # C-level stdout first (RCCL prints a version banner there): the JSON line stays last
        sys.stdout.flush()
        # the full record (per-kernel roofline table, every leg with its breakdown: ~25 KB) goes to a file or to stderr; stdout gets its compact form
        # with a `legs` summary as the LAST key, under 8 KB (a driver that keeps the tail of stdout had lost alt / alt_lex / energy_per_atom of the long line)
        if a.full_line:
            try:
                os.makedirs(os.path.dirname(os.path.abspath(a.full_line)), exist_ok=True)
                with open(a.full_line, "w") as fh:
                    fh.write(json.dumps(out) + "\n")
                out["full_record"] = a.full_line
            except Exception as ex:
                sys.stderr.write("bench.py: could not write %s (%s)\n" % (a.full_line, ex))
        if "full_record" not in out:
            sys.stderr.write("bench.py full record: " + json.dumps(out) + "\n"); sys.stderr.flush()
            out["full_record"] = "stderr of this run (line starting with `bench.py full record:`); --full-line FILE writes it to a file"
        print(compact_line(out), flush=True)
    if eng is not None:
        eng.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
