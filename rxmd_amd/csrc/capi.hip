// capi.hip -- the extern "C" boundary declared in include/rxmd_hip.h, plus the host front-end helpers
// (geninit / rxff.bin reader) that a driver needs around the hot path.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <new>
#include <string>
#include <vector>

#include "engine.h"

namespace rxmd { double stream_probe_ms(Engine &e, int blocks); void spmv_bisect_ms(Engine &e, double *out4); void spmv_isolated_ms(Engine &e, double *out); void spmv_tile_probe_ms(Engine &e, double *out); }
using rxmd::Engine;
using rxmd::EngineError;

struct rxmd_hip_engine {
  Engine *e = nullptr;
  std::string err;
};

static thread_local std::string g_create_error;

template <class F>
static int guarded(rxmd_handle h, F &&f) {
  if (!h || !h->e) return RXMD_E_ARG;
  try {
    f(*h->e);
    return RXMD_OK;
  } catch (const EngineError &e) {
    h->err = e.msg;
    return e.code;
  } catch (const std::bad_alloc &) {
    h->err = "host out of memory";
    return RXMD_E_NBUFFER;
  } catch (const std::exception &e) {
    h->err = e.what();
    return RXMD_E_STATE;
  }
}

extern "C" {

void rxmd_hip_default_config(rxmd_config *c) {
  std::memset(c, 0, sizeof(*c));
  c->vprocs[0] = c->vprocs[1] = c->vprocs[2] = 1;
  c->lattice[3] = c->lattice[4] = c->lattice[5] = 90.0;
  c->isQEq = 1; c->NMAXQEq = 500; c->QEq_tol = 1e-7; c->qstep = 1;   // rxmd.in defaults of examples/1-reaxff
  c->dt_fs = 0.25; c->Lex_fqs = 1.0; c->Lex_k = 2.0;                  // module.F90:164
}

int rxmd_hip_has_device_code(void) { return 1; }

int rxmd_hip_create(const rxmd_config *cfg, rxmd_handle *out) {
  if (!cfg || !out) return RXMD_E_ARG;
  *out = nullptr;
  rxmd_hip_engine *h = new (std::nothrow) rxmd_hip_engine;
  if (!h) return RXMD_E_NBUFFER;
  try {
    h->e = new Engine(*cfg);
  } catch (const EngineError &e) {
    g_create_error = e.msg;
    std::fprintf(stderr, "rxmd_hip_create: %s\n", e.msg.c_str());
    delete h;
    return e.code;
  } catch (const std::exception &e) {
    g_create_error = e.what();
    delete h;
    return RXMD_E_STATE;
  }
  *out = h;
  return RXMD_OK;
}

int rxmd_hip_destroy(rxmd_handle h) {
  if (!h) return RXMD_E_ARG;
  delete h->e;
  delete h;
  return RXMD_OK;
}

const char *rxmd_hip_last_error(rxmd_handle h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int rxmd_hip_set_atoms_rxff(rxmd_handle h, int natoms, const double *rec10) {
  if (!rec10) return RXMD_E_ARG;
  return guarded(h, [&](Engine &e) { e.set_atoms_rxff(natoms, rec10); });
}

int rxmd_hip_get_atoms_rxff(rxmd_handle h, double *rec10, int capacity) {
  int n = 0;
  const int rc = guarded(h, [&](Engine &e) { n = e.get_atoms_rxff(rec10, capacity); });
  return rc < 0 ? rc : n;
}

int rxmd_hip_get_atoms(rxmd_handle h, int capacity, long long *gid, int *type, double *pos, double *v, double *f, double *q) {
  int n = 0;
  const int rc = guarded(h, [&](Engine &e) {
    if (!e.atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
    n = e.N;
    if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity smaller than natoms");
    RX_HIP(hipStreamSynchronize(e.stream));
    if (gid) RX_HIP(hipMemcpy(gid, e.gid, sizeof(long long) * n, hipMemcpyDeviceToHost));
    if (type) RX_HIP(hipMemcpy(type, e.type, sizeof(int) * n, hipMemcpyDeviceToHost));
    if (q) RX_HIP(hipMemcpy(q, e.q, sizeof(double) * n, hipMemcpyDeviceToHost));
    std::vector<double> tmp(n);
    auto pull3 = [&](double *const src[3], double *dst) {
      for (int a = 0; a < 3; ++a) {
        RX_HIP(hipMemcpy(tmp.data(), src[a], sizeof(double) * n, hipMemcpyDeviceToHost));
        for (int i = 0; i < n; ++i) dst[3 * static_cast<size_t>(i) + a] = tmp[i];
      }
    };
    if (pos) pull3(e.pos, pos);
    if (v) pull3(e.vel, v);
    if (f) pull3(e.frc, f);
  });
  return rc < 0 ? rc : n;
}

int rxmd_hip_set_charges(rxmd_handle h, int natoms, const double *q) {
  return guarded(h, [&](Engine &e) {
    if (natoms != e.N || !q) throw EngineError(RXMD_E_ARG, "natoms mismatch");
    RX_HIP(hipMemcpy(e.q, q, sizeof(double) * natoms, hipMemcpyHostToDevice));
    e.lists_valid = false;   // ghost charges are refreshed by the next ghost build
  });
}

int rxmd_hip_set_velocities(rxmd_handle h, int natoms, const double *v) {
  return guarded(h, [&](Engine &e) {
    if (natoms != e.N || !v) throw EngineError(RXMD_E_ARG, "natoms mismatch");
    std::vector<double> tmp(natoms);
    for (int a = 0; a < 3; ++a) {
      for (int i = 0; i < natoms; ++i) tmp[i] = v[3 * static_cast<size_t>(i) + a];
      RX_HIP(hipMemcpy(e.vel[a], tmp.data(), sizeof(double) * natoms, hipMemcpyHostToDevice));
    }
  });
}

int rxmd_hip_get_bonds(rxmd_handle h, int capacity, int maxnb, int *count, long long *partner_gid, double *bo) {
  int n = 0;
  const int rc = guarded(h, [&](Engine &e) {
    if (!count || !partner_gid || !bo || capacity < e.N || maxnb < e.MAXNB) throw EngineError(RXMD_E_ARG, "capacity smaller than natoms or maxnb smaller than MAXNEIGHBS");
    if (!e.lists_valid) throw EngineError(RXMD_E_STATE, "no bond lists: call rxmd_hip_force first");
    RX_HIP(hipStreamSynchronize(e.stream));
    const int N = e.N, G = e.G;
    std::vector<int> off(static_cast<size_t>(N) + 1);
    std::vector<long long> g(G);
    RX_HIP(hipMemcpy(off.data(), e.boff, sizeof(int) * (static_cast<size_t>(N) + 1), hipMemcpyDeviceToHost));
    RX_HIP(hipMemcpy(g.data(), e.gid, sizeof(long long) * G, hipMemcpyDeviceToHost));
    const size_t nbd = static_cast<size_t>(off[N]);                 // the residents' bonds are the first boff[N] entries of the compact tables
    std::vector<int> nb(nbd); std::vector<double> b(nbd);
    RX_HIP(hipMemcpy(nb.data(), e.nbr, sizeof(int) * nbd, hipMemcpyDeviceToHost));
    RX_HIP(hipMemcpy(b.data(), e.bo0, sizeof(double) * nbd, hipMemcpyDeviceToHost));
    for (int i = 0; i < N; ++i) {
      const int c = std::min(off[i + 1] - off[i], e.MAXNB);
      count[i] = c;
      for (int s = 0; s < c; ++s) {
        partner_gid[static_cast<size_t>(i) * maxnb + s] = g[nb[static_cast<size_t>(off[i]) + s]];
        bo[static_cast<size_t>(i) * maxnb + s] = b[static_cast<size_t>(off[i]) + s];
      }
    }
    n = N;
  });
  return rc ? rc : n;
}

int rxmd_hip_get_shells(rxmd_handle h, double *spos3, int capacity) {
  int n = 0;
  const int rc = guarded(h, [&](Engine &e) {
    if (!e.ff.pqeq) throw EngineError(RXMD_E_STATE, "the engine was created without a PQEq parameter file");
    if (!spos3 || capacity < e.N) throw EngineError(RXMD_E_ARG, "capacity smaller than natoms");
    RX_HIP(hipStreamSynchronize(e.stream));
    std::vector<double> tmp(e.N);
    for (int a = 0; a < 3; ++a) {
      RX_HIP(hipMemcpy(tmp.data(), e.shl[a], sizeof(double) * e.N, hipMemcpyDeviceToHost));
      for (int i = 0; i < e.N; ++i) spos3[3 * static_cast<size_t>(i) + a] = tmp[i];
    }
    n = e.N;
  });
  return rc ? rc : n;
}

int rxmd_hip_set_shells(rxmd_handle h, int natoms, const double *spos3) {
  return guarded(h, [&](Engine &e) {
    if (!e.ff.pqeq) throw EngineError(RXMD_E_STATE, "the engine was created without a PQEq parameter file");
    if (natoms != e.N || !spos3) throw EngineError(RXMD_E_ARG, "natoms mismatch");
    std::vector<double> tmp(natoms);
    for (int a = 0; a < 3; ++a) {
      for (int i = 0; i < natoms; ++i) tmp[i] = spos3[3 * static_cast<size_t>(i) + a];
      RX_HIP(hipMemcpy(e.shl[a], tmp.data(), sizeof(double) * natoms, hipMemcpyHostToDevice));
    }
    e.lists_valid = false;
  });
}

int rxmd_hip_qeq(rxmd_handle h, int *iters, double *est) {
  return guarded(h, [&](Engine &e) {
    e.qeq();
    if (iters) *iters = e.nstep_qeq;
    if (est) *est = e.last_est;
  });
}

int rxmd_hip_force(rxmd_handle h, double pe[14]) {
  return guarded(h, [&](Engine &e) {
    e.force();
    if (pe) std::memcpy(pe, e.pe, sizeof(double) * 14);
  });
}

int rxmd_hip_step(rxmd_handle h, int nsteps) {
  return guarded(h, [&](Engine &e) {
    const auto t0 = std::chrono::steady_clock::now();
    e.step(nsteps);
    e.st.ms_step_total += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  });
}

int rxmd_hip_minimise(rxmd_handle h, double ftol, int max_loops, double *pe, long long *evaluations) {
  int loops = 0;
  const int rc = guarded(h, [&](Engine &e) { loops = e.minimise(ftol, max_loops > 0 ? max_loops : 500, pe, evaluations); });
  return rc ? rc : loops;
}

int rxmd_hip_last_qeq_iters(rxmd_handle h) {
  int n = 0;
  const int rc = guarded(h, [&](Engine &e) { n = e.nstep_qeq; });
  return rc ? rc : n;
}

int rxmd_hip_thermostat(rxmd_handle h, int mdmode, double treq_K, double vsfact, double gke_per_atom) {
  return guarded(h, [&](Engine &e) { e.thermostat(mdmode, treq_K, vsfact, gke_per_atom); });
}

int rxmd_hip_get_energy(rxmd_handle h, double *ke, double *qsum, double pe[14], double astr[6]) {
  return guarded(h, [&](Engine &e) {
    if (!e.atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
    double k = 0.0, qs = 0.0;
    e.kinetic_and_charge(k, qs);                       // main.F90:225-230, summed on the device in a fixed order
    if (ke) *ke = k;
    if (qsum) *qsum = qs;
    if (pe) std::memcpy(pe, e.pe, sizeof(double) * 14);
    if (astr) {     // accumulated since the previous read; reading resets the accumulators as PRINTE does (main.F90:270)
      RX_HIP(hipMemcpy(e.astr, e.scal + 48, sizeof(double) * 6, hipMemcpyDeviceToHost));
      RX_HIP(hipMemset(e.scal + 48, 0, sizeof(double) * 6));
      std::memcpy(astr, e.astr, sizeof(double) * 6);
    }
  });
}

// ---- the reference's own argument shapes (QEq(atype,pos,q) qeq.F90:2 ; FORCE(atype,pos,f,q) pot.F90:2) ----
static void upload_reference_arrays(Engine &e, int nbuffer, int natoms, const double *atype, const double *pos, const double *q) {
  if (natoms < 0 || (natoms == 0 && e.nprocs == 1) || nbuffer < natoms) throw EngineError(RXMD_E_ARG, "bad natoms/nbuffer");
  // FORCE right after QEq (src/main.F90:77-84) hands over the same atoms: keep ghosts, cells and both lists, refresh the charges only
  {
    bool same = e.atoms_set && e.lists_valid && natoms == e.N && e.last_atype.size() == static_cast<size_t>(natoms);
    if (same) same = std::memcmp(e.last_atype.data(), atype, sizeof(double) * natoms) == 0;
    for (int a = 0; a < 3 && same; ++a) same = std::memcmp(e.last_pos[a].data(), pos + a * static_cast<size_t>(nbuffer), sizeof(double) * natoms) == 0;
    // The shortcut decides whether the next qeq()/force() rebuilds ghosts and lists -- a COLLECTIVE exchange.  With several ranks it must be the
    // same decision everywhere (a rank that owns no atom compares equal trivially and would sit out the six-stage exchange its neighbours
    // enter): every rank keeps its lists only if every rank may.
    if (e.nprocs > 1) {
      double differ = same ? 0.0 : 1.0;
      e.allreduce_host(&differ, 1);
      same = differ == 0.0;
    }
    if (same) {
      if (q) RX_HIP(hipMemcpy(e.q, q, sizeof(double) * natoms, hipMemcpyHostToDevice));
      if (e.lex_pending && e.lex_p.size() == static_cast<size_t>(natoms)) {
        RX_HIP(hipMemcpy(e.qsfp, e.lex_p.data(), sizeof(double) * natoms, hipMemcpyHostToDevice));
        RX_HIP(hipMemcpy(e.qsfv, e.lex_v.data(), sizeof(double) * natoms, hipMemcpyHostToDevice));
      }
      return;
    }
  }
  const bool lex = e.lex_pending && e.lex_p.size() == static_cast<size_t>(natoms);
  e.last_atype.assign(atype, atype + natoms);
  for (int a = 0; a < 3; ++a) e.last_pos[a].assign(pos + a * static_cast<size_t>(nbuffer), pos + a * static_cast<size_t>(nbuffer) + natoms);
  // every call after the first (the engine is sized, its tables exist): the caller's arrays go to the device as they are -- no
  // 10-double records, no second copy of the coordinates; the packed type is split there (k_split_atype)
  const bool level1_records = e.opt.level1_records;     // (experiments build: A/B against the old path through 10-double host records)
  if (e.tables_ready && natoms <= e.rows10 && natoms < e.NB && !level1_records) {
    e.set_atoms_arrays(natoms, atype, pos, pos + static_cast<size_t>(nbuffer), pos + 2 * static_cast<size_t>(nbuffer), q,
                       lex ? e.lex_p.data() : nullptr, lex ? e.lex_v.data() : nullptr);
    return;
  }
  std::vector<double> rec(10 * static_cast<size_t>(natoms), 0.0);
  for (int i = 0; i < natoms; ++i) {
    const double r[3] = {pos[i], pos[static_cast<size_t>(nbuffer) + i], pos[2 * static_cast<size_t>(nbuffer) + i]};   // pos(NBUFFER,3) column-major
    double *o = rec.data() + 10 * static_cast<size_t>(i);
    for (int a = 0; a < 3; ++a) o[a] = (e.box.Hi[a][0] * r[0] + e.box.Hi[a][1] * r[1] + e.box.Hi[a][2] * r[2]) - e.box.obox[a];
    o[6] = q ? q[i] : 0.0;
    o[7] = atype[i];
    if (lex) { o[8] = e.lex_p[i]; o[9] = e.lex_v[i]; }
  }
  e.set_atoms_rxff(natoms, rec.data());
  // keep the caller's real coordinates bit for bit (the record round trip is only used for sizing/setup)
  std::vector<double> tmp(natoms);
  for (int a = 0; a < 3; ++a) {
    for (int i = 0; i < natoms; ++i) tmp[i] = pos[a * static_cast<size_t>(nbuffer) + i];
    RX_HIP(hipMemcpy(e.pos[a], tmp.data(), sizeof(double) * natoms, hipMemcpyHostToDevice));
  }
}

// isQEq = 2 starts the single CG step from qs = fqs*qsfp + (1-fqs)*q (qeq.F90:51-57): the host's qsfp must have come with put_lex
static void require_lex(Engine &e, int natoms) {
  if (e.cfg.isQEq == 2 && !(e.lex_pending && e.lex_p.size() == static_cast<size_t>(natoms)))
    throw EngineError(RXMD_E_ARG, "isQEq = 2 through the array-shaped entry points needs rxmd_hip_put_lex(qsfp, qsfv) before every QEq/PQEq call");
}
int rxmd_hip_put_lex(rxmd_handle h, int natoms, const double *qsfp, const double *qsfv) {
  if (!qsfp || !qsfv || natoms < 0) return RXMD_E_ARG;
  return guarded(h, [&](Engine &e) { e.lex_p.assign(qsfp, qsfp + natoms); e.lex_v.assign(qsfv, qsfv + natoms); e.lex_pending = true; });
}
int rxmd_hip_get_lex(rxmd_handle h, int natoms, double *qsfp, double *qsfv) {
  if (!qsfp || !qsfv) return RXMD_E_ARG;
  return guarded(h, [&](Engine &e) {
    if (!e.atoms_set || natoms != e.N) throw EngineError(RXMD_E_ARG, "get_lex: natoms does not match the atoms of the last call");
    RX_HIP(hipStreamSynchronize(e.stream));
    RX_HIP(hipMemcpy(qsfp, e.qsfp, sizeof(double) * natoms, hipMemcpyDeviceToHost));
    RX_HIP(hipMemcpy(qsfv, e.qsfv, sizeof(double) * natoms, hipMemcpyDeviceToHost));
  });
}

int rxmd_hip_QEq(rxmd_handle h, int nbuffer, int natoms, const double *atype, const double *pos, double *q) {
  if (!atype || !pos || !q) return RXMD_E_ARG;
  return guarded(h, [&](Engine &e) {
    require_lex(e, natoms);
    upload_reference_arrays(e, nbuffer, natoms, atype, pos, q);
    e.lex_pending = false;
    e.qeq();
    RX_HIP(hipMemcpy(q, e.q, sizeof(double) * natoms, hipMemcpyDeviceToHost));
  });
}

int rxmd_hip_FORCE(rxmd_handle h, int nbuffer, int natoms, const double *atype, const double *pos, double *f, const double *q, double pe[14]) {
  if (!atype || !pos || !f || !q) return RXMD_E_ARG;
  return guarded(h, [&](Engine &e) {
    upload_reference_arrays(e, nbuffer, natoms, atype, pos, q);
    e.force();
    for (int a = 0; a < 3; ++a)                      // f(NBUFFER,3) column-major: component a of the residents is contiguous
      RX_HIP(hipMemcpy(f + a * static_cast<size_t>(nbuffer), e.frc[a], sizeof(double) * natoms, hipMemcpyDeviceToHost));
    if (pe) std::memcpy(pe, e.pe, sizeof(double) * 14);
  });
}

// PQEq in the reference's shapes: spos(NBUFFER,3) column-major travels with the call (the reference keeps it in module atoms and
// moves it with the atoms in COPYATOMS, comm.F90:122-167)
static void upload_shells(Engine &e, int nbuffer, int natoms, const double *spos) {
  if (!e.ff.pqeq) throw EngineError(RXMD_E_STATE, "the engine was created without a PQEq parameter file");
  for (int a = 0; a < 3; ++a) RX_HIP(hipMemcpy(e.shl[a], spos + a * static_cast<size_t>(nbuffer), sizeof(double) * natoms, hipMemcpyHostToDevice));
}
int rxmd_hip_PQEq(rxmd_handle h, int nbuffer, int natoms, const double *atype, const double *pos, double *q, double *spos) {
  if (!atype || !pos || !q || !spos) return RXMD_E_ARG;
  return guarded(h, [&](Engine &e) {
    require_lex(e, natoms);
    upload_reference_arrays(e, nbuffer, natoms, atype, pos, q);
    e.lex_pending = false;
    upload_shells(e, nbuffer, natoms, spos);
    e.qeq();
    RX_HIP(hipMemcpy(q, e.q, sizeof(double) * natoms, hipMemcpyDeviceToHost));
    for (int a = 0; a < 3; ++a) RX_HIP(hipMemcpy(spos + a * static_cast<size_t>(nbuffer), e.shl[a], sizeof(double) * natoms, hipMemcpyDeviceToHost));
  });
}
int rxmd_hip_FORCE_pqeq(rxmd_handle h, int nbuffer, int natoms, const double *atype, const double *pos, double *f, const double *q, const double *spos, double pe[14]) {
  if (!atype || !pos || !f || !q || !spos) return RXMD_E_ARG;
  return guarded(h, [&](Engine &e) {
    upload_reference_arrays(e, nbuffer, natoms, atype, pos, q);
    upload_shells(e, nbuffer, natoms, spos);
    e.force();
    for (int a = 0; a < 3; ++a)                      // f(NBUFFER,3) column-major: component a of the residents is contiguous
      RX_HIP(hipMemcpy(f + a * static_cast<size_t>(nbuffer), e.frc[a], sizeof(double) * natoms, hipMemcpyDeviceToHost));
    if (pe) std::memcpy(pe, e.pe, sizeof(double) * 14);
  });
}

// ---- introspection ----
int rxmd_hip_get_stats(rxmd_handle h, rxmd_stats *out) {
  if (!out) return RXMD_E_ARG;
  return guarded(h, [&](Engine &e) {
    if (e.atoms_set && e.lists_valid) {
      RX_HIP(hipStreamSynchronize(e.stream));
      e.collect_timers();
      std::vector<int> n10(e.N), nb(e.G);
      RX_HIP(hipMemcpy(n10.data(), e.n10, sizeof(int) * e.N, hipMemcpyDeviceToHost));
      RX_HIP(hipMemcpy(nb.data(), e.nbrcnt, sizeof(int) * e.G, hipMemcpyDeviceToHost));
      long long s10 = 0, sb = 0; int m10 = 0, mb = 0;
      for (int v : n10) { v &= rxmd::N10_COUNT; s10 += v; m10 = std::max(m10, v); }
      for (int v : nb) { sb += v; mb = std::max(mb, v); }
      e.st.nnz10 = s10; e.st.nbonds = sb; e.st.max_n10 = m10; e.st.max_nb = mb;
    }
    e.st.natoms = e.N;
    e.st.n_boundary_rows = (e.multi() && e.lists_valid && !e.rows_split_pending_invalid()) ? e.n_bnd : 0;
    e.st.win_groups = e.win_groups; e.st.win_max_units = e.win_maxunits; e.st.win_in_use = e.win_used ? 1 : 0;
    // the matrix pass is timed on a sample of its launches (Engine::qeq): average of the timed ones x all of them
    e.st.spmv_launches_timed = e.pass_timed_n;
    e.st.ms_qeq_spmv = e.pass_timed_n > 0 ? e.pass_timed_ms / static_cast<double>(e.pass_timed_n) * static_cast<double>(e.st.spmv_launches) : 0.0;
    *out = e.st;
  });
}

int rxmd_hip_reset_timers(rxmd_handle h) {
  return guarded(h, [&](Engine &e) {
    e.st.ms_qeq = e.st.ms_qeq_list = e.st.ms_qeq_spmv = e.st.ms_force = e.st.ms_lists = e.st.ms_bo = e.st.ms_nonbond = e.st.ms_bonded = e.st.ms_step_total = 0.0;
    e.st.spmv_launches = 0; e.st.spmv_noop_launches = 0; e.st.qeq_iters_total = 0; e.st.qeq_calls = 0;
    e.st.ms_ghost_build = e.st.ms_migrate = e.st.ms_halo = e.st.ms_halo_exposed = e.st.ms_allreduce = e.st.ms_fold = 0.0;
    e.st.halo_calls = e.st.allreduce_calls = 0; e.st.timer_pairs_dropped = 0; e.pass_timed_ms = 0.0; e.pass_timed_n = 0; e.st.spmv_launches_timed = 0;
    e.st.ms_k_list10 = e.st.ms_k_nonbond = e.st.ms_k_e3b = e.st.ms_k_e4b = e.st.ms_k_ehb = e.st.ms_k_bondorder = e.st.ms_k_assemble = e.st.ms_k_winbuild = e.st.ms_k_blist = e.st.ms_bond_exposed = 0.0;
  });
}

int rxmd_hip_set_qeq_mode(rxmd_handle h, int mode) {
  return guarded(h, [&](Engine &e) {
    if (mode != 0 && mode != 1) throw EngineError(RXMD_E_ARG, "qeq_mode must be 0 or 1");
    e.cfg.qeq_mode = mode;
  });
}

int rxmd_hip_get_table(rxmd_handle h, int which, double *out, int capacity) {
  int n = 0;
  const int rc = guarded(h, [&](Engine &e) {
    if (!e.tables_ready) throw EngineError(RXMD_E_STATE, "tables are built when atoms are first set");
    n = e.ff.nboty;
    if (!out || capacity < n * rxmd::NTABLE) throw EngineError(RXMD_E_ARG, "capacity too small");
    const std::vector<double> *src[5] = {&e.ff.tblEvdw, &e.ff.tbldEvdw, &e.ff.tblEclmb, &e.ff.tbldEclmb, &e.ff.tblQEq};
    if (which < 0 || which > 4) throw EngineError(RXMD_E_ARG, "which must be 0..4");
    for (int r = 1; r <= n; ++r)
      for (int i = 1; i <= rxmd::NTABLE; ++i) out[static_cast<size_t>(r - 1) * rxmd::NTABLE + (i - 1)] = (*src[which])[static_cast<size_t>(r) * (rxmd::NTABLE + 2) + i];
  });
  return rc < 0 ? rc : n;
}

int rxmd_hip_get_cutoffs(rxmd_handle h, double *rc, int capacity, double *maxrc) {
  int n = 0;
  const int r = guarded(h, [&](Engine &e) {
    if (!e.tables_ready) throw EngineError(RXMD_E_STATE, "cutoffs are computed when atoms are first set");
    n = e.ff.nboty;
    if (rc) { if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity too small"); for (int k = 1; k <= n; ++k) rc[k - 1] = e.ff.bond[k].rc; }
    if (maxrc) *maxrc = e.ff.maxrc;
  });
  return r < 0 ? r : n;
}

int rxmd_hip_debug_get(rxmd_handle h, int what, double *out, int capacity) {
  int n = 0;
  const int rc = guarded(h, [&](Engine &e) {
    if (!e.atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
    RX_HIP(hipStreamSynchronize(e.stream));
    auto pull_d = [&](const double *src, int cnt, int stride, int off) {
      std::vector<double> t(cnt);
      RX_HIP(hipMemcpy(t.data(), src, sizeof(double) * cnt, hipMemcpyDeviceToHost));
      for (int i = 0; i < cnt; ++i) out[static_cast<size_t>(i) * stride + off] = t[i];
    };
    const int G = e.G, N = e.N;
    switch (what) {
      case 0: n = G; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity"); pull_d(e.delta, G, 1, 0); break;
      case 1: n = G; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity"); pull_d(e.deltap, G, 1, 0); break;
      case 2: { n = G; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity"); std::vector<int> t(G); RX_HIP(hipMemcpy(t.data(), e.nbrcnt, sizeof(int) * G, hipMemcpyDeviceToHost)); for (int i = 0; i < G; ++i) out[i] = t[i]; break; }
      case 3: n = G; if (capacity < 3 * n) throw EngineError(RXMD_E_ARG, "capacity"); for (int a = 0; a < 3; ++a) pull_d(e.pos[a], G, 3, a); break;
      case 4: { n = G; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity"); std::vector<long long> t(G); RX_HIP(hipMemcpy(t.data(), e.gid, sizeof(long long) * G, hipMemcpyDeviceToHost)); for (int i = 0; i < G; ++i) out[i] = static_cast<double>(t[i]); break; }
      case 5: { n = G; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity"); std::vector<int> t(G); RX_HIP(hipMemcpy(t.data(), e.type, sizeof(int) * G, hipMemcpyDeviceToHost)); for (int i = 0; i < G; ++i) out[i] = t[i]; break; }
      case 6: { n = N; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity"); std::vector<int> t(N); RX_HIP(hipMemcpy(t.data(), e.n10, sizeof(int) * N, hipMemcpyDeviceToHost)); for (int i = 0; i < N; ++i) out[i] = t[i] & rxmd::N10_COUNT; break; }
      case 7: {
        n = N; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity");
        std::vector<int> c(N); RX_HIP(hipMemcpy(c.data(), e.n10, sizeof(int) * N, hipMemcpyDeviceToHost));
        for (int &v : c) v &= rxmd::N10_COUNT;
        std::vector<double> row(e.S10);
        for (int i = 0; i < N; ++i) {
          RX_HIP(hipMemcpy(row.data(), e.hess + static_cast<size_t>(i) * e.S10, sizeof(double) * c[i], hipMemcpyDeviceToHost));
          double s = 0; for (int k = 0; k < c[i]; ++k) s += row[k];
          out[i] = s;
        }
        break;
      }
      case 8: n = G; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity"); pull_d(e.cd, G, 1, 0); break;
      case 9: n = G; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity"); pull_d(e.q, G, 1, 0); break;
      case 10: n = G; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity"); pull_d(e.cc_, G, 1, 0); break;
      case 14: {   // RXMD_POISON_ALLOC: {1 if the pattern is on, the LAST entry of row 0 of the 10 A value array} -- an element no kernel writes (rows are shorter than their slot)
        n = 2; if (capacity < 2) throw EngineError(RXMD_E_ARG, "capacity");
        out[0] = e.poison_on() ? 1.0 : 0.0;
        RX_HIP(hipMemcpy(out + 1, e.hess + (e.S10 - 1), sizeof(double), hipMemcpyDeviceToHost));
        break;
      }
      case 13: {   // Est of the start vector and after every CG iteration of the last QEq call (what the reference prints with -DQEQDUMP, qeq.F90:117)
        n = static_cast<int>(e.est_trace.size()); if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity");
        for (int k = 0; k < n; ++k) out[k] = e.est_trace[k];
        break;
      }
      case 11: {   // window form of the 10 A matrix, checked on the host: per resident (atom order) the number of list entries whose 16-bit slot leads
                   // back to the entry's own cell-sorted position and ghost flag through the group's window (== n10 when the window is right; -1: no windows)
        n = N; if (capacity < n) throw EngineError(RXMD_E_ARG, "capacity");
        if (!e.win_valid) { for (int i = 0; i < N; ++i) out[i] = -1.0; break; }
        RX_HIP(hipStreamSynchronize(e.stream));
        const size_t nrs = static_cast<size_t>(e.win_groups) * rxmd::WIN_ROWS;
        std::vector<int> c(N), rs(nrs), wc(e.win_groups), wk(static_cast<size_t>(e.win_groups) * rxmd::WIN_MAXUNITS);
        RX_HIP(hipMemcpy(c.data(), e.n10, sizeof(int) * N, hipMemcpyDeviceToHost));
        RX_HIP(hipMemcpy(rs.data(), e.rows_sorted, sizeof(int) * nrs, hipMemcpyDeviceToHost));
        RX_HIP(hipMemcpy(wc.data(), e.win_cnt, sizeof(int) * e.win_groups, hipMemcpyDeviceToHost));
        RX_HIP(hipMemcpy(wk.data(), e.win_k, sizeof(int) * wk.size(), hipMemcpyDeviceToHost));
        std::vector<int> ent(e.S10); std::vector<unsigned short> sl(e.S10);
        for (int i = 0; i < N; ++i) out[i] = -2.0;                 // a resident that is in no group
        for (size_t r = 0; r < nrs; ++r) {
          const int i = rs[r], g = static_cast<int>(r / rxmd::WIN_ROWS);
          if (i < 0 || i >= N) continue;                           // unused row of a cell column's last group
          const int cnt = c[i] & rxmd::N10_COUNT;
          RX_HIP(hipMemcpy(ent.data(), e.nb10 + static_cast<size_t>(i) * e.S10, sizeof(int) * cnt, hipMemcpyDeviceToHost));
          RX_HIP(hipMemcpy(sl.data(), e.sl10 + static_cast<size_t>(i) * e.S10, sizeof(unsigned short) * cnt, hipMemcpyDeviceToHost));
          int good = 0;
          for (int k = 0; k < cnt; ++k) {
            const unsigned en = static_cast<unsigned>(ent[k]);
            const int slot = sl[k] & 0x7fff, unit = slot / rxmd::WIN_UNIT;
            const bool gh = (sl[k] & 0x8000) != 0;
            if (unit < wc[g] && wk[static_cast<size_t>(g) * rxmd::WIN_MAXUNITS + unit] + (slot % rxmd::WIN_UNIT) == static_cast<int>(en & rxmd::NB10_IDX_MASK) && gh == ((en & rxmd::NB10_GHOST) != 0)) ++good;
          }
          out[i] = good;
        }
        break;
      }
      case 12: {   // read-bandwidth probe over the whole value array (a plain 16-byte-per-lane read; bench.py quotes the matrix pass next to it): out = {ms, bytes} for a few grid sizes
        n = 4; if (capacity < 8) throw EngineError(RXMD_E_ARG, "capacity");
        const int grids[4] = {2048, 8192, 32768, 131072};
        for (int g = 0; g < 4; ++g) { out[2 * g] = rxmd::stream_probe_ms(e, grids[g]); out[2 * g + 1] = static_cast<double>(e.rows10) * e.S10 * 8.0; }
        n = 8; break;
      }
#ifdef RXMD_EXPERIMENTS
      case 104: n = 20; if (capacity < 20) throw EngineError(RXMD_E_ARG, "capacity"); rxmd::spmv_isolated_ms(e, out); break;   // real window pass / row pass back to back (experiments)
      case 105: n = 5; if (capacity < 5) throw EngineError(RXMD_E_ARG, "capacity"); rxmd::spmv_tile_probe_ms(e, out); break;   // half-storage pass over 3-D tiles: timing probe (experiments)
      case 102: n = 7; if (capacity < 7) throw EngineError(RXMD_E_ARG, "capacity"); rxmd::spmv_bisect_ms(e, out); break;   // stripped-down forms of the row kernel (experiments)
#endif
      default: throw EngineError(RXMD_E_ARG, "unknown debug tap");
    }
  });
  return rc < 0 ? rc : n;
}

int rxmd_hip_copy_to_host(const double *dev, double *host, long long n) {
  if (n < 0 || (n > 0 && (!dev || !host))) return RXMD_E_ARG;
  return (n == 0 || hipMemcpy(host, dev, sizeof(double) * static_cast<size_t>(n), hipMemcpyDeviceToHost) == hipSuccess) ? RXMD_OK : RXMD_E_HIP;
}
int rxmd_hip_copy_to_device(double *dev, const double *host, long long n) {
  if (n < 0 || (n > 0 && (!dev || !host))) return RXMD_E_ARG;
  return (n == 0 || hipMemcpy(dev, host, sizeof(double) * static_cast<size_t>(n), hipMemcpyHostToDevice) == hipSuccess) ? RXMD_OK : RXMD_E_HIP;
}
int rxmd_hip_device_count(void) {
  int n = 0;
  return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

int rxmd_hip_set_comm(rxmd_handle h, const rxmd_comm_ops *ops) {
  return guarded(h, [&](Engine &e) {
    if (!ops) { e.has_comm = false; return; }
    e.comm = *ops; e.has_comm = true;
  });
}

int rxmd_hip_comm_init_rccl(rxmd_handle h, const unsigned char id128[128], int rank, int world) {
  return guarded(h, [&](Engine &e) {
    if (!id128) throw EngineError(RXMD_E_ARG, "NULL unique id");
    e.rccl_init(id128, rank, world);
  });
}

int rxmd_hip_set_exchange_buffers(rxmd_handle h, double *send, double *recv, long long ndoubles) {
  return guarded(h, [&](Engine &e) {
    if (!send || !recv || ndoubles < 1024) throw EngineError(RXMD_E_ARG, "bad exchange buffers");
    if (e.xbuf_owned) { (void)hipFree(e.xbuf_send); (void)hipFree(e.xbuf_recv); }
    e.xbuf_send = send; e.xbuf_recv = recv; e.xbuf_doubles = static_cast<size_t>(ndoubles); e.xbuf_owned = false;
  });
}

int rxmd_host_comm_selftest(const rxmd_comm_ops *ops, int myid, int nprocs) {
  if (!ops || !ops->exchange || !ops->allreduce_sum || nprocs < 1 || myid < 0 || myid >= nprocs) return RXMD_E_ARG;
  const int right = (myid + 1) % nprocs, left = (myid + nprocs - 1) % nprocs;
  std::vector<double> snd(4096), rcv(4096);
  for (int dir = 0; dir < 2; ++dir) {
    const int to = dir ? left : right, from = dir ? right : left;
    const long long ns = 7 + 13 * myid + dir;                 // ragged lengths, different per rank
    for (long long k = 0; k < ns; ++k) snd[k] = 1000.0 * myid + k + 0.5 * dir;
    const long long nr = ops->exchange(ops->ctx, to, snd.data(), ns, from, rcv.data(), 4096);
    if (nr != 7 + 13 * from + dir) return RXMD_E_COMM;
    for (long long k = 0; k < nr; ++k) if (rcv[k] != 1000.0 * from + k + 0.5 * dir) return RXMD_E_COMM;
    const long long nz = ops->exchange(ops->ctx, to, snd.data(), 0, from, rcv.data(), 4096);   // empty message (comm.F90:321-327)
    if (nz != 0) return RXMD_E_COMM;
    if (ops->exchange_known) {                                  // size known to the receiver: payload only
      const long long nk = ops->exchange_known(ops->ctx, to, snd.data(), ns, from, rcv.data(), 7 + 13 * from + dir);
      if (nk != 7 + 13 * from + dir) return RXMD_E_COMM;
      for (long long k = 0; k < nk; ++k) if (rcv[k] != 1000.0 * from + k + 0.5 * dir) return RXMD_E_COMM;
    }
  }
  double v[3] = {1.0 * (myid + 1), 0.25, -2.0 * myid};
  if (ops->allreduce_sum(ops->ctx, v, 3)) return RXMD_E_COMM;
  const double n = nprocs;
  if (v[0] != n * (n + 1) / 2 || v[1] != 0.25 * n || v[2] != -n * (n - 1)) return RXMD_E_COMM;
  return RXMD_OK;
}

// ---- host front-end helpers ---------------------------------------------------------------------
static int g_host_lg = 0;
int rxmd_host_ffield_lg(int on) { const int was = g_host_lg; g_host_lg = on != 0; return was; }
// geninit (reference init/geninit.F90:399-575)
long long rxmd_host_geninit(const char *ffield_path, int natoms0, const char *elem4, const double *frac, const double lattice[6], const int mc[3],
                            const int vprocs[3], int myid, double *rec10, long long capacity, double lattice_out[6]) {
  if (!ffield_path || !elem4 || !frac || natoms0 < 1) return RXMD_E_ARG;
  rxmd::ForceField ff;
  try { ff.parse(ffield_path, g_host_lg != 0); } catch (const std::exception &) { return RXMD_E_FFIELD; }
  std::vector<int> t0(natoms0, 0);
  for (int i = 0; i < natoms0; ++i) {
    std::string nm(elem4 + 4 * static_cast<size_t>(i));
    for (int t = 1; t <= ff.nso; ++t) if (ff.atom[t].name == nm) { t0[i] = t; break; }
    if (!t0[i]) return RXMD_E_ARG;
  }
  const long long ntot = static_cast<long long>(natoms0) * mc[0] * mc[1] * mc[2];
  std::vector<double> p(3 * static_cast<size_t>(ntot));
  double rmin[3] = {1e300, 1e300, 1e300};
  long long n = 0;
  for (int ix = 0; ix < mc[0]; ++ix) for (int iy = 0; iy < mc[1]; ++iy) for (int iz = 0; iz < mc[2]; ++iz)
    for (int i = 0; i < natoms0; ++i, ++n) {                               // geninit.F90:447-462: order defines the global id
      const double r[3] = {(frac[3 * i] + ix) / mc[0], (frac[3 * i + 1] + iy) / mc[1], (frac[3 * i + 2] + iz) / mc[2]};
      for (int a = 0; a < 3; ++a) { p[3 * n + a] = r[a]; rmin[a] = std::min(rmin[a], r[a]); }
    }
  for (long long k = 0; k < ntot; ++k)
    for (int a = 0; a < 3; ++a) p[3 * k + a] = std::fmod(p[3 * k + a] - rmin[a], 1.0) + 1e-9;   // geninit.F90:465-478
  if (lattice_out) { for (int a = 0; a < 3; ++a) lattice_out[a] = lattice[a] * mc[a]; for (int a = 3; a < 6; ++a) lattice_out[a] = lattice[a]; }
  const double lbox[3] = {1.0 / vprocs[0], 1.0 / vprocs[1], 1.0 / vprocs[2]};
  const int vi = myid % vprocs[0], vj = (myid / vprocs[0]) % vprocs[1], vk = myid / (vprocs[0] * vprocs[1]);
  const double obox[3] = {lbox[0] * vi, lbox[1] * vj, lbox[2] * vk};
  long long cnt = 0;
  for (long long k = 0; k < ntot; ++k) {
    const int i = static_cast<int>(p[3 * k] * vprocs[0]), j = static_cast<int>(p[3 * k + 1] * vprocs[1]), l = static_cast<int>(p[3 * k + 2] * vprocs[2]);
    if (i + j * vprocs[0] + l * vprocs[0] * vprocs[1] != myid) continue;  // geninit.F90:495-500
    if (rec10) {
      if (cnt >= capacity) return RXMD_E_NBUFFER;
      double *o = rec10 + 10 * cnt;
      for (int a = 0; a < 3; ++a) o[a] = p[3 * k + a] - obox[a];
      o[3] = o[4] = o[5] = 0.0; o[6] = 0.0;
      o[7] = t0[k % natoms0] + (k + 1) * 1e-13 + 1e-14;                    // geninit.F90:459
      o[8] = 0.0; o[9] = 0.0;
    }
    ++cnt;
  }
  return cnt;
}

int rxmd_host_ffield_table(const char *ffield_path, const long long *npt, int which, double *out, long long capacity) {
  if (!ffield_path || !out) return RXMD_E_ARG;
  rxmd::ForceField ff;
  try { ff.parse(ffield_path, g_host_lg != 0); } catch (const std::exception &) { return RXMD_E_FFIELD; }
  std::vector<long long> n(ff.nso + 2, 1);
  if (npt) for (int t = 1; t <= ff.nso; ++t) n[t] = npt[t];
  ff.compute_cutoffs(n);
  ff.build_taper(10.0);
  ff.build_tables();
  if (which >= 0 && which <= 4) {
    if (capacity < static_cast<long long>(ff.nboty) * rxmd::NTABLE) return RXMD_E_ARG;
    const std::vector<double> *src[5] = {&ff.tblEvdw, &ff.tbldEvdw, &ff.tblEclmb, &ff.tbldEclmb, &ff.tblQEq};
    for (int r = 1; r <= ff.nboty; ++r)
      for (int i = 1; i <= rxmd::NTABLE; ++i) out[static_cast<size_t>(r - 1) * rxmd::NTABLE + (i - 1)] = (*src[which])[static_cast<size_t>(r) * (rxmd::NTABLE + 2) + i];
  } else if (which == 5) {
    if (capacity < ff.nboty + 1) return RXMD_E_ARG;
    for (int r = 1; r <= ff.nboty; ++r) out[r - 1] = ff.bond[r].rc;
    out[ff.nboty] = ff.maxrc;
  } else if (which == 6) {
    if (capacity < 5 + 3 * ff.nso) return RXMD_E_ARG;
    out[0] = ff.nso; out[1] = ff.nboty; out[2] = ff.nvaty; out[3] = ff.ntoty; out[4] = ff.nhbty;
    for (int t = 1; t <= ff.nso; ++t) { out[4 + t] = ff.atom[t].chi; out[4 + ff.nso + t] = ff.atom[t].eta; out[4 + 2 * ff.nso + t] = ff.atom[t].mass; }
  } else return RXMD_E_ARG;
  return ff.nboty;
}

// ReadBIN header + records (reference src/fileio.F90:444-555)
long long rxmd_host_read_rxff(const char *path, int myid, double lattice_out[6], int vprocs_out[3], double *rec10, long long capacity) {
  std::ifstream in(path, std::ios::binary);
  if (!in) return RXMD_E_ARG;
  int head[4];
  in.read(reinterpret_cast<char *>(head), 16);
  if (!in || head[0] < 1 || head[0] != head[1] * head[2] * head[3]) return RXMD_E_ARG;
  const int np = head[0];
  if (myid < 0 || myid >= np) return RXMD_E_ARG;
  std::vector<int> nat(np);
  in.read(reinterpret_cast<char *>(nat.data()), 4 * np);
  int cur = 0;
  in.read(reinterpret_cast<char *>(&cur), 4);
  double lat[6];
  in.read(reinterpret_cast<char *>(lat), 48);
  if (!in) return RXMD_E_ARG;
  if (lattice_out) std::memcpy(lattice_out, lat, 48);
  if (vprocs_out) { vprocs_out[0] = head[1]; vprocs_out[1] = head[2]; vprocs_out[2] = head[3]; }
  if (!rec10) return nat[myid];
  if (capacity < nat[myid]) return RXMD_E_NBUFFER;
  long long skip = 0;
  for (int p = 0; p < myid; ++p) skip += nat[p];
  in.seekg(static_cast<std::streamoff>(skip) * 80, std::ios::cur);
  in.read(reinterpret_cast<char *>(rec10), static_cast<std::streamsize>(nat[myid]) * 80);
  return in ? nat[myid] : RXMD_E_ARG;
}

}  // extern "C"
