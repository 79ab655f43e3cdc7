// minimise.hip -- the reference's geometry minimiser (mdmode 10, src/cg.F90) on the device-resident QEq + FORCE primitive.
//   ConjugateGradient      cg.F90:26-98     Polak-Ribiere directions from the FORCE output, energy criterion |dE| <= ftol * GNATOMS
//   BracketSearchRange     cg.F90:101-140   step doubled from 1e-2/GNATOMS until the Armijo rule (WolfeConditions, cg.F90:143-209) fails
//   LineMinimization       cg.F90:212-241   golden-section search on [0, bracket] (cg.F90:244-283), then the atoms move and migrate
//   EvaluateEnergyWithStep cg.F90:358-392   COPYATOMS(MODE_MOVE) + QEq + FORCE at pos + step * p on a COPY of the atoms
// Positions, search direction and gradients stay in HBM; a trial point is the engine's own migrate() + qeq() + force() on the
// displaced atoms, after which the base state is restored from a device snapshot.  Vectors that must follow the atoms through a
// migration ride in the velocity slot, as the reference sends them through COPYATOMS' `v` argument (cg.F90:230,236).
// Restated, not copied: the reference re-evaluates the energy and forces of the unmoved atoms in every bracketing round and
// evaluates a second Wolfe condition it never uses (cg.F90:127: `.not.WolfeC1 .or. .not.WolfeC1`); both are left out -- they
// change nothing but the start vector of the next charge solve.
// Deliberate deviation: LineMinimization calls MigrateVec3D(pos, p, g, stepl) (cg.F90:226,299-310), which migrates the DIRECTION p displaced
// along g with an uninitialised atype -- the compaction of COPYATOMS(MODE_MOVE) then destroys p.  Here the gradient g
// rides with the atoms displaced along p, which is what the surrounding code needs: the search direction and the gradient of the next
// loop in the atoms' new order.
// State on return: positions, charges, forces, PE(1:13), ghosts and lists are those of the final positions on BOTH exits (energy
// criterion met, or max_loops reached -- the bracketing of a further loop is not started then).
#include "engine.h"

#include <cmath>
#include <cstdio>
#include <vector>

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }

__global__ void k_axpy3(int n, double s, const double *__restrict__ px, const double *__restrict__ py, const double *__restrict__ pz,
                        double *__restrict__ x, double *__restrict__ y, double *__restrict__ z) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  x[i] += s * px[i]; y[i] += s * py[i]; z[i] += s * pz[i];
}
// p = b * p + g   (cg.F90:91)
__global__ void k_direction3(int n, double b, const double *__restrict__ gx, const double *__restrict__ gy, const double *__restrict__ gz,
                             double *__restrict__ px, double *__restrict__ py, double *__restrict__ pz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  px[i] = b * px[i] + gx[i]; py[i] = b * py[i] + gy[i]; pz[i] = b * pz[i] + gz[i];
}
// DotProductVec3D (cg.F90:322-339): per-block partial sums in a fixed order, finished on the host
__global__ void __launch_bounds__(256) k_dot3(int n, const double *__restrict__ ax, const double *__restrict__ ay, const double *__restrict__ az,
                                               const double *__restrict__ bx, const double *__restrict__ by, const double *__restrict__ bz, double *__restrict__ partials) {
  __shared__ double sm[256];
  double s = 0.0;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) s += ax[i] * bx[i] + ay[i] * by[i] + az[i] * bz[i];
  sm[threadIdx.x] = s;
  __syncthreads();
  for (int t = 128; t > 0; t >>= 1) { if (threadIdx.x < t) sm[threadIdx.x] += sm[threadIdx.x + t]; __syncthreads(); }
  if (threadIdx.x == 0) partials[blockIdx.x] = sm[0];
}

namespace {
struct Vec3 { double *c[3] = {nullptr, nullptr, nullptr}; };
struct Snapshot { Vec3 pos, vel, shl; double *q = nullptr, *qsfp = nullptr, *qsfv = nullptr; int *type = nullptr; long long *gid = nullptr; int N = 0; };

struct Minimiser {
  Engine &e;
  Vec3 P, G, Gold;           // search direction, new and old gradient (= FORCE output, as the reference uses it)
  Snapshot snap;
  double gnatoms = 0.0;
  long long evals = 0;
  explicit Minimiser(Engine &en) : e(en) {
    auto al = [&](Vec3 &v) { for (int a = 0; a < 3; ++a) { RX_HIP(hipMalloc(reinterpret_cast<void **>(&v.c[a]), sizeof(double) * e.NB)); RX_HIP(hipMemset(v.c[a], 0, sizeof(double) * e.NB)); } };
    al(P); al(G); al(Gold); al(snap.pos); al(snap.vel);
    if (e.ff.pqeq) al(snap.shl);          // PQEq: the shell displacements migrate with their atoms (comm.F90:153,165-167) and are relaxed by every trial solve
    RX_HIP(hipMalloc(reinterpret_cast<void **>(&snap.q), sizeof(double) * e.NB)); RX_HIP(hipMalloc(reinterpret_cast<void **>(&snap.qsfp), sizeof(double) * e.NB));
    RX_HIP(hipMalloc(reinterpret_cast<void **>(&snap.qsfv), sizeof(double) * e.NB)); RX_HIP(hipMalloc(reinterpret_cast<void **>(&snap.type), sizeof(int) * e.NB));
    RX_HIP(hipMalloc(reinterpret_cast<void **>(&snap.gid), sizeof(long long) * e.NB));
  }
  ~Minimiser() {
    auto fr = [](Vec3 &v) { for (int a = 0; a < 3; ++a) if (v.c[a]) (void)hipFree(v.c[a]); };
    fr(P); fr(G); fr(Gold); fr(snap.pos); fr(snap.vel); fr(snap.shl);
    (void)hipFree(snap.q); (void)hipFree(snap.qsfp); (void)hipFree(snap.qsfv); (void)hipFree(snap.type); (void)hipFree(snap.gid);
  }
  void copy3(Vec3 &dst, double *const src[3], int n) { for (int a = 0; a < 3; ++a) RX_HIP(hipMemcpyAsync(dst.c[a], src[a], sizeof(double) * n, hipMemcpyDeviceToDevice, e.stream)); }
  void copy3(double *const dst[3], const Vec3 &src, int n) { for (int a = 0; a < 3; ++a) RX_HIP(hipMemcpyAsync(dst[a], src.c[a], sizeof(double) * n, hipMemcpyDeviceToDevice, e.stream)); }
  void save() {
    const int n = e.N; snap.N = n;
    copy3(snap.pos, e.pos, n); copy3(snap.vel, e.vel, n);
    if (e.ff.pqeq) copy3(snap.shl, e.shl, n);
    RX_HIP(hipMemcpyAsync(snap.q, e.q, sizeof(double) * n, hipMemcpyDeviceToDevice, e.stream));
    RX_HIP(hipMemcpyAsync(snap.qsfp, e.qsfp, sizeof(double) * n, hipMemcpyDeviceToDevice, e.stream));
    RX_HIP(hipMemcpyAsync(snap.qsfv, e.qsfv, sizeof(double) * n, hipMemcpyDeviceToDevice, e.stream));
    RX_HIP(hipMemcpyAsync(snap.type, e.type, sizeof(int) * n, hipMemcpyDeviceToDevice, e.stream));
    RX_HIP(hipMemcpyAsync(snap.gid, e.gid, sizeof(long long) * n, hipMemcpyDeviceToDevice, e.stream));
  }
  void restore() {
    const int n = snap.N;
    copy3(e.pos, snap.pos, n); copy3(e.vel, snap.vel, n);
    if (e.ff.pqeq) copy3(e.shl, snap.shl, n);
    RX_HIP(hipMemcpyAsync(e.q, snap.q, sizeof(double) * n, hipMemcpyDeviceToDevice, e.stream));
    RX_HIP(hipMemcpyAsync(e.qsfp, snap.qsfp, sizeof(double) * n, hipMemcpyDeviceToDevice, e.stream));
    RX_HIP(hipMemcpyAsync(e.qsfv, snap.qsfv, sizeof(double) * n, hipMemcpyDeviceToDevice, e.stream));
    RX_HIP(hipMemcpyAsync(e.type, snap.type, sizeof(int) * n, hipMemcpyDeviceToDevice, e.stream));
    RX_HIP(hipMemcpyAsync(e.gid, snap.gid, sizeof(long long) * n, hipMemcpyDeviceToDevice, e.stream));
    e.N = n; e.G = n; e.st.natoms = n; e.lists_valid = false; e.ghosts_valid = false;
  }
  double allsum(double v) { if (e.nprocs > 1) e.allreduce_host(&v, 1); return v; }
  double dot(const Vec3 &a, const Vec3 &b, int n) {
    const int nb = 240;
    k_dot3<<<nb, 256, 0, e.stream>>>(n, a.c[0], a.c[1], a.c[2], b.c[0], b.c[1], b.c[2], e.partials);
    std::vector<double> h(nb);
    RX_HIP(hipMemcpyAsync(h.data(), e.partials, sizeof(double) * nb, hipMemcpyDeviceToHost, e.stream));
    e.sync_stream();
    double s = 0.0;
    for (double x : h) s += x;
    return allsum(s);
  }
  double energy_here() { e.qeq(); e.force(); ++evals; return allsum(e.pe[0]); }        // PE(0) = sum PE(1:13), all-reduced (cg.F90:50-54)
  // atoms displaced by step * P and migrated; `rider` (if any) travels in the velocity slot and comes back in the new order
  void displace_and_migrate(double step, Vec3 *rider) {
    const int n = e.N;
    if (rider) copy3(e.vel, *rider, n);
    k_axpy3<<<nblk(n, 256), 256, 0, e.stream>>>(n, step, P.c[0], P.c[1], P.c[2], e.pos[0], e.pos[1], e.pos[2]);
    e.migrate();
    if (rider) copy3(*rider, e.vel, e.N);
  }
  // EvaluateEnergyWithStep, cg.F90:358-392: energy at pos + step * P on a copy of the atoms
  double energy_with_step(double step) {
    save();
    displace_and_migrate(step, nullptr);
    const double pe = energy_here();
    restore();
    return pe;
  }
  // BracketSearchRange + the Armijo half of WolfeConditions, cg.F90:101-209
  // e0 = energy of the unmoved atoms, G = their FORCE output (both just evaluated by the caller)
  double bracket(double e0) {
    const double pdf = dot(P, G, e.N);                     // p . f(x), cg.F90:189
    double stepl = 1e-2 / gnatoms;
    for (int it = 0; it < 20; ++it) {                      // CG_MaxBracketLoop
      stepl *= 2.0;
      const double e1 = energy_with_step(stepl);
      if (!(e1 <= e0 + pdf * 1e-4 * stepl)) return stepl;  // Armijo rule violated: the minimum is bracketed (cg.F90:190,127)
    }
    throw EngineError(RXMD_E_STATE, "minimiser: bracket was not found (cg.F90:135-139)");
  }
  // GoldenSectionSearch, cg.F90:244-283; returns the right end of the final interval, which LineMinimization uses as the step
  double golden(double dx) {
    double ax = 0.0;
    const double ratio = 1.0 / 1.61803398875;
    double bx = dx - (dx - ax) * ratio, cx = ax + (dx - ax) * ratio;
    double pb = energy_with_step(bx), pc = energy_with_step(cx);
    for (int it = 0; it < 100; ++it) {                     // CG_MaxLineMinLoop
      if (std::fabs(ax - dx) <= 1e-6 / gnatoms) break;     // CG_GStol
      if (pb < pc) dx = cx; else ax = bx;
      bx = dx - (dx - ax) * ratio; cx = ax + (dx - ax) * ratio;
      pb = energy_with_step(bx); pc = energy_with_step(cx);
    }
    return dx;
  }
};
}  // namespace

// ConjugateGradient, cg.F90:26-98.  Returns the number of CG loops; pe_final = the last total potential energy.
int Engine::minimise(double ftol, int max_loops, double *pe_final, long long *evaluations) {
  if (!atoms_set) throw EngineError(RXMD_E_STATE, "atoms were never set");
  Minimiser m(*this);
  { double n = static_cast<double>(N); m.gnatoms = m.allsum(n); }
  for (int a = 0; a < 3; ++a) RX_HIP(hipMemsetAsync(vel[a], 0, sizeof(double) * NB, stream));   // v(:,:) = 0, cg.F90:39
  double genew = m.energy_here();
  m.copy3(m.G, frc, N); m.copy3(m.P, frc, N);                  // gnew = f ; p = gnew (cg.F90:46-48)
  double stepl = m.bracket(genew);
  int loop = 0;
  for (loop = 0; loop < max_loops; ++loop) {
    // LineMinimization: the step, then the gradient and the search direction follow the atoms through the migration (cg.F90:226-236)
    stepl = m.golden(stepl);
    m.save(); m.displace_and_migrate(stepl, &m.G); m.restore();                 // MigrateVec3D(pos, p, g, stepl)
    m.displace_and_migrate(stepl, &m.P);                                         // pos += stepl * p ; COPYATOMS(MOVE, atype, pos, p)
    for (int a = 0; a < 3; ++a) RX_HIP(hipMemsetAsync(vel[a], 0, sizeof(double) * NB, stream));
    for (int a = 0; a < 3; ++a) RX_HIP(hipMemcpyAsync(m.Gold.c[a], m.G.c[a], sizeof(double) * N, hipMemcpyDeviceToDevice, stream));   // gold = gnew
    const double geold = genew;
    genew = m.energy_here();
    m.copy3(m.G, frc, N);
    if (std::fabs(genew - geold) <= ftol * m.gnatoms) { ++loop; break; }       // cg.F90:75-80
    const double b1 = m.dot(m.Gold, m.Gold, N), b2 = m.dot(m.G, m.G, N), b3 = m.dot(m.G, m.Gold, N);
    k_direction3<<<nblk(N, 256), 256, 0, stream>>>(N, (b2 - b3) / b1, m.G.c[0], m.G.c[1], m.G.c[2], m.P.c[0], m.P.c[1], m.P.c[2]);   // cg.F90:91
    if (loop + 1 >= max_loops) { ++loop; break; }          // out of loops: no trial points behind the last evaluation -- forces, PE(1:13), lists and *pe_final all belong to the final positions
    stepl = m.bracket(genew);
  }
  sync_stream();
  if (pe_final) *pe_final = genew;
  if (evaluations) *evaluations = m.evals;
  return loop;
}

}  // namespace rxmd
