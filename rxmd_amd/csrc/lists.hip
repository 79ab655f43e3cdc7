// lists.hip -- bonded (3 A) neighbour list with reverse index, and the 10 A pair list with the QEq
// "hessian" (shielded-Coulomb matrix) built in the same sweep.
//   NEIGHBORLIST            reference src/main.F90:321-417   -> k_bonded_list + k_bond_csr
//   qeq_initialize          reference src/qeq.F90:183-268    \  one sweep: k_list10
//   GetNonbondingPairList   reference src/main.F90:420-477   /  (the reference walks the stencil twice)
// Candidates come from the engine's own cell grid (cells >= max(rctap/2, maxrc) wide, perpendicular to their faces), sorted by
// (cell x, cell y, z-slice): a (dx,dy) column of the stencil is ONE contiguous, z-ordered run of the sorted arrays.
#include "engine.h"

#include <hipcub/hipcub.hpp>

#include <cstdlib>

namespace rxmd {

static inline int nblk(long long n, int b) { return n > 0 ? static_cast<int>((n + b - 1) / b) : 1; }   // an empty rank still launches (kernels guard their range)


// ---- geometry of a sweep over the engine's grid ---------------------------------------------------------------------------
// The atoms are sorted by (cell x, cell y, z-slice): column (x, y) is one contiguous run, ordered in z to the width of a slice
// (Grid::fz slices per cell).  A sweep around an atom therefore takes from every column of its stencil only the slices that can
// hold a partner: with gap = distance from the atom to the column's footprint in the x-y plane, partners lie within
// dz = sqrt(rc^2 - gap^2) of the atom's z.  For the 10 A list that is 53 % of the atoms of the 5 x 5 x 5 cells (a third of them pass
// the distance test instead of a quarter... of twice as many).  All bounds are conservative (padded by SWEEP_PAD); the exact
// distance test decides.
constexpr double SWEEP_PAD = 1e-6;      // [A] covers the rounding of the normalised <-> real round trip and of the face positions
// real-space gap along axis a between normalised coordinate s (an atom of cell c_self) and cell c2
__device__ inline double axis_gap(const Grid &g, int a, double s, int c_self, int c2) {
  if (c2 == c_self) return 0.0;
  const double face = g.org[a] + static_cast<double>(c2 > c_self ? c2 : c2 + 1) * g.cw[a];   // the face of c2 that looks at the atom
  const double d = (c2 > c_self ? face - s : s - face) * g.wid[a];
  return d > 0.0 ? d : 0.0;
}
// ---- the reference's pair selection in a non-orthogonal box (RefMesh, engine.h) ----
__device__ inline void ref_norm(const RefMesh &m, double x, double y, double z, double (&s)[3]) {     // xu2xs, main.F90:596-616
  s[0] = (m.Hi[0] * x + m.Hi[1] * y + m.Hi[2] * z) - m.obox[0];
  s[1] = (m.Hi[3] * x + m.Hi[4] * y + m.Hi[5] * z) - m.obox[1];
  s[2] = (m.Hi[6] * x + m.Hi[7] * y + m.Hi[8] * z) - m.obox[2];
}
// NEIGHBORLIST visits the 3 x 3 x 3 cells around an atom's cell (main.F90:349-351): cells = floor(rnorm / lcsize), main.F90:305
__device__ inline bool ref_bonded_cells_adjacent(const RefMesh &m, const double (&si)[3], const double (&sj)[3]) {
  for (int a = 0; a < 3; ++a) {
    const int d = static_cast<int>(floor(sj[a] / m.lc[a])) - static_cast<int>(floor(si[a] / m.lc[a]));
    if (d < -1 || d > 1) return false;
  }
  return true;
}
// the non-bonded mesh: cell offset (i,j,k) is visited when the cells' nearest faces, taken as if the axes were orthogonal, are within
// rctap (init.F90:556-592)
__device__ inline bool ref_nb_cells_in_mesh(const RefMesh &m, const double (&si)[3], const double (&sj)[3], double rctap2) {
  double dr2 = 0.0;
  for (int a = 0; a < 3; ++a) {
    int d = static_cast<int>(floor(sj[a] / m.nbl[a])) - static_cast<int>(floor(si[a] / m.nbl[a]));
    d = d > 0 ? d - 1 : (d < 0 ? d + 1 : 0);
    const double rr = d * m.nblr[a];
    dr2 += rr * rr;
  }
  return dr2 <= rctap2;
}

__device__ inline int z_slice(const Grid &g, double sz) {          // the expression of k_cell_ids, monotone in sz
  const int b = static_cast<int>(floor((sz - g.org[2]) * (g.inv[2] * g.fz)));
  return min(max(b, 0), g.nzf - 1);
}
// run [k0, k0 + len) of the sorted arrays that column (x2, y2) contributes to a sweep of radius rcp around (sx, sy, sz); len = 0: none
template <bool ORTHO>
__device__ inline void column_run(const Grid &g, const int *__restrict__ cellstart, double sx, double sy, double sz, int cx, int cy, int x2, int y2,
                                  double rcp, int &k0, int &len) {
  k0 = 0; len = 0;
  if (x2 < 0 || x2 >= g.n[0] || y2 < 0 || y2 >= g.n[1]) return;
  const double gx = axis_gap(g, 0, sx, cx, x2), gy = axis_gap(g, 1, sy, cy, y2);
  // orthogonal axes: the three gaps add in quadrature; skewed axes: each perpendicular gap alone is a lower bound of the distance
  const double d2 = ORTHO ? gx * gx + gy * gy : fmax(gx, gy) * fmax(gx, gy);
  if (d2 > rcp * rcp) return;
  const double dzs = (ORTHO ? sqrt(rcp * rcp - d2) : rcp) * g.iwz;
  const int lo = z_slice(g, sz - dzs), hi = z_slice(g, sz + dzs);
  const int cbf = (x2 * g.n[1] + y2) * g.nzf;
  k0 = cellstart[cbf + lo];
  len = cellstart[cbf + hi + 1] - k0;
}

// Round 5.  Counters of the round-4 form (profiles/r05_a_sq_*): 81 % of the wave cycles parked -- every thread walked its ~135 candidates one
// dependent 32-byte gather after the other -- and the slot-major staging made the mirror search of the packing pass touch one cache line per slot of
// the partner (0.34 ms).  Now four candidates are in flight per thread (tested together, appended in order: same lists), and the lists are staged
// ATOM-major: 32 ints = one 128-byte line per atom, so the packing pass finds the mirror slot in ONE line of the partner.
// (Measured and dropped, NOTES.md round 5: a workgroup per 128 cell-sorted atoms of one cell column with the union of their candidate runs staged in
// LDS -- 43 KB per workgroup leave two wavefronts per SIMD, and 1.05 ms against 0.57 for this kernel's predecessor: the gathers were never the
// cost, the serial dependence was, and 32 wavefronts per CU hide it better than LDS does at 6.)
// ORTHO = false: the instance for skewed boxes carries the reference's cell-mesh tests (RefMesh); the orthogonal one does not pay for them.
constexpr int BL_STRIDE = 32;          // ints per atom of the staging array (MAXNB <= 31)
template <bool ORTHO>
__global__ void __launch_bounds__(256) k_bonded_list(int G, int MAXNB, Grid g, RefMesh rm, DevFF ff, const int *__restrict__ cellid, const int *__restrict__ cellstart,
                                                      const double4 *__restrict__ sorted, const double *__restrict__ x, const double *__restrict__ y,
                                                      const double *__restrict__ z, const double *__restrict__ sx, const double *__restrict__ sy, const double *__restrict__ sz,
                                                      const int *__restrict__ type, int *__restrict__ nbr, int *__restrict__ nbrcnt, int *err) {   // nbr: the atom-major staging array (nbr_sm)
  // squared bond cut-off of every type pair in LDS (0 = the pair has no bond row): one LDS read per candidate instead of two
  // dependent global look-ups (inxn2, then bond[inxn].rc2); and per type the largest cut-off it has with any partner
  __shared__ double s_rc2[256], s_rmax[16];
  for (int t = threadIdx.x; t < ff.n1 * ff.n1 && t < 256; t += blockDim.x) { const int ix = ff.inxn2[t]; s_rc2[t] = ix ? ff.bond[ix].rc2 : 0.0; }
  __syncthreads();
  if (threadIdx.x < ff.n1 && threadIdx.x < 16) {
    double m = 0.0;
    for (int t = 0; t < ff.n1; ++t) m = fmax(m, s_rc2[threadIdx.x * ff.n1 + t]);
    s_rmax[threadIdx.x] = sqrt(m) + SWEEP_PAD;
  }
  __syncthreads();
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x;
  if (i >= G) return;
  const int c = cellid[i];
  const int cy = (c / g.nzf) % g.n[1], cx = c / (g.nzf * g.n[1]);
  const double xi = x[i], yi = y[i], zi = z[i];
  const double sxi = sx[i], syi = sy[i], szi = sz[i];
  const int ti = type[i];
  const double *rc2row = s_rc2 + ti * ff.n1;
  const double rcp = s_rmax[ti];
  double si[3] = {0.0, 0.0, 0.0};
  if (!ORTHO) ref_norm(rm, xi, yi, zi, si);
  int *mine = nbr + static_cast<size_t>(i) * BL_STRIDE;
  int cnt = 0;
  for (int dx = -1; dx <= 1; ++dx) {
    for (int dy = -1; dy <= 1; ++dy) {
      int k0, len;
      column_run<ORTHO>(g, cellstart, sxi, syi, szi, cx, cy, cx + dx, cy + dy, rcp, k0, len);
      const int kend = k0 + len;
      for (int kk0 = k0; kk0 < kend; kk0 += 4) {                             // four candidates in flight, appended in order
        double4 p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u] = sorted[min(kk0 + u, kend - 1)];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const long long w = __double_as_longlong(p[u].w);
          const int j = static_cast<int>(w & 0xffffffffLL), tj = static_cast<int>(w >> 32);
          const double d0 = p[u].x - xi, d1 = p[u].y - yi, d2 = p[u].z - zi;
          const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
          bool in = (kk0 + u < kend) && (j != i) && (r2 < rc2row[tj]);         // dr2 < rc2(inxn), main.F90:366 (no bond row: cut-off 0)
          if (!ORTHO && in) { double sj[3]; ref_norm(rm, p[u].x, p[u].y, p[u].z, sj); in = ref_bonded_cells_adjacent(rm, si, sj); }
          if (in) {
            if (cnt < MAXNB) mine[cnt] = j;
            ++cnt;
          }
        }
      }
    }
  }
  if (cnt > MAXNB) { atomicMax(&err[1], cnt); atomicCAS(&err[0], DERR_NONE, DERR_MAXNB); cnt = MAXNB; }  // main.F90:402-407
  nbrcnt[i] = cnt;
  // err[2] = the longest list of this build if any is longer than 8 (the torsion kernel packs eight atoms into a wavefront when none is longer than 15,
  // and sizes the rows of its k-l delivery table by it)
  if (cnt > 8 && __hip_atomic_load(&err[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cnt) atomicMax(&err[2], cnt);
}

// The sweep above leaves the lists in an atom-major staging array (slot s of atom i at i * 32 + s: a thread appends without knowing the
// totals).  This pass packs them: bond o = boff[i] + s (boff = exclusive prefix of the counts) carries its partner nbr[o], its owner bown[o] = i
// and its MIRROR brev[o] = boff[j] + j1 with nbrlist(j, j1) == i -- the reference's nbrindx (main.F90:383-399) as a direct index into the
// compact tables.  Every per-bond array of the engine is indexed by o: 5.3 entries per RDX atom instead of a 30-slot stride.
__global__ void k_bond_csr(int G, int nres, int NB, long long bcap, const int *__restrict__ nbr_sm, const int *__restrict__ nbrcnt, const int *__restrict__ boff,
                           int *__restrict__ nbr, int *__restrict__ brev, int *__restrict__ bown, const int *__restrict__ type, unsigned char *__restrict__ btype, int *err) {
  // one thread per (atom, slot): eight neighbouring lanes take eight consecutive slots of ONE atom (its staging line is read as a 32-byte piece, not by 8 workgroups),
  // a workgroup 32 consecutive atoms; blockIdx.y = the block of eight slots -- the workgroups of slots no atom of theirs uses leave after one read of the counts
  const int i = xcd_swizzle(blockIdx.x, gridDim.x) * (blockDim.x >> 3) + (threadIdx.x >> 3), s = 8 * blockIdx.y + (threadIdx.x & 7);
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) { err[7] = boff[G]; err[9] = boff[nres]; }   // bonds of this build and the residents' share: the host reads them with the error word it waits for anyway (until round 6: two 5 us copy launches)
  if (i >= G) return;
  const int ni = nbrcnt[i];
  if (s >= ni) return;
  const int ob = boff[i];
  if (static_cast<long long>(ob) + ni > bcap) return;             // the tables are too small for this build: the host sees boff[G] and grows them
  const int j = nbr_sm[static_cast<size_t>(i) * BL_STRIDE + s];
  const int nj = nbrcnt[j];
  const int4 *lj = reinterpret_cast<const int4 *>(nbr_sm + static_cast<size_t>(j) * BL_STRIDE);     // the partner's whole list: one 128-byte line, eight independent loads
  int4 v[BL_STRIDE / 4];
#pragma unroll
  for (int q = 0; q < BL_STRIDE / 4; ++q) v[q] = (4 * q < nj) ? lj[q] : make_int4(-1, -1, -1, -1);
  int found = -1;
#pragma unroll
  for (int q = 0; q < BL_STRIDE / 4; ++q) {
    if (4 * q + 0 < nj && v[q].x == i) found = 4 * q + 0;
    if (4 * q + 1 < nj && v[q].y == i) found = 4 * q + 1;
    if (4 * q + 2 < nj && v[q].z == i) found = 4 * q + 2;
    if (4 * q + 3 < nj && v[q].w == i) found = 4 * q + 3;
  }
  if (found < 0) { atomicCAS(&err[0], DERR_NONE, DERR_NBRINDX); found = 0; }
  nbr[ob + s] = j; brev[ob + s] = boff[j] + found; bown[ob + s] = i; btype[ob + s] = static_cast<unsigned char>(type[j]);   // the partner's type rides with the bond: one dependent gather less where a kernel walks another atom's list
}

// get_coulomb_and_dcoulomb_pqeq (reference src/module.F90:401-418): energy kernel and (1/r) dE/dr at squared distance r2.
// Beyond the taper cutoff the reference returns without touching its outputs (callers then see the previous pair's
// values); here such a lookup contributes nothing -- see DESIGN.md "PQEq beyond-cutoff lookups".
__device__ inline bool pq_lookup(const DevFF &ff, const double4 *__restrict__ tab, int row, double r2, double &E, double &F) {
  if (r2 > ff.rctap2) { E = 0.0; F = 0.0; return false; }
  const int itb = static_cast<int>(r2 * ff.UDRi);
  double t = r2 - itb * ff.UDR;
  t = t * ff.UDRi;
  const double4 nd = tab[static_cast<size_t>(row) * (NTABLE + 2) + itb];
  E = nd.x + t * nd.y; F = nd.z + t * nd.w;
  return true;
}
__device__ inline double wave_sum_l(double v) { return wave_sum64(v); }   // DPP reduction, engine.h

// One wavefront per resident row, two phases.
// Phase 1 (sparse): the candidates of the row = the z-trimmed runs of its 25 stencil columns (column_run above), laid end to end and
// cut into chunks of 64 -- every lane finds its column by a 5-step search of the run offsets in LDS -- four chunks loaded at once;
// the reference's test  dr2 <= rctap2  (main.F90:458); survivors are compacted with a ballot into an LDS queue.  (A first test on an
// FP32 copy of the positions, with only the candidates inside the rounding band of the cutoff decided in FP64, was measured slower:
// 4.62 against 4.43 ms -- the sweep is not bound by the arithmetic of this test.)
// Phase 2 (dense): 64 queued survivors at a time -> exact FP64 distance, table interpolation, list entry, hessian value, row sums;
// a row is written as contiguous runs (coalesced 8-byte + 4-byte streams) in a deterministic order.
// A list entry names the partner by its CELL-SORTED position (the loop variable of this sweep), not by atom index: the consumers
// (QEq matrix passes, ENbond, Ehb) gather from cell-sorted copies, so the 64 lanes of a wavefront hit a handful of cache lines
// instead of 64 scattered ones.  Bits: see NB10_* in engine.h.
// PQ: PQEq variant of qeq_initialize (pqeq.F90:262-353): core-core hessian from the pcc table, the shell-core matrix hsc of
// get_hsh's Csicj term, and per row (fpqeq Eq. 30, sum_j H Z_j, sum_j hsc Z_j, shell-shell energy) -> pqrow
// Round 4: the sweep also writes the WINDOW FORM of the matrix (engine.h WIN_*; until then a second kernel, k_win_build, re-read the entries the
// sweep had just written: 0.8 ms and 2.5 GB per build).  A window group = WIN_ROWS residents of ONE cell column (x, y) of the grid that are
// neighbours in cell-sorted order (rows_sorted, build_windows): all rows of a group have the SAME 25 stencil columns, and the window of the group
// is, per stencil column, the union of the rows' CANDIDATE runs -- [smallest first position, largest end), known from the per-row set-up before
// any distance test -- rounded to units of WIN_UNIT positions: ~2,000 slots (the exact marking of k_win_build came to ~1,500: it counted only
// units with an accepted position).  k_win_columns (one workgroup per group, two rows per wavefront) does that set-up once: it leaves every row's
// 25 runs for the sweep (rowcols), the group's unit list (win_k, win_cnt) and per stencil column what an entry's slot needs (grp_base = first
// position of the interval - 8 x its first unit): slot = position - grp_base | ghost bit.  The sweep itself stays four rows per workgroup
// with no barrier: as ONE kernel of 16-wavefront workgroups it lost 0.8 ms to wavefront slots that idle until the last row of a group is done.
template <bool ORTHO>
__global__ void __launch_bounds__(32 * WIN_ROWS) k_win_columns(int N, Grid g, const int *__restrict__ cellid, const int *__restrict__ cellstart,
                                                               const double *__restrict__ spx, const double *__restrict__ spy, const double *__restrict__ spz, double rcp,
                                                               const int *__restrict__ rows_sorted, int *__restrict__ rowcols, int *__restrict__ grp_base,
                                                               int *__restrict__ win_k, int *__restrict__ win_cnt, int *err) {
  __shared__ int t_lo[32], t_hi[32];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int grp = xcd_swizzle(blockIdx.x, gridDim.x);
  if (grp >= err[8]) return;                  // the launch covers the host's bound of the group count; err[8] = the groups of this build (whole workgroup leaves)
  if (threadIdx.x < 32) { t_lo[threadIdx.x] = 0x7fffffff; t_hi[threadIdx.x] = -1; }
  __syncthreads();
  const int l5 = lane & 31;                   // two rows per wavefront (25 of 32 lanes each: a row has 25 stencil columns)
  const int ridx = grp * WIN_ROWS + 2 * w + (lane >> 5);
  const int i = rows_sorted[ridx];
  if (i < N) {                                // (the last group of a cell column may be short)
    const int c = cellid[i];
    const int cy = (c / g.nzf) % g.n[1], cx = c / (g.nzf * g.n[1]);
    int k0 = 0, len = 0;
    if (l5 < 25) column_run<ORTHO>(g, cellstart, spx[i], spy[i], spz[i], cx, cy, cx + l5 / 5 - 2, cy + l5 % 5 - 2, rcp, k0, len);
    if (len > 0) { atomicMin(&t_lo[l5], k0); atomicMax(&t_hi[l5], k0 + len); }
    __builtin_nontemporal_store(k0, rowcols + static_cast<size_t>(ridx) * 64 + l5); __builtin_nontemporal_store(len, rowcols + static_cast<size_t>(ridx) * 64 + 32 + l5);   // (read once, by the sweep)
  }
  __syncthreads();
  if (w != 0) return;
  int lo8 = 0, nu = 0;                         // units per stencil column and their exclusive prefix (lanes 0..24)
  if (lane < 25 && t_hi[lane] >= 0) { lo8 = t_lo[lane] & ~(WIN_UNIT - 1); nu = (t_hi[lane] - lo8 + WIN_UNIT - 1) / WIN_UNIT; }
  int inc = nu;
#pragma unroll
  for (int o = 1; o < 32; o <<= 1) { const int t2 = __shfl_up(inc, o, 64); if (lane >= o) inc += t2; }
  const int ub = inc - nu, nunits = __shfl(inc, 31, 64);
  if (lane < 32) grp_base[static_cast<size_t>(grp) * 32 + lane] = lo8 - WIN_UNIT * ub;
  const bool fail = nunits > WIN_MAXUNITS;
  if (lane == 0) {
    win_cnt[grp] = fail ? 0 : nunits;
    if (fail) atomicExch(&err[6], 1);
    if (__hip_atomic_load(&err[5], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nunits) atomicMax(&err[5], nunits);
  }
  if (!fail && lane < 25)
    for (int u = 0; u < nu; ++u) win_k[static_cast<size_t>(grp) * WIN_MAXUNITS + ub + u] = lo8 + WIN_UNIT * u;
}

constexpr int L10_ROWS = 8;      // rows (wavefronts) of a workgroup of the 10 A sweep: half a window group
constexpr int L10_CAP = 1600;    // staged candidate positions of a workgroup (28 B each: 44.8 KB + 8 KB of queues and tables = three workgroups per CU; RDX: ~1,500 per 8 rows)
template <bool SELFCHECK, bool PQ, bool ORTHO>
__global__ void __launch_bounds__(64 * L10_ROWS) k_list10(int N, int S10, Grid g, RefMesh rm, DevFF ff, const int *__restrict__ cellid, const int *__restrict__ cellstart,
                                                 const double4 *__restrict__ sorted,
                                                 const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                                                 const double *__restrict__ spx, const double *__restrict__ spy, const double *__restrict__ spz,
                                                 const int *__restrict__ type, const long long *__restrict__ gid,
                                                 int *__restrict__ nb10, double *__restrict__ hess, int *__restrict__ n10, int *err,
                                                 const double4 *__restrict__ sorted_shl, const double *__restrict__ shx, const double *__restrict__ shy, const double *__restrict__ shz,
                                                 double *__restrict__ hsc, double4 *__restrict__ pqrow,
                                                 const double2 *__restrict__ xs0, double2 *__restrict__ s_all, double2 *__restrict__ s_gh, int *__restrict__ rowflag,
                                                 const int *__restrict__ rows_sorted, unsigned short *__restrict__ sl10, const int *__restrict__ rowcols, const int *__restrict__ grp_base, int *__restrict__ gflag) {
  __shared__ int s_q[L10_ROWS][128];     // accepted candidates: (stencil column of the row << 20 | candidate number in the row): position, atom, type and distance come back from the staged copy
  __shared__ int s_K[L10_ROWS][32], s_E[L10_ROWS][32], s_L[L10_ROWS][32];  // per stencil column of a row: first sorted position of its run - candidates before it / slot base of the column in the group's window / the same as s_K for the staged copy
  __shared__ int s_ix2[256];             // inxn2 row of the row's type would do; the whole (n1 x n1) table is 64-256 words
  // Round 5: the candidates of the workgroup's rows, staged ONCE.  The L10_ROWS rows of a workgroup are neighbours in cell-sorted order inside one
  // window group: per stencil column their candidate runs overlap almost entirely.  Until round 4 every row gathered its ~1,300 candidates (32 B
  // each) from L2 on its own -- 41 of the kernel's 82 GB of L1 <- L2 traffic, the wavefronts parked 62 % of their cycles -- now the union of the
  // runs (per column [smallest first position, largest end), LDS min / max over the rows) is copied to LDS with coalesced loads and both phases
  // read it from there.  A union that does not fit (a denser system) leaves the workgroup on the gather path.
  __shared__ double s_x[L10_CAP], s_y[L10_CAP], s_z[L10_CAP];
  __shared__ int s_j[L10_CAP];           // atom index | type << 26 (indices and cell-sorted positions stay below 2^26, NB10_IDX_BITS)
  __shared__ int t_lo[32], t_hi[32], t_off[33];
  const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));   // wave-uniform -> the row's constants live in scalar registers
  for (int t = threadIdx.x; t < ff.n1 * ff.n1 && t < 256; t += blockDim.x) s_ix2[t] = ff.inxn2[t];
  if (threadIdx.x < 32) { t_lo[threadIdx.x] = 0x7fffffff; t_hi[threadIdx.x] = -1; }
  __syncthreads();
  // rows in the order of the window groups (cell-sorted: the rows of a workgroup test nearly the same candidates); the launch covers the
  // host's bound of the group count, err[8] = the groups of this build
  const int ridx = xcd_swizzle(blockIdx.x, gridDim.x) * L10_ROWS + w;
  if (ridx - w >= err[8] * WIN_ROWS) return;  // (the whole workgroup: WIN_ROWS is a multiple of L10_ROWS)
  const int i_raw = rows_sorted[ridx];
  const bool rowlive = i_raw < N;             // false: unused row of a cell column's last group -- the wavefront stays for the barriers and the staging
  const int i = rowlive ? i_raw : 0;
  const int grp = ridx / WIN_ROWS;
  constexpr bool live = true;
  int *sq = s_q[w];
  int *cK = s_K[w], *cE = s_E[w], *cL = s_L[w];
  const double xi = x[i], yi = y[i], zi = z[i];
  const int ti = type[i];
  const size_t row = static_cast<size_t>(i) * S10;
  double sni[3] = {0.0, 0.0, 0.0};
  if (!ORTHO) ref_norm(rm, xi, yi, zi, sni);
  double sxi = 0.0, syi = 0.0, szi = 0.0, Zi = 0.0, p_f = 0.0, p_hz = 0.0, p_bz = 0.0, p_ss = 0.0;
  if (PQ) { sxi = shx[i]; syi = shy[i]; szi = shz[i]; Zi = ff.Zpq[ti]; }
  // xs0 != nullptr: the sweep also forms the row sums H.(qs,qt) of the CG start vector (qt = 0) -- the matrix pass that
  // get_gradient would need before the first iteration (qeq.F90:87) comes for free while the entries are in registers
  double ra = 0.0, rg = 0.0;
  bool anyghost = false;                          // does the row have a ghost partner (boundary row of the domain)?
  int cnt = 0;      // entries written so far
  int qn = 0;       // accepted candidates waiting in the queue
  bool staged = false;   // the workgroup's candidates are in LDS (decided after the set-up below)

  // Phase 2, dense: one queued candidate per lane -> table interpolation, list entry, hessian value.  Only a third of the
  // candidates pass the distance test, so doing this work on compacted batches keeps every lane busy.
  auto emit = [&](int nproc) {
#ifdef RXMD_EXPERIMENTS
    if (g.probe == 2) { cnt += nproc; return; }
#endif
    if (lane < nproc) {
      const int qw = sq[lane], slot = cnt + lane;
      if (slot < S10) {
        const int t_ = qw >> 20, cc_ = qw & 0xfffff;
        const int k = cK[t_] + cc_;                                    // cell-sorted position of the partner
        double px, py, pz; int j, tj;
        if (staged) { const int li = cL[t_] + cc_; px = s_x[li]; py = s_y[li]; pz = s_z[li]; const int jw = s_j[li]; j = jw & static_cast<int>(NB10_IDX_MASK); tj = (jw >> NB10_IDX_BITS) & 31; }
        else { const double4 p = sorted[k]; const long long wv = __double_as_longlong(p.w); px = p.x; py = p.y; pz = p.z; j = static_cast<int>(wv & 0xffffffffLL); tj = static_cast<int>((wv >> 32) & 255); }
        const double d0 = xi - px, d1 = yi - py, d2 = zi - pz;          // the arithmetic of the test, again: the same bits
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        // hessian entry as qeq_initialize computes it: r^2 rounded to REAL(4) first (qeq.F90:191,222-240)
        const float r2f = static_cast<float>(r2);
        double h = 0.0, hc = 0.0;
        const int inxn = s_ix2[ti * ff.n1 + tj];
        // skewed box: a ghost partner beyond the reference's QEq ghost shell is in its FORCE list but not in its QEq matrix
        bool inq = true;
        if (!ORTHO && j >= N) {
          const double g0 = spx[j], g1 = spy[j], g2 = spz[j];
          inq = g0 > rm.qlo[0] && g0 <= rm.qhi[0] && g1 > rm.qlo[1] && g1 <= rm.qhi[1] && g2 > rm.qlo[2] && g2 <= rm.qhi[2];
        }
        if (PQ) {
          if (inq && static_cast<double>(r2f) < ff.rctap2) {                  // the pair is in PQEq's own list (real(4) test, pqeq.F90:305)
            const double C0q = 14.4;                                   // Cclmb0_qeq, module.F90:682
            const double4 sj = sorted_shl[k];
            const double Zj = ff.Zpq[tj];
            const int prow = ff.inxnpq[ti * ff.npq1 + tj];
            double E, F;
            pq_lookup(ff, ff.tabPcc, prow, r2, E, F);                  // core(i)-core(j)
            h = C0q * E;
            p_hz += h * Zj;
            // Eq. 30: field of core(j) minus field of shell(j) at core(i); table row (jty,ity), pqeq.F90:328-334
            double e0 = d0 - sj.x, e1 = d1 - sj.y, e2 = d2 - sj.z;
            pq_lookup(ff, ff.tabPsc, prow, e0 * e0 + e1 * e1 + e2 * e2, E, F);
            p_f += h * Zj - C0q * E * Zj;
            // shell(i)-core(j): Csicj = -hsc * (q_j + Z_j), pqeq.F90:392-395
            e0 = d0 + sxi; e1 = d1 + syi; e2 = d2 + szi;
            pq_lookup(ff, ff.tabPsc, prow, e0 * e0 + e1 * e1 + e2 * e2, E, F);
            hc = C0q * E * Zi;
            p_bz += hc * Zj;
            // shell(i)-shell(j): Csisj, pqeq.F90:397-401 (half of it per row, :409)
            e0 -= sj.x; e1 -= sj.y; e2 -= sj.z;
            pq_lookup(ff, ff.tabPss, prow, e0 * e0 + e1 * e1 + e2 * e2, E, F);
            p_ss += 0.5 * C0q * E * Zi * Zj;
          }
          __builtin_nontemporal_store(hc, hsc + row + slot);
        } else if (inq && static_cast<double>(r2f) < ff.rctap2 && inxn != 0) {
          const int itb = static_cast<int>(static_cast<double>(r2f) * ff.UDRi);
          double drtb = static_cast<double>(r2f) - itb * ff.UDR;
          drtb = drtb * ff.UDRi;
          const double2 T = ff.tabQEq2[static_cast<size_t>(inxn) * (NTABLE + 2) + itb];        // (T[itb], T[itb + 1]) in one 16-byte load
          h = (1.0 - drtb) * T.x + drtb * T.y;
        }
        {   // window slot: the candidate's column (of this row) -> the group's table entry -> first unit of the column + offset inside it
#ifdef RXMD_EXPERIMENTS
          if (!(g.probe & 4))
#endif
          __builtin_nontemporal_store(static_cast<unsigned short>((k - cE[t_]) | (j >= N ? 0x8000 : 0)), sl10 + row + slot);     // cE: first position of the column's interval - 8 x its first unit
        }
        unsigned ent = static_cast<unsigned>(k) | (static_cast<unsigned>(tj) << NB10_IDX_BITS) | (j >= N ? NB10_GHOST : 0u);
        if (SELFCHECK && gid[j] == gid[i]) ent |= NB10_SELF;           // an atom and its own periodic image (small boxes only)
        if (xs0) {
          const double qsj = xs0[k].x;
          ra += h * qsj;
          if (PQ) rg += hc * qsj; else if (j >= N) rg += h * qsj;
        }
        anyghost |= (j >= N);
#ifdef RXMD_EXPERIMENTS
        if (!(g.probe & 8))
#endif
        __builtin_nontemporal_store(static_cast<int>(ent), nb10 + row + slot);      // the three streams are written once and read by other kernels: past the L2, which holds the table nodes
#ifdef RXMD_EXPERIMENTS
        if (!(g.probe & 16))
#endif
        __builtin_nontemporal_store(h, hess + row + slot);
      }
    }
    cnt += nproc;
  };

  // the 25 stencil columns: lane t < 25 owns column t; an inclusive scan over the lanes lays the runs end to end
  int L, myP;
  {   // the row's 25 candidate runs and the slot bases of its group's stencil columns, as k_win_columns left them
    int k0 = 0, len = 0, gb = 0;
    if (lane < 32) { k0 = rowcols[static_cast<size_t>(ridx) * 64 + lane]; len = rowcols[static_cast<size_t>(ridx) * 64 + 32 + lane]; gb = grp_base[static_cast<size_t>(grp) * 32 + lane]; }
    int lpre = len;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) {
      const int l2 = __shfl_up(lpre, o, 64);
      if (lane >= o) lpre += l2;
    }
    if (!rowlive) { len = 0; lpre = 0; }
    if (lane < 32) { cK[lane] = k0 - (lpre - len); cE[lane] = gb; if (len > 0) { atomicMin(&t_lo[lane], k0); atomicMax(&t_hi[lane], k0 + len); } }
    myP = (lane < 32) ? lpre - len : 0x7fffffff;   // candidates before column `lane` (columns 25..31: L)
    L = __shfl(lpre, 31, 64);
  }
  __syncthreads();
  if (w == 0) {                                    // the union of the workgroup's runs, column by column, laid end to end in the staged copy
    const int nu = (lane < 32 && t_hi[lane] >= 0) ? t_hi[lane] - t_lo[lane] : 0;
    int inc = nu;
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) { const int t2 = __shfl_up(inc, o, 64); if (lane >= o) inc += t2; }
    if (lane < 32) t_off[lane] = inc - nu;
    if (lane == 31) t_off[32] = inc;
  }
  __syncthreads();
  staged = t_off[32] <= L10_CAP;                   // (uniform over the workgroup)
  if (staged) {
    for (int c = w; c < 25; c += L10_ROWS) {       // a wavefront copies whole columns: consecutive lanes, consecutive positions
      const int lo = t_lo[c], n_ = t_hi[c] >= 0 ? t_hi[c] - lo : 0, o_ = t_off[c];
      for (int p_ = lane; p_ < n_; p_ += 64) {
        const double4 pp = sorted[lo + p_];
        const long long wv = __double_as_longlong(pp.w);
        s_x[o_ + p_] = pp.x; s_y[o_ + p_] = pp.y; s_z[o_ + p_] = pp.z; s_j[o_ + p_] = static_cast<int>(wv & 0xffffffffLL) | (static_cast<int>((wv >> 32) & 31) << NB10_IDX_BITS);
      }
    }
    if (lane < 32) cL[lane] = cK[lane] - t_lo[lane] + t_off[lane];
  }
  __syncthreads();
  if (!rowlive) return;                            // (no barrier below)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#ifdef RXMD_EXPERIMENTS
  if (g.probe == 1) { if (lane == 0) n10[i] = L; return; }
#endif
  // The column of a candidate = the last one that starts at or before it.  Candidates are walked in order, so the wavefront keeps the next column
  // boundary it has not passed (tn, a scalar; the boundaries sit one per lane in myP and are read with v_readlane): a batch of 64 candidates
  // starts in column tn - 1 and a lane adds one for every boundary of the batch at or below its candidate -- 1.2 boundaries per batch on
  // average, where a binary search over the 32 entries in LDS took five dependent reads per candidate.
  int tn = 1;
  for (int c0 = 0; c0 < L; c0 += 256) {
    // Phase 1, sparse: distance test of 4 x 64 candidates (all loads first), survivors appended to the queue in candidate order
    int kk[4], tcol[4], pj[4];
    bool ok[4];
    double pdx[4], pdy[4], pdz[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int cc = c0 + 64 * u + lane;
      ok[u] = cc < L;
      int t = tn - 1;                            // the column of candidate cc
      for (;;) {
        const int pm = __builtin_amdgcn_readlane(myP, tn);
        if (pm > c0 + 64 * u + 63) break;        // (lanes 32.. hold INT_MAX: tn stops at 32)
        t += (pm <= cc) ? 1 : 0;
        ++tn;
      }
      kk[u] = ok[u] ? cc : 0;                      // candidate number in the row (the position is cK[column] + it)
      tcol[u] = t;
      pdx[u] = 0.0; pdy[u] = 0.0; pdz[u] = 0.0; pj[u] = 0;
      if (ok[u]) {
        if (staged) { const int li = cL[t] + cc; pdx[u] = s_x[li]; pdy[u] = s_y[li]; pdz[u] = s_z[li]; pj[u] = s_j[li] & static_cast<int>(NB10_IDX_MASK); }
        else { const double4 p = sorted[cK[t] + cc]; pdx[u] = p.x; pdy[u] = p.y; pdz[u] = p.z; pj[u] = static_cast<int>(__double_as_longlong(p.w) & 0xffffffffLL); }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      bool in = false;
      double r2q = 0.0;
      if (ok[u]) {
        const int j = pj[u];
        const double d0 = xi - pdx[u], d1 = yi - pdy[u], d2 = zi - pdz[u];
        const double r2 = d0 * d0 + d1 * d1 + d2 * d2;
        r2q = r2;
        in = (j != i) && (r2 <= ff.rctap2);     // dr2 <= rctap2, main.F90:458
        if (!ORTHO && in) { double sj[3]; ref_norm(rm, pdx[u], pdy[u], pdz[u], sj); in = ref_nb_cells_in_mesh(rm, sni, sj, ff.rctap2); }
      }
      (void)r2q;
      const unsigned long long m = __ballot(in);
      if (in) { const int qp = qn + __popcll(m & ((1ULL << lane) - 1ULL)); sq[qp] = kk[u] | (tcol[u] << 20); }
      qn += __popcll(m);
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      if (qn >= 64) {
        emit(64);
        const int rest = qn - 64;
        int v = 0;
        if (lane < rest) v = sq[64 + lane];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (lane < rest) sq[lane] = v;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        qn = rest;
      }
    }
  }
  if (qn > 0) emit(qn);
  if (live) {
  if (cnt > S10) { if (lane == 0) { atomicMax(&err[1], cnt); atomicCAS(&err[0], DERR_NONE, DERR_MAXN10); } cnt = S10; }  // qeq.F90:248-252
  if (lane < 4 && cnt + lane < ((cnt + 3) & ~3)) { nb10[row + cnt + lane] = 0; hess[row + cnt + lane] = 0.0; sl10[row + cnt + lane] = 0; }   // zero-pad the row to a multiple of 4 (value 0, slot 0)
  if (xs0) {
    ra = wave_sum_l(ra); rg = wave_sum_l(rg);
    if (lane == 0) { s_all[i] = make_double2(ra, 0.0); s_gh[i] = make_double2(rg, 0.0); }
  }
  if (PQ) {
    p_f = wave_sum_l(p_f); p_hz = wave_sum_l(p_hz); p_bz = wave_sum_l(p_bz); p_ss = wave_sum_l(p_ss);
    if (lane < 4 && cnt + lane < ((cnt + 3) & ~3)) hsc[row + cnt + lane] = 0.0;
    if (lane == 0) pqrow[i] = make_double4(p_f, p_hz, p_bz, p_ss);
  }
  const unsigned long long mg = __ballot(anyghost);
  if (lane == 0 && rowflag) rowflag[i] = (mg != 0ULL) ? 1 : 0;
  if (lane == 0) {
    n10[i] = cnt | (mg != 0ULL ? N10_GHOST_ROW : 0);
    // err[3] = the longest 10 A row of this build (the ring matrix pass issues a fixed number of DMA instructions per row and needs the bound;
    // read with the error word the host waits for anyway).  One atomic per new maximum, not per row.
    if (__hip_atomic_load(&err[3], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < cnt) atomicMax(&err[3], cnt);
    if (__hip_atomic_load(&err[4], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > cnt) atomicMin(&err[4], cnt);     // ... and the shortest
    if (gflag && mg != 0ULL) gflag[grp] = 1;      // multi-rank: a group with a row that has a ghost partner waits for the vector halo (every writer writes 1; cleared by build_windows)
  }
  }   // live
}

// boundary rows keep their order, interior rows too: index lists for the two launches of the matrix pass
__global__ void k_split_rows(int N, const int *__restrict__ flag, const int *__restrict__ scan, int *__restrict__ rows_int, int *__restrict__ rows_bnd) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  if (flag[i]) rows_bnd[scan[i]] = i; else rows_int[i - scan[i]] = i;
}

// ---- window form of the 10 A matrix (engine.h: WIN_*) -------------------------------------------------------------------------------
__global__ void k_resident_flags(int G, int N, const int *__restrict__ perm, int *__restrict__ flag, int *__restrict__ rows_sorted, int nrows_fill, int *__restrict__ win_cnt, int *__restrict__ win_flag, int ngroups_fill) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k <= G) flag[k] = (k < G && perm[k] < N) ? 1 : 0;
  // what three memsets did until round 6: every row slot a sentinel >= N until k_rows_sorted names its row, empty windows and "interior" for the groups
  // between this build's count and the launch bound
  for (int t = k; t < nrows_fill; t += gridDim.x * blockDim.x) rows_sorted[t] = 0x7f7f7f7f;
  for (int t = k; t < ngroups_fill; t += gridDim.x * blockDim.x) { win_cnt[t] = 0; win_flag[t] = 0; }
  if (k == 0) win_flag[ngroups_fill] = 0;
}
// Window groups never straddle a cell column (x, y) of the grid: the rows of a group then share their 25 stencil columns, and what each of those
// contributes to the group's window is ONE short z-interval.  (A group that ran over into the next cell column held rows from the top of one
// column and the bottom of the next: the union interval of a shared stencil column covered the whole box height.)  Per column: groups =
// ceil(residents / WIN_ROWS); rows_sorted is laid out by groups, the unused rows of a column's last group hold a sentinel >= N.
__global__ void k_col_groups(int ncol, int nzf, const int *__restrict__ cellstart, const int *__restrict__ rank, int *__restrict__ colg) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c > ncol) return;
  colg[c] = c < ncol ? (rank[cellstart[(c + 1) * nzf]] - rank[cellstart[c * nzf]] + WIN_ROWS - 1) / WIN_ROWS : 0;
}
__global__ void k_rows_sorted(int G, int N, int nzf, const int *__restrict__ perm, const int *__restrict__ rank, const int *__restrict__ cid_sorted, const int *__restrict__ cellstart,
                              const int *__restrict__ colgo, int *__restrict__ rows_sorted, int ncol, int *__restrict__ err) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) err[8] = colgo[ncol];                              // groups of this build: the sweep's workgroups beyond it leave at once; the host reads it with the error word
  if (k >= G || perm[k] >= N) return;
  const int c = cid_sorted[k] / nzf;
  const int r = rank[k] - rank[cellstart[c * nzf]];              // this resident's place among the residents of its column
  rows_sorted[(colgo[c] + r / WIN_ROWS) * WIN_ROWS + (r % WIN_ROWS)] = perm[k];
}
__global__ void k_split_groups(int ng, const int *__restrict__ flag, const int *__restrict__ scan, int *__restrict__ g_int, int *__restrict__ g_bnd) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= ng) return;
  if (flag[g]) g_bnd[scan[g]] = g; else g_int[g - scan[g]] = g;
}

// the residents in cell-sorted order (rows_sorted): the rows of a window group, and the work order of the 10 A sweep that builds the windows
void Engine::build_windows() {
  win_groups = static_cast<int>(win_groups_bound(N));                                                         // until the host has read this build's count: the bound (launch size of the sweep)
  k_resident_flags<<<nblk(G + 1, 256), 256, 0, stream>>>(G, N, perm, flags2, rows_sorted, win_groups * WIN_ROWS, win_cnt, win_flag, win_groups);
  size_t tb = cubtmp_bytes;
  RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags2, scanout2, G + 1, stream));
  const int ncol = grid.n[0] * grid.n[1];
  k_col_groups<<<nblk(ncol + 1, 256), 256, 0, stream>>>(ncol, grid.nzf, cellstart, scanout2, flags);     // (flags / scanout: free until the sweep writes its row flags)
  tb = cubtmp_bytes;
  RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags, scanout, ncol + 1, stream));
  k_rows_sorted<<<nblk(G, 256), 256, 0, stream>>>(G, N, grid.nzf, perm, scanout2, cellid_sorted, cellstart, scanout, rows_sorted, ncol, d_err);
}

// The words a list build starts from, in ONE launch (until round 6: nine 4-byte memsets of ~5 us each, spread over the build): the error word's
// companions -- err[2] longest bonded list, [3] longest / [4] shortest 10 A row, [5] largest window, [6] window overflow -- and the count behind the last
// atom that the prefix sum of the bond counts reads.
__global__ void k_build_prologue(int *__restrict__ err, int *__restrict__ nbrcnt_end, int what) {
  if (threadIdx.x != 0 || blockIdx.x != 0) return;
  if (what & 1) { err[2] = 0; *nbrcnt_end = 0; }
  if (what & 2) { err[3] = 0; err[4] = 0x7f7f7f7f; err[5] = 0; err[6] = 0; }      // 0x7f7f7f7f: larger than any row
}
void Engine::build_prologue(int what) { k_build_prologue<<<1, 64, 0, stream>>>(d_err, nbrcnt + G, what); }

void Engine::build_bonded_list(bool pack_only) {
  // pack_only: the compact tables were too small for this build and have been re-allocated; the staging lines and the counts of the sweep are intact
  if (!pack_only) {
  if (grid.ortho) k_bonded_list<true><<<nblk(G, 256), 256, 0, stream>>>(G, MAXNB, grid, rmesh, dff, cellid, cellstart, sorted_xyzi, pos[0], pos[1], pos[2], spos[0], spos[1], spos[2], type, nbr_sm, nbrcnt, d_err);
  else k_bonded_list<false><<<nblk(G, 256), 256, 0, stream>>>(G, MAXNB, grid, rmesh, dff, cellid, cellstart, sorted_xyzi, pos[0], pos[1], pos[2], spos[0], spos[1], spos[2], type, nbr_sm, nbrcnt, d_err);
  }
  // (nbrcnt[G] = 0 by build_prologue: G < NB always, the scan below runs over G + 1 counts)
  size_t tb = cubtmp_bytes;
  RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, nbrcnt, boff, G + 1, stream));
  k_bond_csr<<<dim3(nblk(G, 32), (MAXNB + 7) / 8), 256, 0, stream>>>(G, N, NB, static_cast<long long>(bcap), nbr_sm, nbrcnt, boff, nbr, brev, bown, type, btype, d_err);
}

constexpr int L10_ROWS_LAUNCH = L10_ROWS;
void Engine::build_list10() {
  grid.probe = static_cast<int>(opt.list_probe);    // (experiments build only: 0 otherwise)
  if (list10_retry) build_prologue(2);                              // (a first build of the step had them from build_prologue(3) at the head of the build)
  // an atom can meet its own image within rctap only if some box edge is shorter than 2*rctap
  const bool selfcheck = (grid.wid[0] < 2.0 * ff.rctap + 1.0) || (grid.wid[1] < 2.0 * ff.rctap + 1.0) || (grid.wid[2] < 2.0 * ff.rctap + 1.0);
  list_selfcheck = selfcheck;
#define RX_LIST10(SC, PQF) do { if (grid.ortho) RX_LIST10_O(SC, PQF, true); else RX_LIST10_O(SC, PQF, false); } while (0)
#define RX_LIST10_O(SC, PQF, OR)                                                                                                               \
  k_list10<SC, PQF, OR><<<std::max(win_groups, 1) * (WIN_ROWS / L10_ROWS_LAUNCH), 64 * L10_ROWS_LAUNCH, 0, stream>>>(N, S10, grid, rmesh, dff, cellid, cellstart, sorted_xyzi, pos[0], pos[1], pos[2], spos[0], spos[1], spos[2], type, gid, \
                                                    nb10, hess, n10, d_err, sorted_shl, shl[0], shl[1], shl[2], hsc, pqrow, sums_from_list ? xs : nullptr, sall, sgh, multi() ? flags : nullptr, \
                                                    rows_sorted, sl10, rowcols, grp_base, gflag)
  win_valid = false;
  build_windows();
  int *gflag = win_flag;
  const bool kt10 = kt_begin(&st.ms_k_list10);
  if (grid.ortho) k_win_columns<true><<<std::max(win_groups, 1), 32 * WIN_ROWS, 0, stream>>>(N, grid, cellid, cellstart, spos[0], spos[1], spos[2], dff.rctap_pad, rows_sorted, rowcols, grp_base, win_k, win_cnt, d_err);
  else k_win_columns<false><<<std::max(win_groups, 1), 32 * WIN_ROWS, 0, stream>>>(N, grid, cellid, cellstart, spos[0], spos[1], spos[2], dff.rctap_pad, rows_sorted, rowcols, grp_base, win_k, win_cnt, d_err);
  if (ff.pqeq) { if (selfcheck) RX_LIST10(true, true); else RX_LIST10(false, true); }
  else { if (selfcheck) RX_LIST10(true, false); else RX_LIST10(false, false); }
  kt_end(kt10);
#undef RX_LIST10
#undef RX_LIST10_O
  if (multi()) {                                     // interior groups (no row with a ghost partner) / boundary groups: the two launches of an overlapped pass
    size_t tb = cubtmp_bytes;
    RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, win_flag, scanout2, win_groups + 1, stream));
    k_split_groups<<<nblk(win_groups, 256), 256, 0, stream>>>(win_groups, win_flag, scanout2, win_gint, win_gbnd);
    RX_HIP(hipMemcpyAsync(&win_nbnd, scanout2 + win_groups, sizeof(int), hipMemcpyDeviceToHost, stream));   // valid after the synchronisation of the list build's error check
  }
  if (multi()) {     // interior rows (no ghost partner) and boundary rows: the matrix pass does the former while the vector halo is in flight
    RX_HIP(hipMemsetAsync(flags + N, 0, sizeof(int), stream));
    size_t tb = cubtmp_bytes;
    RX_HIP(hipcub::DeviceScan::ExclusiveSum(cubtmp, tb, flags, scanout, N + 1, stream));
    k_split_rows<<<nblk(N, 256), 256, 0, stream>>>(N, flags, scanout, rows_int, rows_bnd);
    RX_HIP(hipMemcpyAsync(&n_bnd, scanout + N, sizeof(int), hipMemcpyDeviceToHost, stream));
    rows_split_pending = true;          // n_bnd is valid after the next stream synchronisation (check_device_error of the list build)
  }
}

}  // namespace rxmd
