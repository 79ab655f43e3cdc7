// experiments.hip -- timing probes and stripped-down / variant forms of the QEq matrix pass.  NOT part of the product library: compiled only by
// `make -C rxmd_amd/csrc experiments` (-DRXMD_EXPERIMENTS -> rxmd_amd/librxmd_hip_exp.so), reached through debug taps >= 100 (capi.hip).
// Until round 6 this code sat inside qeq.hip behind #ifdef.
#ifdef RXMD_EXPERIMENTS
#include "spmv_kernels.h"

#include <cstdio>
#include <cstdlib>
#include <vector>

namespace rxmd {

// ---- where does the row kernel's time go?  Stripped-down forms of k_spmv, timed in isolation (debug tap 102; experiments only) -------
//   LEVEL 0: the two streams of a row only (4 x 64 entries per trip as k_spmv), one sum, one wavefront reduction, no store
//   LEVEL 1: + the 16-byte gather per entry and the two FMAs
//   LEVEL 2: + the ghost-column sums and all four reductions
//   LEVEL 3: + the tail operands (type, hst, gst of the row)   LEVEL 4: + the two 16-byte row stores
//   LEVEL 5: + the per-workgroup partials with their barrier (= the work of k_spmv<HSH, STORE>)   LEVEL 6: as 5 with ONE 32-byte row store
template <int LEVEL>
__global__ void __launch_bounds__(1024) k_spmv_bisect(int N, int S10, const int *__restrict__ nb10, const double *__restrict__ hess, const int *__restrict__ n10,
                                                      const double2 *__restrict__ xv, const double2 *__restrict__ hst, const double2 *__restrict__ gst,
                                                      const int *__restrict__ type, double *__restrict__ partials, double2 *__restrict__ rs_all, double2 *__restrict__ rs_gh, double *__restrict__ sink) {
  const int lane = threadIdx.x & 63, wpb = blockDim.x >> 6;
  const int row = xcd_swizzle(blockIdx.x, gridDim.x) * wpb + __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  if (row < N) {
    const size_t base = static_cast<size_t>(row) * S10;
    unsigned e[UNR];
    double h[UNR];
    auto request = [&](int kb, int bound) {
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        const int k = kb + lane + 64 * u;
        const bool ok = k < bound;
        e[u] = ok ? static_cast<unsigned>(__builtin_nontemporal_load(nb10 + base + k)) : 0u;
        h[u] = ok ? __builtin_nontemporal_load(hess + base + k) : 0.0;
      }
    };
    request(0, S10);
    const int n = n10[row] & N10_COUNT;
    int pf_t = 0; double2 pf_a = make_double2(0, 0), pf_b = make_double2(0, 0);
    if (LEVEL >= 3) { pf_t = type[row]; pf_a = hst[row]; pf_b = gst[row]; }
#pragma unroll
    for (int u = 0; u < UNR; ++u) { const bool ok = lane + 64 * u < n; e[u] = ok ? e[u] : 0u; h[u] = ok ? h[u] : 0.0; }
    double as = 0.0, at = 0.0, gs_ = 0.0, gt_ = 0.0;
    for (int kb = 0; kb < n; kb += 64 * UNR) {
      if (kb > 0) request(kb, n);
#pragma unroll
      for (int u = 0; u < UNR; ++u) {
        if (LEVEL == 0) { as += h[u] * static_cast<double>(e[u] & 255u); }
        else {
          const double2 v = xv[e[u] & NB10_IDX_MASK];
          as += h[u] * v.x; at += h[u] * v.y;
          if (LEVEL >= 2) { const double hg = (e[u] & NB10_GHOST) ? h[u] : 0.0; gs_ += hg * v.x; gt_ += hg * v.y; }
        }
      }
    }
    as = wave_sum(as);
    if (LEVEL >= 1) at = wave_sum(at);
    if (LEVEL >= 2) { gs_ = wave_sum(gs_); gt_ = wave_sum(gt_); }
    if (lane == 0) {
      if (LEVEL >= 3) {
        if (LEVEL == 4 || LEVEL == 5) { rs_all[row] = make_double2(as, at); rs_gh[row] = make_double2(gs_, gt_); }
        if (LEVEL == 6) reinterpret_cast<double4 *>(sink)[row + 1] = make_double4(as, at, gs_, gt_);      // (sink: a 32-byte-per-row buffer for this level)
        acc[0] = (0.5 * pf_a.x + as) * pf_a.x; acc[1] = (0.5 * pf_a.y + at) * pf_a.y; acc[2] = pf_b.x * pf_a.x + pf_t; acc[3] = pf_b.y * pf_a.y;
        if (LEVEL == 3 && acc[0] + acc[1] + acc[2] + acc[3] == 1.2345e-300) sink[0] = as;
      } else if (as + at + gs_ + gt_ == 1.2345e-300) sink[0] = as;       // keeps the sums alive
    }
  }
  if (LEVEL >= 5) block_store_partials<4>(acc, partials, 4);
}
void spmv_bisect_ms(Engine &e, double *out4) {
  const int rb = nblk(e.N, 16);
  double *buf32 = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&buf32), sizeof(double) * 4 * (static_cast<size_t>(e.N) + 2)) != hipSuccess) return;
  auto run = [&](auto lv) {
    constexpr int L = decltype(lv)::value;
    for (int r = 0; r < 11; ++r) {
      if (r == 1) hipEventRecord(e.ev[2], e.stream);
      k_spmv_bisect<L><<<rb, 1024, 0, e.stream>>>(e.N, e.S10, e.nb10, e.hess, e.n10, e.xs, e.hst, e.gst, e.type, e.partials, e.wall, e.wgh, buf32);
    }
    hipEventRecord(e.ev[3], e.stream); hipEventSynchronize(e.ev[3]);
    float ms = 0; hipEventElapsedTime(&ms, e.ev[2], e.ev[3]);
    return static_cast<double>(ms) / 10.0;
  };
  out4[0] = run(std::integral_constant<int, 0>{}); out4[1] = run(std::integral_constant<int, 1>{});
  out4[2] = run(std::integral_constant<int, 2>{}); out4[3] = run(std::integral_constant<int, 3>{});
  out4[4] = run(std::integral_constant<int, 4>{}); out4[5] = run(std::integral_constant<int, 5>{}); out4[6] = run(std::integral_constant<int, 6>{});
  (void)hipFree(buf32);
}

__global__ void k_rows_to_rank_order(int N, int S10, const int *__restrict__ rows_sorted, const double *__restrict__ h, const unsigned short *__restrict__ sl, double *__restrict__ h2, unsigned short *__restrict__ s2) {
  const int r = blockIdx.x;                          // destination row = rank
  if (r >= N) return;
  const size_t src = static_cast<size_t>(rows_sorted[r]) * S10, dst = static_cast<size_t>(r) * S10;
  for (int k = threadIdx.x; k < S10; k += blockDim.x) { h2[dst + k] = h[src + k]; s2[dst + k] = sl[src + k]; }
}
// the real window pass and the real row pass back to back, alternating in ONE process (timings repeat to 0.1 % inside a process and differ
// by +-6 % between processes on the same box, so variants are compared here, compiled side by side) (debug tap 104; experiments only)
void spmv_isolated_ms(Engine &e, double *out) {
  for (int k = 0; k < 20; ++k) out[k] = -1.0;
  const int reps = std::max(1, static_cast<int>(e.opt.iso_reps));
  auto timed = [&](auto launch) {
    for (int r = 0; r < reps + 1; ++r) {
      if (r == 1) hipEventRecord(e.ev[2], e.stream);
      launch();
    }
    hipEventRecord(e.ev[3], e.stream); hipEventSynchronize(e.ev[3]);
    float ms = 0; hipEventElapsedTime(&ms, e.ev[2], e.ev[3]);
    return static_cast<double>(ms) / reps;
  };
  if (e.ff.pqeq) return;
  double acc[4] = {0.0, 0.0, 0.0, 0.0};
  const int rounds = 3;                             // the three kernels alternate: one state of the box for all of them
  for (int rd = 0; rd < rounds; ++rd) {
    if (e.win_valid) {
      const size_t lds = static_cast<size_t>(e.win_maxunits) * WIN_UNIT * sizeof(double2);
      acc[0] += timed([&] { k_spmv_win<MODE_HSH, true, false><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, e.hess, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr); });
    }
    acc[1] += timed([&] { k_spmv<MODE_HSH, true, false, 1><<<nblk(e.N, 16), 1024, 0, e.stream>>>(e.N, e.S10, e.dff, e.nb10, e.hess, e.n10, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, 1, nullptr, e.N, 0, nullptr); });
  }
  out[1] = acc[1] / rounds;
  if (e.win_valid) {                               // variant: 384 entries in flight
    const size_t lds = static_cast<size_t>(e.win_maxunits) * WIN_UNIT * sizeof(double2);
    double a3 = 0.0;
    for (int rd = 0; rd < rounds; ++rd)
      a3 += timed([&] { k_spmv_win<MODE_HSH, true, false, 3><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, e.hess, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr); });
    out[2] = a3 / rounds;
    double a4 = 0.0;                                // variant: second batch requested before the barrier (VAR & 2)
    for (int rd = 0; rd < rounds; ++rd)
      a4 += timed([&] { k_spmv_win<MODE_HSH, true, false, 2, 2><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, e.hess, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr); });
    out[3] = a4 / rounds;
    double a5 = 0.0;                                // the default once more, after the variants (drift of the box)
    for (int rd = 0; rd < rounds; ++rd)
      a5 += timed([&] { k_spmv_win<MODE_HSH, true, false><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, e.hess, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr); });
    out[4] = a5 / rounds;
    {   // interior groups without the ghost-column sums (VAR & 4), with and without the prefetch; default after them once more
      double b0 = 0.0, b1 = 0.0, b2 = 0.0;
      for (int rd = 0; rd < rounds; ++rd) {
        b0 += timed([&] { k_spmv_win<MODE_HSH, true, false, 2, 6><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, e.hess, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr, e.win_flag); });
        b1 += timed([&] { k_spmv_win<MODE_HSH, true, false, 2, 2><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, e.hess, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr); });
        b2 += timed([&] { k_spmv_win<MODE_HSH, true, false, 2, 4><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, e.hess, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr, e.win_flag); });
      }
      out[5] = b0 / rounds; out[6] = b1 / rounds; out[7] = b2 / rounds;
    }
  }
  // RXMD_ISO_COPIES=1: does the pass time depend on WHERE its streams lie?  Four copies of the value and slot arrays held at the same time, the
  // pass on each, twice round (out[2..9]): a property of the buffer repeats in the second round, a drift in time does not.  out[10]: the last
  // copy with its rows in cell-sorted order, read without the row indirection.
  if (e.win_valid && e.opt.iso_copies) {
    const size_t ne = static_cast<size_t>(e.rows10) * e.S10;
    const size_t lds = static_cast<size_t>(e.win_maxunits) * WIN_UNIT * sizeof(double2);
    double *h2[4] = {nullptr, nullptr, nullptr, nullptr}; unsigned short *s2[4] = {nullptr, nullptr, nullptr, nullptr};
    bool ok = true;
    for (int c = 0; c < 4 && ok; ++c) {
      if (c >= 2) {                                  // copies C and D: physically contiguous, if the driver grants it
        ok = hipExtMallocWithFlags(reinterpret_cast<void **>(&h2[c]), ne * sizeof(double), hipDeviceMallocContiguous) == hipSuccess &&
             hipExtMallocWithFlags(reinterpret_cast<void **>(&s2[c]), ne * sizeof(unsigned short), hipDeviceMallocContiguous) == hipSuccess;
        if (!ok) { (void)hipGetLastError(); std::fprintf(stderr, "contiguous allocation refused\n"); if (h2[c]) { (void)hipFree(h2[c]); h2[c] = nullptr; } if (s2[c]) { (void)hipFree(s2[c]); s2[c] = nullptr; } }
      }
      if (c < 2 || !ok)
      ok = hipMalloc(reinterpret_cast<void **>(&h2[c]), ne * sizeof(double)) == hipSuccess && hipMalloc(reinterpret_cast<void **>(&s2[c]), ne * sizeof(unsigned short)) == hipSuccess;
      if (ok) { hipMemcpyAsync(h2[c], e.hess, ne * sizeof(double), hipMemcpyDeviceToDevice, e.stream); hipMemcpyAsync(s2[c], e.sl10, ne * sizeof(unsigned short), hipMemcpyDeviceToDevice, e.stream); }
    }
    if (ok) {
      for (int round2 = 0; round2 < 2; ++round2)
        for (int c = 0; c < 4; ++c)
          out[2 + 4 * round2 + c] = timed([&] { k_spmv_win<MODE_HSH, true, false><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, s2[c], h2[c], e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr); });
      k_rows_to_rank_order<<<e.N, 256, 0, e.stream>>>(e.N, e.S10, e.rows_sorted, e.hess, e.sl10, h2[3], s2[3]);
      out[10] = timed([&] { k_spmv_win<MODE_HSH, true, false, 2, 1><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, s2[3], h2[3], e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr); });
      out[11] = timed([&] { k_spmv_win<MODE_HSH, true, false><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, e.hess, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr); });
    } else (void)hipGetLastError();
    for (int c = 0; c < 4; ++c) { if (h2[c]) (void)hipFree(h2[c]); if (s2[c]) (void)hipFree(s2[c]); }
    // ... and on ONE allocation with the value array at different offsets inside it (out[12..19]): a dependence on low address bits would show here
    {
      static const size_t offs[8] = {0, 4096, 65536, size_t(1) << 20, (size_t(2) << 20) + 4096, size_t(16) << 20, (size_t(37) << 20) + 8192, size_t(64) << 20};
      char *big = nullptr;
      if (hipMalloc(reinterpret_cast<void **>(&big), ne * sizeof(double) + (size_t(65) << 20)) == hipSuccess) {
        for (int c = 0; c < 8; ++c) {
          double *hh = reinterpret_cast<double *>(big + offs[c]);
          hipMemcpyAsync(hh, e.hess, ne * sizeof(double), hipMemcpyDeviceToDevice, e.stream);
          out[12 + c] = timed([&] { k_spmv_win<MODE_HSH, true, false><<<e.win_groups, 64 * WIN_ROWS, lds, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, hh, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr); });
        }
        (void)hipFree(big);
      } else (void)hipGetLastError();
    }
  }
  if (e.win_valid) out[0] = acc[0] / rounds;      // out[2], out[3]: variants of the window pass (template parameter VAR) when some are being compared
}


// ---- timing probe: the symmetric matrix read ONCE over 3-D tiles (debug tap 105; experiments only) -----------------------------------
// What a half-storage pass would cost before any of it is built: a workgroup of 16 wavefronts owns a tile of TILE_ROWS rows and holds the vector AND an
// accumulator for TILE_WS window slots in LDS (x 16 B + y 16 B per slot: 147 KB, one workgroup per CU).  Per entry of a HALF row: value (8 B) + slot (2 B)
// streamed as in the real pass, the partner's x from LDS, two FMAs into the row's sums and two LDS atomic adds H_ij x_i into the partner's accumulator;
// at the end the accumulators leave as plain coalesced stores (a second kernel gathers a row's ~23 halo contributions in a fixed order: k_tile_gather_probe).
// The bytes are real (the engine's own value / slot arrays, the first half of every row), the slots are scrambled into the tile's window (the LDS access
// pattern of a 3-D tile: uniformly spread), the arithmetic is what the real pass would do.  Results mean nothing; times do.
constexpr int TILE_ROWS = 416, TILE_WS = 4608;
template <bool TRANSPOSED>     // false: the forward products only (what the streams and the LDS reads cost at one workgroup per CU)
__global__ void __launch_bounds__(1024, 1) k_spmv_tile_probe(int N, int G, int S10, const unsigned short *__restrict__ sl10, const double *__restrict__ hess, const int *__restrict__ n10,
                                                            const double2 *__restrict__ xv, double2 *__restrict__ rs_all, double2 *__restrict__ ybuf) {
  extern __shared__ double2 s_xy[];                 // [0, TILE_WS): x ; [TILE_WS, 2 TILE_WS): y
  double2 *s_x = s_xy, *s_y = s_xy + TILE_WS;
  typedef double d2v __attribute__((ext_vector_type(2)));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x >> 6));
  const int tile = xcd_swizzle(blockIdx.x, gridDim.x);
  const int row0 = tile * TILE_ROWS;
  const size_t wbase = (static_cast<size_t>(tile) * 1531u) % static_cast<size_t>(max(G - TILE_WS, 1));
  for (int t = threadIdx.x; t < TILE_WS; t += 1024) { s_x[t] = xv[wbase + t]; s_y[t] = make_double2(0.0, 0.0); }
  // two rows of a wavefront in flight: the next row's batch is requested before the current row's arithmetic (512 entries per wavefront in flight)
  d2v v[2][2]; unsigned ss[2][2]; int nh[2] = {0, 0};
  auto request = [&](int buf, int row) {
    const bool live = row < N && row < row0 + TILE_ROWS;
    const size_t base = static_cast<size_t>(live ? row : 0) * S10;
    const d2v *hv2 = reinterpret_cast<const d2v *>(hess + base);
    const unsigned *sl2 = reinterpret_cast<const unsigned *>(sl10 + base);
    nh[buf] = live ? ((n10[row] & N10_COUNT) + 1) / 2 : 0;                       // half of the row's entries
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = 128 * u + 2 * lane;
      const bool ok = live && k < S10 / 2;
      v[buf][u] = ok ? __builtin_nontemporal_load(hv2 + (k >> 1)) : d2v{0.0, 0.0};
      ss[buf][u] = ok ? __builtin_nontemporal_load(sl2 + (k >> 1)) : 0u;
    }
  };
  request(0, row0 + wave);
  __syncthreads();
  int cur = 0;
  for (int r = row0 + wave; r < row0 + TILE_ROWS; r += 16) {
    request(cur ^ 1, r + 16);
    const int n = nh[cur];
    const double2 xi = s_x[(r * 11) % TILE_WS];
    double as = 0.0, at = 0.0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int k = 128 * u + 2 * lane;
      const unsigned s0 = ((ss[cur][u] & 0x7fffu) * 13u + lane * 71u) % TILE_WS, s1 = (((ss[cur][u] >> 16) & 0x7fffu) * 13u + lane * 71u + 37u) % TILE_WS;
      const double h0 = k < n ? v[cur][u].x : 0.0, h1 = k + 1 < n ? v[cur][u].y : 0.0;
      const double2 y0 = s_x[s0], y1 = s_x[s1];
      as += h0 * y0.x; at += h0 * y0.y; as += h1 * y1.x; at += h1 * y1.y;
      if (TRANSPOSED && k < n) { __hip_atomic_fetch_add(&s_y[s0].x, h0 * xi.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&s_y[s0].y, h0 * xi.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if (TRANSPOSED && k + 1 < n) { __hip_atomic_fetch_add(&s_y[s1].x, h1 * xi.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); __hip_atomic_fetch_add(&s_y[s1].y, h1 * xi.y, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
    as = wave_sum(as); at = wave_sum(at);
    if (lane == 0 && r < N) rs_all[r] = make_double2(as, at);
    cur ^= 1;
  }
  __syncthreads();
  double2 *yb = ybuf + static_cast<size_t>(blockIdx.x) * TILE_WS;
  for (int t = threadIdx.x; t < TILE_WS; t += 1024) __builtin_nontemporal_store(s_y[t].x, &yb[t].x), __builtin_nontemporal_store(s_y[t].y, &yb[t].y);
}
// the second kernel of that scheme: a row adds the contributions its slot received in the ~23 tiles whose window holds it, in a fixed order
__global__ void __launch_bounds__(256) k_tile_gather_probe(int N, int ntiles, const double2 *__restrict__ ybuf, double2 *__restrict__ rs_all) {
  const int r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= N) return;
  const int tile = r / TILE_ROWS;
  double2 a = rs_all[r];
  for (int c = 0; c < 23; ++c) {
    const int t2 = (tile + (c % 3 - 1) + 3 * ((c / 3) % 3 - 1) * 7 + 9 * (c / 9 - 1) * 41 + ntiles) % ntiles;      // neighbours in a 3-D arrangement of the tiles
    const double2 y = ybuf[static_cast<size_t>(t2) * TILE_WS + (static_cast<unsigned>(r) * 29u + c * 613u) % TILE_WS];
    a.x += y.x; a.y += y.y;
  }
  rs_all[r] = a;
}
void spmv_tile_probe_ms(Engine &e, double *out) {
  for (int k = 0; k < 5; ++k) out[k] = -1.0;
  if (e.ff.pqeq || e.G < TILE_WS + 16) return;
  const int ntiles = (e.N + TILE_ROWS - 1) / TILE_ROWS;
  double2 *ybuf = nullptr;
  if (hipMalloc(reinterpret_cast<void **>(&ybuf), sizeof(double2) * static_cast<size_t>(ntiles) * TILE_WS) != hipSuccess) { (void)hipGetLastError(); return; }
  const size_t lds = sizeof(double2) * 2 * TILE_WS;
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_spmv_tile_probe<true>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) { (void)hipGetLastError(); }
  if (hipFuncSetAttribute(reinterpret_cast<const void *>(k_spmv_tile_probe<false>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess) { (void)hipGetLastError(); }
  auto timed = [&](auto launch) {
    for (int r = 0; r < 11; ++r) { if (r == 1) hipEventRecord(e.ev[2], e.stream); launch(); }
    hipEventRecord(e.ev[3], e.stream); hipEventSynchronize(e.ev[3]);
    float ms = 0; hipEventElapsedTime(&ms, e.ev[2], e.ev[3]);
    return static_cast<double>(ms) / 10.0;
  };
  out[0] = timed([&] { k_spmv_tile_probe<true><<<ntiles, 1024, lds, e.stream>>>(e.N, e.G, e.S10, e.sl10, e.hess, e.n10, e.xs, e.wall, ybuf); });
  out[4] = timed([&] { k_spmv_tile_probe<false><<<ntiles, 1024, lds, e.stream>>>(e.N, e.G, e.S10, e.sl10, e.hess, e.n10, e.xs, e.wall, ybuf); });
  out[1] = timed([&] { k_tile_gather_probe<<<nblk(e.N, 256), 256, 0, e.stream>>>(e.N, ntiles, ybuf, e.wall); });
  if (e.win_valid) {                                 // the real pass in the same process, after the probes
    const size_t ldsw = static_cast<size_t>(e.win_maxunits) * WIN_UNIT * sizeof(double2);
    out[2] = timed([&] { k_spmv_win<MODE_HSH, true, false, 2, WIN_PREFETCH | WIN_LEAN><<<e.win_groups, 64 * WIN_ROWS, ldsw, e.stream>>>(e.N, e.G, e.S10, e.dff, e.sl10, e.hess, e.n10, e.rows_sorted, e.win_k, e.win_cnt, e.xs, e.hst, e.gst, e.qst, e.q, e.type, e.scal, e.partials, e.wall, e.wgh, e.hsc, e.pqrow, nullptr, e.win_groups, 0, nullptr, e.win_flag); });
  }
  out[3] = (hipGetLastError() == hipSuccess) ? 0.0 : 1.0;
  (void)hipFree(ybuf);
}


}  // namespace rxmd
#endif   // RXMD_EXPERIMENTS
