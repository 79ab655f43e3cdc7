// micro-test: does the immediate offset of global_load_lds_dwordx4 move the LDS destination as well as the global source?
// and: saddr form (SGPR base + 32-bit VGPR offset), EXEC = 0 instruction counted by vmcnt.
#include <hip/hip_runtime.h>
#include <cstdio>
extern __shared__ __attribute__((aligned(16))) char smem[];
__global__ void k(const double *a, double *out) {
  const unsigned base = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<size_t>(smem)));
  const int lane = threadIdx.x;
  double *s = reinterpret_cast<double *>(smem);
  for (int i = lane; i < 1024; i += 64) s[i] = -1.0;
  __syncthreads();
  const unsigned voff = lane * 16;
  // saddr form, offset:1024 -> global bytes [1024, 2048) = doubles 128..255 ; M0 = base + 0
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 offset:1024 nt" : : "v"(voff), "s"(a), "s"(base) : "memory");
  // masked: first 10 lanes only, offset 0, M0 = base + 4096
  asm volatile("s_bfm_b64 exec, %3, 0\n\ts_cmp_gt_i32 %3, 63\n\ts_cmov_b64 exec, -1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\ts_mov_b64 exec, -1" : : "v"(voff), "s"(a), "s"(base + 4096u), "s"(10) : "memory", "scc");
  // EXEC = 0
  asm volatile("s_bfm_b64 exec, %3, 0\n\ts_cmp_gt_i32 %3, 63\n\ts_cmov_b64 exec, -1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\ts_mov_b64 exec, -1" : : "v"(voff), "s"(a), "s"(base + 6144u), "s"(0) : "memory", "scc");
  // all 64 (cnt = 64)
  asm volatile("s_bfm_b64 exec, %3, 0\n\ts_cmp_gt_i32 %3, 63\n\ts_cmov_b64 exec, -1\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt\n\ts_mov_b64 exec, -1" : : "v"(voff), "s"(a), "s"(base + 7168u - 1024u), "s"(64) : "memory", "scc");
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = s[i];
}
int main() {
  double *a, *o; hipMalloc(&a, 8192 * 8); hipMalloc(&o, 1024 * 8);
  double h[8192]; for (int i = 0; i < 8192; ++i) h[i] = i; hipMemcpy(a, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, 1, 64, 8192, 0, a, o); hipDeviceSynchronize();
  double r[1024]; hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
  int i = 0;
  while (i < 1024) { int j = i; while (j + 1 < 1024 && (r[j + 1] == r[j] + 1 || (r[j + 1] == -1 && r[j] == -1))) ++j; printf("lds doubles [%d..%d] = %g..%g\n", i, j, r[i], r[j]); i = j + 1; }
  return 0;
}
