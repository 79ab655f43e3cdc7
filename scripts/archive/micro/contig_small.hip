// Do small hipDeviceMallocContiguous allocations alias each other (or anything else)?  Round 3 found that an engine whose buffers came from
// hipExtMallocWithFlags(hipDeviceMallocContiguous) failed 15 unrelated tests with run-to-run different results, that 4 MB guard bands cured it
// and 4 KB bands did not; round 4's poisoned-allocation run of the whole GPU suite (RXMD_POISON_ALLOC=1) found no kernel that reads an element
// nobody wrote.  This probe takes the engine out of the picture: NBUF buffers of mixed small sizes (4 KB ... 3 MB) and a few large ones, each filled
// with its own pattern by a kernel, all verified afterwards -- with plain hipMalloc and with contiguous allocations; then the same with a
// hipMemsetAsync of every buffer on a second stream in between (what the engine's dmalloc does).
//   hipcc --offload-arch=gfx950 -O2 -o contig_small contig_small.hip && ./contig_small
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
__global__ void k_fill(unsigned long long *p, size_t n, unsigned long long tag) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x) p[i] = (tag << 40) | i;
}
__global__ void k_check(const unsigned long long *p, size_t n, unsigned long long tag, unsigned long long *bad) {
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += static_cast<size_t>(gridDim.x) * blockDim.x)
    if (p[i] != ((tag << 40) | i)) atomicAdd(bad, 1ULL);
}
static int run(bool contiguous, bool memset_between) {
  const int NBUF = 96;
  std::vector<unsigned long long *> buf(NBUF, nullptr);
  std::vector<size_t> len(NBUF);
  unsigned long long *bad = nullptr;
  CK(hipMalloc(reinterpret_cast<void **>(&bad), 8)); CK(hipMemset(bad, 0, 8));
  hipStream_t s1, s2; CK(hipStreamCreate(&s1)); CK(hipStreamCreate(&s2));
  unsigned seed = 12345u;
  for (int b = 0; b < NBUF; ++b) {
    seed = seed * 1664525u + 1013904223u;
    size_t bytes = (b % 12 == 11) ? (size_t(64) << 20) + (seed % 4096) * 8 : 4096 + (seed % (3u << 20)) / 8 * 8;   // mostly 4 KB ... 3 MB, every twelfth 64 MB
    len[b] = bytes / 8;
    if (contiguous) { if (hipExtMallocWithFlags(reinterpret_cast<void **>(&buf[b]), bytes, hipDeviceMallocContiguous) != hipSuccess) { std::printf("contiguous allocation of %zu bytes refused\n", bytes); (void)hipGetLastError(); return 3; } }
    else CK(hipMalloc(reinterpret_cast<void **>(&buf[b]), bytes));
    if (memset_between) CK(hipMemset(buf[b], 0xFF, bytes));      // (null stream, like the engine's dmalloc)
  }
  // address overlap of the virtual ranges
  long long overlaps = 0;
  for (int a = 0; a < NBUF; ++a) for (int b = a + 1; b < NBUF; ++b) {
    const char *a0 = reinterpret_cast<char *>(buf[a]), *a1 = a0 + len[a] * 8, *b0 = reinterpret_cast<char *>(buf[b]), *b1 = b0 + len[b] * 8;
    if (a0 < b1 && b0 < a1) ++overlaps;
  }
  for (int rep = 0; rep < 3; ++rep) {
    for (int b = 0; b < NBUF; ++b) k_fill<<<64, 256, 0, (b & 1) ? s1 : s2>>>(buf[b], len[b], static_cast<unsigned long long>(b + 1 + 100 * rep));
    CK(hipDeviceSynchronize());
    for (int b = 0; b < NBUF; ++b) k_check<<<64, 256, 0, (b & 1) ? s2 : s1>>>(buf[b], len[b], static_cast<unsigned long long>(b + 1 + 100 * rep), bad);
    CK(hipDeviceSynchronize());
  }
  unsigned long long hb = 0; CK(hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost));
  std::printf("%-10s %-14s : %d buffers, %lld overlapping virtual ranges, %llu wrong words\n", contiguous ? "contiguous" : "hipMalloc", memset_between ? "memset at alloc" : "no memset", NBUF, overlaps, hb);
  for (auto *p : buf) (void)hipFree(p);
  (void)hipFree(bad); (void)hipStreamDestroy(s1); (void)hipStreamDestroy(s2);
  return (hb || overlaps) ? 1 : 0;
}
int main() {
  int rc = 0;
  rc |= run(false, false); rc |= run(false, true); rc |= run(true, false); rc |= run(true, true);
  std::printf("rc=%d\n", rc);
  return 0;
}
