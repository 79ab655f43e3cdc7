// rccl_comm.hip -- native transport of the six-stage exchange: RCCL point-to-point and all-reduce enqueued on the engine's
// own HIP stream (reference: the MPI_Send/MPI_Recv pairs of send_recv, src/comm.F90:291-364, and the MPI_ALLREDUCE call
// sites of src/qeq.F90:107,129,144,357).  One communicator per engine = per GPU; between two MI355X of a node the bytes
// go over the xGMI link that joins them.  Nothing here synchronises the host except the size message of a stage whose
// receive count is not known in advance (ghost build, migration: 12 per step); vector halos, the force fold and the CG
// scalars stay in stream order with the kernels that pack, unpack and consume them.
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <thread>

#include "engine.h"

namespace rxmd {

#define RX_NCCL(call)                                                                                          \
  do {                                                                                                         \
    ncclResult_t r_ = (call);                                                                                  \
    if (r_ != ncclSuccess) throw EngineError(RXMD_E_COMM, std::string(#call) + ": " + ncclGetErrorString(r_)); \
  } while (0)

static inline ncclComm_t C(void *p) { return static_cast<ncclComm_t>(p); }

void Engine::rccl_init(const unsigned char id128[128], int rank, int world) {
  if (world != nprocs && !(force_staged && world == 1)) throw EngineError(RXMD_E_ARG, "RCCL world size does not match vprocs");
  if (rank != cfg.myid) throw EngineError(RXMD_E_ARG, "RCCL rank does not match myid");
  ncclUniqueId id;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  std::memcpy(&id, id128, 128);
  RX_HIP(hipSetDevice(cfg.device));
  ncclComm_t c;
  RX_NCCL(ncclCommInitRank(&c, world, id, rank));
  nccl = c;
  int nr = 0;
  RX_NCCL(ncclCommCount(c, &nr));
  if (nr != world) throw EngineError(RXMD_E_COMM, "RCCL communicator has " + std::to_string(nr) + " ranks, expected " + std::to_string(world));
  if (!cnt_dev) { RX_HIP(hipMalloc(reinterpret_cast<void **>(&cnt_dev), 4 * sizeof(double))); RX_HIP(hipHostMalloc(reinterpret_cast<void **>(&cnt_host), 4 * sizeof(double))); }
}

// bounded wait (see engine.h): `query` polls the stream or the event
template <class Q>
static void watched_wait(Engine &e, Q &&query, const char *what) {
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    const hipError_t r = query();
    if (r == hipSuccess) return;
    if (r != hipErrorNotReady) throw EngineError(RXMD_E_HIP, std::string(what) + ": " + hipGetErrorString(r));
    if ((++spins & 1023u) == 0u) {
      const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (waited > e.comm_timeout_s) {
        if (e.nccl) { (void)ncclCommAbort(C(e.nccl)); e.nccl = nullptr; }
        throw EngineError(RXMD_E_COMM, "rank " + std::to_string(e.cfg.myid) + ": a wait that depends on the other ranks did not finish within " +
                                           std::to_string(static_cast<int>(e.comm_timeout_s)) + " s (" + what + "); communicator aborted");
      }
      if (waited > 0.01) std::this_thread::sleep_for(std::chrono::microseconds(50));   // long waits stop burning the core
    }
  }
}
void Engine::sync_stream() {
  if (!nccl) { RX_HIP(hipStreamSynchronize(stream)); return; }
  watched_wait(*this, [&] { return hipStreamQuery(stream); }, "stream synchronisation");
}
void Engine::sync_event(hipEvent_t ev_) {
  if (!nccl) {
    // the per-iteration wait of the CG loop: on some hosts a blocking wait wakes up 50-100 us after the event (the GPU then idles between
    // the iterations); spin_wait polls instead (RXMD_SPIN_WAIT=0: the blocking wait)
    // -- for a bounded time (2 ms: twice a matrix pass), then it blocks: ranks that share host cores with a callback transport's threads, or a
    // hung GPU, do not burn a core for good.  With a callback transport of several ranks the spin is off by default (has_comm && nprocs > 1).
    if (spin_wait && !(has_comm && nprocs > 1)) {
      const auto t0 = std::chrono::steady_clock::now();
      unsigned spins = 0;
      for (;;) {
        const hipError_t r = hipEventQuery(ev_);
        if (r == hipSuccess) return;
        if (r != hipErrorNotReady) RX_HIP(r);
        if ((++spins & 255u) == 0u && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2e-3) break;
      }
    }
    RX_HIP(hipEventSynchronize(ev_)); return;
  }
  watched_wait(*this, [&] { return hipEventQuery(ev_); }, "event synchronisation");
}

// The host's wait of the run-ahead CG loop: the update kernel's tail stores the iteration's scalars into pinned host memory and their sequence number
// last (qeq.hip: scalar_algebra stage 6); the host polls that word -- no HIP call on the way.  Bounded: beyond 2 ms (twice a matrix pass) every round
// asks the stream whether the device is still alive, yields the core, and gives up after the time limit of every other wait.
void Engine::wait_snapshot(int parity, double seq) {
  volatile const double *flag = h_scal + 64 + 64 * parity + 63;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  for (;;) {
    if (*flag == seq) break;
    if ((++spins & 1023u) != 0u && spin_wait) continue;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt <= 2e-3 && spin_wait) continue;
    const hipError_t r = hipStreamQuery(stream);
    if (r != hipSuccess && r != hipErrorNotReady) RX_HIP(r);
    if (r == hipSuccess) { if (*flag == seq) break; throw EngineError(RXMD_E_HIP, "the CG scalars of an iteration never reached the host"); }
    if (dt > comm_timeout_s) throw EngineError(RXMD_E_HIP, "timeout waiting for the CG scalars of an iteration");
    std::this_thread::yield();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
}

// The same for counts a kernel leaves in pinned host memory (engine.h: h_pub): every word must carry the sequence number of the request.
void Engine::pinned_wait(int nwords, unsigned seq, const char *what) {
  volatile const unsigned long long *w = h_pub;
  const auto t0 = std::chrono::steady_clock::now();
  unsigned spins = 0;
  auto all_here = [&] { for (int k = 0; k < nwords; ++k) if (static_cast<unsigned>(w[k] >> 32) != seq) return false; return true; };
  for (;;) {
    if (all_here()) break;
    if ((++spins & 1023u) != 0u && spin_wait) continue;
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (dt <= 2e-3 && spin_wait) continue;
    const hipError_t r = hipStreamQuery(stream);
    if (r != hipSuccess && r != hipErrorNotReady) RX_HIP(r);
    if (r == hipSuccess) { if (all_here()) break; throw EngineError(RXMD_E_HIP, std::string(what) + ": the counts never reached the host"); }
    if (dt > comm_timeout_s) throw EngineError(RXMD_E_HIP, std::string(what) + ": timeout waiting for the counts of a kernel");
    std::this_thread::yield();
  }
  std::atomic_thread_fence(std::memory_order_acquire);
}

void Engine::rccl_destroy() {
  if (nccl) { (void)ncclCommDestroy(C(nccl)); nccl = nullptr; }
  if (cnt_dev) { (void)hipFree(cnt_dev); cnt_dev = nullptr; }
  if (cnt_host) { (void)hipHostFree(cnt_host); cnt_host = nullptr; }
}

// one send_recv: nsend doubles of xbuf_send to `to`, the message of `from` into xbuf_recv; returns the doubles received
long long Engine::rccl_exchange(int to, int from, long long nsend, long long known_nrecv) {
  ncclComm_t c = C(nccl);
  long long nrecv = known_nrecv;
  if (nrecv < 0) {                                        // size message first (the reference sends the size inside MPI_Probe, comm.F90:329-341)
    cnt_host[0] = static_cast<double>(nsend);
    RX_HIP(hipMemcpyAsync(cnt_dev, cnt_host, sizeof(double), hipMemcpyHostToDevice, stream));
    RX_NCCL(ncclGroupStart());
    RX_NCCL(ncclSend(cnt_dev, 1, ncclDouble, to, c, stream));
    RX_NCCL(ncclRecv(cnt_dev + 1, 1, ncclDouble, from, c, stream));
    RX_NCCL(ncclGroupEnd());
    RX_HIP(hipMemcpyAsync(cnt_host + 1, cnt_dev + 1, sizeof(double), hipMemcpyDeviceToHost, stream));
    sync_stream();
    nrecv = static_cast<long long>(cnt_host[1]);
    if (nrecv > static_cast<long long>(xbuf_doubles)) grow_xbuf_keep_send(static_cast<size_t>(nrecv), static_cast<size_t>(nsend));
  }
  if (nsend > 0 || nrecv > 0) {
    RX_NCCL(ncclGroupStart());
    if (nsend > 0) RX_NCCL(ncclSend(xbuf_send, static_cast<size_t>(nsend), ncclDouble, to, c, stream));
    if (nrecv > 0) RX_NCCL(ncclRecv(xbuf_recv, static_cast<size_t>(nrecv), ncclDouble, from, c, stream));
    RX_NCCL(ncclGroupEnd());
  }
  return nrecv;
}

// both send_recv of one axis in one group: four point-to-point operations in flight at once.  Two messages to the same peer
// (an axis split in two: the + and the - neighbour are the same rank) are matched in the order they were issued, which is the
// order the peer posts its receives in (its own stage d0 message first).
void Engine::rccl_exchange_pair(int to0, int from0, long long n0, long long r0, int to1, int from1, long long n1, long long r1) {
  ncclComm_t c = C(nccl);
  if (n0 + n1 + r0 + r1 == 0) return;
  RX_NCCL(ncclGroupStart());
  if (n0 > 0) RX_NCCL(ncclSend(xbuf_send, static_cast<size_t>(n0), ncclDouble, to0, c, stream));
  if (r0 > 0) RX_NCCL(ncclRecv(xbuf_recv, static_cast<size_t>(r0), ncclDouble, from0, c, stream));
  if (n1 > 0) RX_NCCL(ncclSend(xbuf_send + n0, static_cast<size_t>(n1), ncclDouble, to1, c, stream));
  if (r1 > 0) RX_NCCL(ncclRecv(xbuf_recv + r0, static_cast<size_t>(r1), ncclDouble, from1, c, stream));
  RX_NCCL(ncclGroupEnd());
}

// the pair with unknown receive counts: one group for the two size messages, one host wait, one group for the two payloads
void Engine::rccl_exchange_pair_sized(int to0, int from0, long long n0, long long &r0, int to1, int from1, long long n1, long long &r1) {
  ncclComm_t c = C(nccl);
  cnt_host[0] = static_cast<double>(n0); cnt_host[1] = static_cast<double>(n1);
  RX_HIP(hipMemcpyAsync(cnt_dev, cnt_host, 2 * sizeof(double), hipMemcpyHostToDevice, stream));
  RX_NCCL(ncclGroupStart());
  RX_NCCL(ncclSend(cnt_dev, 1, ncclDouble, to0, c, stream));
  RX_NCCL(ncclRecv(cnt_dev + 2, 1, ncclDouble, from0, c, stream));
  RX_NCCL(ncclSend(cnt_dev + 1, 1, ncclDouble, to1, c, stream));
  RX_NCCL(ncclRecv(cnt_dev + 3, 1, ncclDouble, from1, c, stream));
  RX_NCCL(ncclGroupEnd());
  RX_HIP(hipMemcpyAsync(cnt_host + 2, cnt_dev + 2, 2 * sizeof(double), hipMemcpyDeviceToHost, stream));
  sync_stream();
  r0 = static_cast<long long>(cnt_host[2]); r1 = static_cast<long long>(cnt_host[3]);
  if (r0 + r1 > static_cast<long long>(xbuf_doubles)) grow_xbuf_keep_send(static_cast<size_t>(r0 + r1), static_cast<size_t>(n0 + n1));
  rccl_exchange_pair(to0, from0, n0, r0, to1, from1, n1, r1);
}

// direct halo: one group with every peer (engine.hip: exchange_many); the rank's own segment only in the self-loop test mode
void Engine::rccl_exchange_many(const std::vector<long long> &soff, const std::vector<long long> &roff, int ncomp) {
  ncclComm_t c = C(nccl);
  const int np = (force_staged && nprocs == 1) ? 1 : nprocs;
  bool any = false;
  for (int p = 0; p < np && !any; ++p) any = (p != cfg.myid || force_remote) && (soff[p + 1] > soff[p] || roff[p + 1] > roff[p]);
  if (!any) return;
  RX_NCCL(ncclGroupStart());
  for (int p = 0; p < np; ++p) {
    if (p == cfg.myid && !force_remote) continue;
    const long long sc = (soff[p + 1] - soff[p]) * ncomp, rc = (roff[p + 1] - roff[p]) * ncomp;
    if (sc > 0) RX_NCCL(ncclSend(xbuf_send + soff[p] * ncomp, static_cast<size_t>(sc), ncclDouble, p, c, stream));
    if (rc > 0) RX_NCCL(ncclRecv(xbuf_recv + roff[p] * ncomp, static_cast<size_t>(rc), ncclDouble, p, c, stream));
  }
  RX_NCCL(ncclGroupEnd());
}

// MPI_ALLREDUCE(SUM) of n device doubles, in place, in stream order
void Engine::rccl_allreduce_dev(double *dev, int n) {
  RX_NCCL(ncclAllReduce(dev, dev, static_cast<size_t>(n), ncclDouble, ncclSum, C(nccl), stream));
}

}  // namespace rxmd

extern "C" int rxmd_hip_rccl_unique_id(unsigned char out128[128]) {
  ncclUniqueId id;
  if (!out128 || ncclGetUniqueId(&id) != ncclSuccess) return RXMD_E_COMM;
  std::memcpy(out128, &id, 128);
  return RXMD_OK;
}
