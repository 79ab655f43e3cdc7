"""Transport for the engine's six-stage exchange (the reference's COPYATOMS send_recv,
src/comm.F90:291-364, and its MPI_ALLREDUCE call sites) on top of torch.distributed.

The engine owns the exchange algorithm (which atoms, which stage, which partner); this module only
moves bytes:  `exchange(to, send, nsend, from, recv, cap) -> nrecv`  and  `allreduce_sum(buf, n)`.

Modes
  device  : backend "nccl" (= RCCL over xGMI on an MI355X node).  The two message buffers are torch
            CUDA tensors handed to the engine with rxmd_hip_set_exchange_buffers, so the callback
            sends slices of its own tensors: grouped isend/irecv of the count, then of the payload.
  staged  : any backend with CPU tensors ("gloo"): device buffers are staged through host memory.
            Used to run several ranks on ONE GPU in the tests, and on hosts without RCCL.
  host    : pure host buffers (ctypes pointers) over gloo: the CPU test of the transport contract.
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

from . import _lib
from ._lib import RxmdCommOps, EXCHANGE_FN, ALLREDUCE_FN


class TorchTransport:
    def __init__(self, mode="device", group=None, device=None, capacity_doubles=1 << 22):
        assert mode in ("device", "staged", "host")
        self.mode = mode
        self.group = group
        self.rank = dist.get_rank(group)
        self.world = dist.get_world_size(group)
        self.device = device
        self.cap = int(capacity_doubles)
        self.send_t = self.recv_t = None
        if mode in ("device", "staged"):
            self.send_t = torch.zeros(self.cap, dtype=torch.float64, device=device)
            self.recv_t = torch.zeros(self.cap, dtype=torch.float64, device=device)
        self._cb_ex = EXCHANGE_FN(self._exchange)
        self._cb_ar = ALLREDUCE_FN(self._allreduce)
        self._cb_exk = EXCHANGE_FN(self._exchange_known)
        self.ops = RxmdCommOps(None, self._cb_ex, self._cb_ar, self._cb_exk)
        self.n_exchange = 0
        self.n_allreduce = 0
        self.error = None

    # ---- wiring ----
    def attach(self, engine):
        L = engine.L
        if self.send_t is not None:
            rc = L.rxmd_hip_set_exchange_buffers(engine.h, C.c_void_p(self.send_t.data_ptr()), C.c_void_p(self.recv_t.data_ptr()), self.cap)
            engine._chk(rc)
        engine._chk(L.rxmd_hip_set_comm(engine.h, C.byref(self.ops)))
        engine._transport = self      # keep the callbacks alive as long as the engine

    # ---- callbacks ----
    def _p2p(self, to, t_send, frm, t_recv):
        ops = [dist.P2POp(dist.isend, t_send, to, self.group), dist.P2POp(dist.irecv, t_recv, frm, self.group)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()

    def _exchange(self, ctx, to, send_ptr, nsend, frm, recv_ptr, cap):
        try:
            self.n_exchange += 1
            nsend = int(nsend)
            # the engine may pass pointers INTO the registered buffers (second message of an axis pair): honour the offsets
            os_ = (int(send_ptr) - self.send_t.data_ptr()) // 8 if self.mode != "host" else 0
            or_ = (int(recv_ptr) - self.recv_t.data_ptr()) // 8 if self.mode != "host" else 0
            if self.mode == "device":
                cnt_s = torch.tensor([nsend], dtype=torch.int64, device=self.device)
                cnt_r = torch.zeros(1, dtype=torch.int64, device=self.device)
                self._p2p(to, cnt_s, frm, cnt_r)
                nrecv = int(cnt_r.item())
                if nrecv > cap:
                    return -1
                # every rank always posts both operations (a 1-element dummy stands for an empty message, like the
                # reference's 1-double sentinel, comm.F90:321-327) so that sends and receives pair up on any rank grid
                s = self.send_t[os_:os_ + max(nsend, 1)]
                r = self.recv_t[or_:or_ + max(nrecv, 1)] if nrecv > 0 else torch.zeros(1, dtype=torch.float64, device=self.device)
                self._p2p(to, s, frm, r)
                torch.cuda.synchronize(self.device)
                return nrecv
            # host-staged / host
            cnt_s = torch.tensor([nsend], dtype=torch.int64)
            cnt_r = torch.zeros(1, dtype=torch.int64)
            self._p2p(to, cnt_s, frm, cnt_r)
            nrecv = int(cnt_r.item())
            if nrecv > cap:
                return -1
            if self.mode == "staged":
                s = self.send_t[os_:os_ + max(nsend, 1)].cpu()
            else:
                s = torch.from_numpy(np.ctypeslib.as_array(C.cast(C.c_void_p(send_ptr), C.POINTER(C.c_double)), shape=(max(nsend, 1),)).copy()) if nsend > 0 else torch.zeros(1, dtype=torch.float64)
            r = torch.zeros(max(nrecv, 1), dtype=torch.float64)
            self._p2p(to, s, frm, r)
            if nrecv > 0:
                if self.mode == "staged":
                    self.recv_t[or_:or_ + nrecv].copy_(r[:nrecv])
                    torch.cuda.synchronize(self.device)
                else:
                    C.memmove(recv_ptr, r.numpy().ctypes.data, 8 * nrecv)
            return nrecv
        except Exception as ex:          # never let an exception cross the C boundary
            self.error = ex
            return -1

    def _exchange_known(self, ctx, to, send_ptr, nsend, frm, recv_ptr, nrecv):
        """payload only: the receive size is known from the ghost build (vector halos, force return)"""
        try:
            self.n_exchange += 1
            nsend = int(nsend); nrecv = int(nrecv)
            # the engine may pass pointers INTO the registered buffers (second message of an axis pair): honour the offsets
            os_ = (int(send_ptr) - self.send_t.data_ptr()) // 8 if self.mode != "host" else 0
            or_ = (int(recv_ptr) - self.recv_t.data_ptr()) // 8 if self.mode != "host" else 0
            if self.mode == "device":
                s = self.send_t[os_:os_ + max(nsend, 1)]
                r = self.recv_t[or_:or_ + nrecv] if nrecv > 0 else torch.zeros(1, dtype=torch.float64, device=self.device)
                self._p2p(to, s, frm, r)
                torch.cuda.synchronize(self.device)
                return nrecv
            if self.mode == "staged":
                s = self.send_t[os_:os_ + max(nsend, 1)].cpu()
            else:
                s = torch.from_numpy(np.ctypeslib.as_array(C.cast(C.c_void_p(send_ptr), C.POINTER(C.c_double)), shape=(max(nsend, 1),)).copy()) if nsend > 0 else torch.zeros(1, dtype=torch.float64)
            r = torch.zeros(max(nrecv, 1), dtype=torch.float64)
            self._p2p(to, s, frm, r)
            if nrecv > 0:
                if self.mode == "staged":
                    self.recv_t[or_:or_ + nrecv].copy_(r[:nrecv])
                    torch.cuda.synchronize(self.device)
                else:
                    C.memmove(recv_ptr, r.numpy().ctypes.data, 8 * nrecv)
            return nrecv
        except Exception as ex:
            self.error = ex
            return -1

    def _allreduce(self, ctx, buf, n):
        try:
            self.n_allreduce += 1
            a = np.ctypeslib.as_array(buf, shape=(n,))
            if self.mode == "device":
                t = torch.from_numpy(a.copy()).to(self.device)
                dist.all_reduce(t, group=self.group)
                a[:] = t.cpu().numpy()
            else:
                t = torch.from_numpy(a.copy())
                dist.all_reduce(t, group=self.group)
                a[:] = t.numpy()
            return 0
        except Exception as ex:
            self.error = ex
            return 1

    def selftest(self):
        """transport contract check with host buffers (include/rxmd_hip.h: rxmd_host_comm_selftest)"""
        L = _lib.load()
        return L.rxmd_host_comm_selftest(C.byref(self.ops), self.rank, self.world)
