"""Multi-GPU communication layer: one process per GPU, torch.distributed.

The C library decomposes the grid into row slabs and calls back into this module for every
exchange (include/pylamp_hip.h, pl_comm_ops).  Two transports:

  * backend "nccl" (= RCCL over xGMI on a real node): device pointers of the library are
    wrapped zero-copy as torch tensors (__cuda_array_interface__) and moved with
    batch_isend_irecv / all_gather_into_tensor; host scalars go through a side gloo group.
  * backend "gloo" (CPU tests, or several ranks sharing ONE GPU): device buffers are staged
    through host memory with pl_memcpy_d2h / pl_memcpy_h2d.

The reference's own scheme (replicated grid, rank-strided tracers, O(ntrac) Allreduce:
pylamp2.py:445-455,550-555) is deliberately not reproduced.
"""
import ctypes as C

import numpy as np

from . import _lib

_EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int,
                        C.c_int64, C.c_int)
_ALLREDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_int64, C.c_int)
_ALLGATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_int64)
_EXCHANGE_VAR = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_void_p), C.c_int64, C.POINTER(C.c_void_p), C.c_int64,
                            C.POINTER(C.c_void_p), C.c_int64, C.c_int, C.POINTER(C.c_int64))


class CommOps(C.Structure):
    _fields_ = [("exchange", _EXCHANGE), ("allreduce_host", _ALLREDUCE), ("allgather", _ALLGATHER),
                ("exchange_var", _EXCHANGE_VAR), ("user", C.c_void_p)]


class _DevView:
    """Zero-copy view of raw device memory for torch.as_tensor."""

    def __init__(self, ptr, n):
        self.__cuda_array_interface__ = {"shape": (int(n),), "typestr": "<f8", "data": (int(ptr), False),
                                         "version": 3, "strides": None}


def slab_rows(nz, rank, size):
    """Node rows [first, first+n) owned by `rank` (mirrors pl_set_comm)."""
    cells = nz - 1
    if cells % size:
        raise Exception("(nz-1) must be divisible by the number of ranks")
    c = cells // size
    return rank * c, (c + 1 if rank == size - 1 else c)


class Comm:
    def __init__(self, ctx, group=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.ctx = ctx
        self.group = group
        self.rank = dist.get_rank(group)
        self.size = dist.get_world_size(group)
        self.backend = dist.get_backend(group)
        import os
        self.device_mode = self.backend == "nccl" and not os.environ.get("PYLAMP_COMM_STAGING")
        # host scalars (and the staging fallback) always travel through gloo
        self.host_group = group if self.backend == "gloo" else dist.new_group(backend="gloo")
        self.stage_group = self.host_group
        self.errors = []
        self._ops = CommOps(_EXCHANGE(self._exchange), _ALLREDUCE(self._allreduce_host), _ALLGATHER(self._allgather),
                            _EXCHANGE_VAR(self._exchange_var), None)
        if not self.device_mode:
            os.environ["PYLAMP_RCCL"] = "0"      # direct RCCL needs one GPU per rank (nccl backend)
        ctx.check(ctx.lib.pl_set_comm(ctx.handle(), self.rank, self.size, C.byref(self._ops)))
        nat = C.c_int(0)
        ctx.check(ctx.lib.pl_comm_info(ctx.handle(), None, None, C.byref(nat)))
        self.native = bool(nat.value)
        ctx.comm = self          # keep the callbacks alive as long as the context

    # ---- helpers ------------------------------------------------------------------------------
    def _dev(self, ptr, n):
        return self.torch.as_tensor(_DevView(ptr, n), device="cuda")

    def _d2h(self, ptr, n):
        a = np.empty(int(n))
        self.ctx.check(self.ctx.lib.pl_memcpy_d2h(self.ctx.handle(), a.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), a.nbytes))
        return a

    def _h2d(self, ptr, a):
        a = np.ascontiguousarray(a)
        self.ctx.check(self.ctx.lib.pl_memcpy_h2d(self.ctx.handle(), C.c_void_p(ptr), a.ctypes.data_as(C.c_void_p), a.nbytes))

    def _guard(self, fn, *a):
        try:
            fn(*a)
            return 0
        except Exception as e:          # exceptions must not cross the C frame
            self.errors.append(repr(e))
            return 1

    def _agree_device_mode(self):
        """All ranks must use the same transport: if the zero-copy device path is unusable on any
        rank, everybody stages through the host (collective decision, once)."""
        if getattr(self, "_agreed", False):
            return
        ok = 1
        if self.device_mode:
            try:
                probe = self.torch.zeros(4, dtype=self.torch.float64, device="cuda")
                v = self._dev(probe.data_ptr(), 4)
                v += 1.0
                self.torch.cuda.synchronize()
                ok = int(bool((probe == 1.0).all().item()))
            except Exception as e:
                self.errors.append("device path disabled: " + repr(e))
                ok = 0
        t = self.torch.tensor([ok], dtype=self.torch.int64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MIN, group=self.host_group)
        if int(t.item()) == 0:
            self.device_mode = False
        self._agreed = True

    # ---- neighbour exchange ------------------------------------------------------------------------
    def _exchange(self, user, send_lo, recv_lo, send_hi, recv_hi, count, nseg, stride, add):
        return self._guard(self._exchange_impl, send_lo, recv_lo, send_hi, recv_hi, count, nseg, stride, add)

    def _exchange_impl(self, send_lo, recv_lo, send_hi, recv_hi, count, nseg, stride, add):
        dist, torch = self.dist, self.torch
        self._agree_device_mode()
        lo, hi = self.rank - 1, self.rank + 1
        has_lo, has_hi = lo >= 0, hi < self.size
        B = 8
        if self.device_mode:
            ops, post = [], []
            for k in range(nseg):
                o = k * stride * B
                if has_lo:
                    ops.append(dist.P2POp(dist.isend, self._dev(send_lo + o, count), lo, self.group))
                    if add:
                        t = torch.empty(count, dtype=torch.float64, device="cuda")
                        ops.append(dist.P2POp(dist.irecv, t, lo, self.group)); post.append((recv_lo + o, t))
                    else:
                        ops.append(dist.P2POp(dist.irecv, self._dev(recv_lo + o, count), lo, self.group))
                if has_hi:
                    ops.append(dist.P2POp(dist.isend, self._dev(send_hi + o, count), hi, self.group))
                    if add:
                        t = torch.empty(count, dtype=torch.float64, device="cuda")
                        ops.append(dist.P2POp(dist.irecv, t, hi, self.group)); post.append((recv_hi + o, t))
                    else:
                        ops.append(dist.P2POp(dist.irecv, self._dev(recv_hi + o, count), hi, self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            for ptr, t in post:
                self._dev(ptr, count).add_(t)
            torch.cuda.synchronize()
            return
        # gloo: stage through the host; pack all segments of one direction into one message
        def pack(base):
            return np.concatenate([self._d2h(base + k * stride * B, count) for k in range(nseg)])
        reqs, rl, rh = [], None, None
        if has_lo:
            sl = torch.from_numpy(pack(send_lo)); rl = torch.empty(nseg * count, dtype=torch.float64)
            reqs += [dist.isend(sl, lo, self.stage_group), dist.irecv(rl, lo, self.stage_group)]
        if has_hi:
            sh = torch.from_numpy(pack(send_hi)); rh = torch.empty(nseg * count, dtype=torch.float64)
            reqs += [dist.isend(sh, hi, self.stage_group), dist.irecv(rh, hi, self.stage_group)]
        for r in reqs:
            r.wait()
        for base, buf in ((recv_lo, rl), (recv_hi, rh)):
            if buf is None:
                continue
            a = buf.numpy()
            for k in range(nseg):
                seg = a[k * count:(k + 1) * count]
                if add:
                    seg = seg + self._d2h(base + k * stride * B, count)
                self._h2d(base + k * stride * B, seg)

    # ---- host all-reduce ------------------------------------------------------------------------------
    def _allreduce_host(self, user, buf, n, op):
        return self._guard(self._allreduce_impl, buf, n, op)

    def _allreduce_impl(self, buf, n, op):
        a = np.ctypeslib.as_array(buf, shape=(int(n),))
        t = self.torch.from_numpy(a)
        rop = {0: self.dist.ReduceOp.SUM, 1: self.dist.ReduceOp.MIN, 2: self.dist.ReduceOp.MAX}[int(op)]
        self.dist.all_reduce(t, op=rop, group=self.host_group)

    # ---- all-gather (in place: rank r owns recv[r*count : (r+1)*count] of every segment) ----------------
    def _allgather(self, user, recv, count, nseg, stride):
        return self._guard(self._allgather_impl, recv, count, nseg, stride)

    def _allgather_impl(self, recv, count, nseg, stride):
        dist, torch = self.dist, self.torch
        self._agree_device_mode()
        B = 8
        for k in range(nseg):
            base = recv + k * stride * B
            if self.device_mode:
                full = self._dev(base, count * self.size)
                mine = full[self.rank * count:(self.rank + 1) * count].clone()
                dist.all_gather_into_tensor(full, mine, group=self.group)
            else:
                mine = torch.from_numpy(self._d2h(base + self.rank * count * B, count))
                out = [torch.empty(count, dtype=torch.float64) for _ in range(self.size)]
                dist.all_gather(out, mine, group=self.stage_group)
                self._h2d(base, np.concatenate([o.numpy() for o in out]))
        if self.device_mode:
            torch.cuda.synchronize()

    # ---- variable-size neighbour exchange of tracer columns ----------------------------------------------
    def _exchange_var(self, user, send_lo, n_lo, send_hi, n_hi, recv, cap, ncol, got):
        return self._guard(self._exchange_var_impl, send_lo, n_lo, send_hi, n_hi, recv, cap, ncol, got)

    def _exchange_var_impl(self, send_lo, n_lo, send_hi, n_hi, recv, cap, ncol, got):
        dist, torch = self.dist, self.torch
        self._agree_device_mode()
        lo, hi = self.rank - 1, self.rank + 1
        has_lo, has_hi = lo >= 0, hi < self.size
        # 1. counts (host, gloo)
        cnt_from_lo = torch.zeros(1, dtype=torch.int64); cnt_from_hi = torch.zeros(1, dtype=torch.int64)
        reqs = []
        if has_lo:
            reqs += [dist.isend(torch.tensor([int(n_lo)], dtype=torch.int64), lo, self.host_group),
                     dist.irecv(cnt_from_lo, lo, self.host_group)]
        if has_hi:
            reqs += [dist.isend(torch.tensor([int(n_hi)], dtype=torch.int64), hi, self.host_group),
                     dist.irecv(cnt_from_hi, hi, self.host_group)]
        for r in reqs:
            r.wait()
        m_lo, m_hi = int(cnt_from_lo.item()), int(cnt_from_hi.item())
        if m_lo + m_hi > cap:
            raise Exception("tracer migration exceeds the receive capacity")
        got[0] = m_lo + m_hi
        B = 8
        # 2. payload, column by column (all columns of one direction in one message under gloo)
        if self.device_mode:
            ops = []
            for k in range(ncol):
                if has_lo and n_lo:
                    ops.append(dist.P2POp(dist.isend, self._dev(send_lo[k], n_lo), lo, self.group))
                if has_hi and n_hi:
                    ops.append(dist.P2POp(dist.isend, self._dev(send_hi[k], n_hi), hi, self.group))
                if m_lo:
                    ops.append(dist.P2POp(dist.irecv, self._dev(recv[k], m_lo), lo, self.group))
                if m_hi:
                    ops.append(dist.P2POp(dist.irecv, self._dev(recv[k] + m_lo * B, m_hi), hi, self.group))
            if ops:
                for w in dist.batch_isend_irecv(ops):
                    w.wait()
            torch.cuda.synchronize()
            return
        reqs, rl, rh = [], None, None
        if has_lo and n_lo:
            reqs.append(dist.isend(torch.from_numpy(np.concatenate([self._d2h(send_lo[k], n_lo) for k in range(ncol)])),
                                   lo, self.stage_group))
        if has_hi and n_hi:
            reqs.append(dist.isend(torch.from_numpy(np.concatenate([self._d2h(send_hi[k], n_hi) for k in range(ncol)])),
                                   hi, self.stage_group))
        if m_lo:
            rl = torch.empty(ncol * m_lo, dtype=torch.float64); reqs.append(dist.irecv(rl, lo, self.stage_group))
        if m_hi:
            rh = torch.empty(ncol * m_hi, dtype=torch.float64); reqs.append(dist.irecv(rh, hi, self.stage_group))
        for r in reqs:
            r.wait()
        for k in range(ncol):
            if m_lo:
                self._h2d(recv[k], rl.numpy()[k * m_lo:(k + 1) * m_lo])
            if m_hi:
                self._h2d(recv[k] + m_lo * B, rh.numpy()[k * m_hi:(k + 1) * m_hi])
