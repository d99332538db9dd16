"""One device context per (grid) — owner of all HIP memory behind the module API."""
import collections
import ctypes as C
import os
import weakref

import numpy as np

from . import _lib


class Context:
    def __init__(self, nx, grid, device=None, attach_dist=True):
        lib = _lib.load()
        self.lib = lib
        self.nz, self.nx = int(nx[0]), int(nx[1])
        self.gz = _lib.f64(grid[0]).copy()
        self.gx = _lib.f64(grid[1]).copy()
        if self.gz.size != self.nz or self.gx.size != self.nx:
            raise Exception("grid arrays do not match nx")
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0")) if os.environ.get("PYLAMP_DEVICE") is None \
                else int(os.environ["PYLAMP_DEVICE"])
        h = C.c_void_p()
        rc = lib.pl_create(C.byref(h), int(device), self.nz, self.nx, _lib.dptr(self.gz), _lib.dptr(self.gx))
        if rc != 0:
            msg = lib.pl_last_error(None)
            raise Exception(msg.decode() if msg else "pl_create failed")
        self.h = h
        self._fin = weakref.finalize(self, lib.pl_destroy, h)
        # operators handed out by makeStokesMatrix / makeDiffusionMatrix on this context (weak: an operator that
        # is still alive keeps its context from being evicted; see get_context) and the generation of the
        # coefficients currently resident on the device (an operator of an older generation re-uploads its own)
        self.operators = weakref.WeakSet()
        self.stokes_gen = 0
        self.heat_gen = 0
        self.comm = None
        self.rank, self.nranks = 0, 1
        # under torch.distributed (torchrun) every rank owns a block of the grid
        try:
            import torch.distributed as dist
            if attach_dist and dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1 and not os.environ.get("PYLAMP_NO_DIST"):
                from .parallel import Comm
                Comm(self)
                self.rank, self.nranks = self.comm.rank, self.comm.size
        except ImportError:
            pass

    @property
    def closed(self):
        return self.h is None

    def handle(self):
        """Native handle for a library call; a closed context raises instead of passing freed memory to C."""
        if self.h is None:
            raise Exception("pylamp_amd: this device context has been closed")
        return self.h

    def check(self, rc):
        if self.h is None:
            raise Exception("pylamp_amd: this device context has been closed")
        if rc != 0 and self.comm is not None and self.comm.errors:
            raise Exception("communication layer: " + "; ".join(self.comm.errors[-3:]))
        _lib.check(self.h, rc)

    def local_rows(self):
        a = C.c_int(); b = C.c_int()
        self.check(self.lib.pl_local_rows(self.handle(), C.byref(a), C.byref(b)))
        return a.value, b.value

    def local_block(self):
        """(first_row, n_rows, first_col, n_cols, Pz, Px) of the node block this rank owns."""
        v = [C.c_int() for _ in range(6)]
        self.check(self.lib.pl_local_block(self.handle(), *[C.byref(x) for x in v]))
        return tuple(x.value for x in v)

    def attach_local(self, group, rank, Pz, Px):
        """Join an in-process group of virtual ranks (pl_local_group_create) as `rank` of a Pz x Px layout."""
        self.check(self.lib.pl_set_comm_local(self.handle(), group, int(rank), int(Pz), int(Px)))
        self.rank, self.nranks = int(rank), int(Pz) * int(Px)

    def close(self):
        if self.h is not None:
            self._fin()
            self.h = None

    def device_info(self):
        name = C.create_string_buffer(256)
        cu = C.c_int(); mem = C.c_size_t()
        self.check(self.lib.pl_device_info(self.h, name, 256, C.byref(cu), C.byref(mem)))
        return name.value.decode(), cu.value, mem.value


_cache = collections.OrderedDict()
_CACHE_MAX = 4


def get_context(nx, grid):
    """Context for this grid; cached so consecutive module calls share device state.  Least-recently-used
    contexts are closed once more than _CACHE_MAX are held -- but never one that still backs a live operator
    (A @ x / solve(A, ...) on it must keep working, like the reference's independent matrices)."""
    gz = _lib.f64(grid[0]); gx = _lib.f64(grid[1])
    key = (int(nx[0]), int(nx[1]), gz.tobytes(), gx.tobytes())
    ctx = _cache.get(key)
    if ctx is not None and ctx.closed:
        del _cache[key]; ctx = None
    if ctx is None:
        if len(_cache) >= _CACHE_MAX:            # keep device memory bounded
            for k in list(_cache.keys()):        # oldest first
                if len(_cache) < _CACHE_MAX:
                    break
                if len(_cache[k].operators) == 0:
                    _cache.pop(k).close()
        ctx = Context(nx, [gz, gx])
        _cache[key] = ctx
    else:
        _cache.move_to_end(key)
    return ctx


def clear_contexts():
    while _cache:
        _, c = _cache.popitem()
        c.close()
