"""3-D staggered Stokes + heat on HIP kernels (BASELINE config 5) -- PARITY UNPINNED.

The reference implements DIM = 2 only (pylamp_const.py:6; pylamp_stokes.gidx prints "NOT IMPLEMENTED" for dim != 2,
pylamp_stokes.py:30-35).  What it fixes is the intent: axis order z, x, y (pylamp_const.py:9-13), arrays (nz, nx, ny),
IP = DIM = 3 and the DOF order of the comment at pylamp_stokes.py:24.  The functions below mirror the 2-D module API
(makeStokesMatrix / x2vp / solve, makeDiffusionMatrix / x2t / solve) with those conventions; the operators extend the
2-D rows dimension by dimension (pylamp_amd/csrc/pl_3d.hip) so that a y-invariant extrusion reproduces the 2-D
operator and solution on every y-slice.
"""
import ctypes as C
import weakref

import numpy as np

from . import _lib

DIM3 = 3
IZ, IX, IY, IP3 = 0, 1, 2, 3
BC_TYPE_FREESLIP = 1
BC_TYPE_FIXTEMP = 0
BC_TYPE_FIXFLOW = 1
DEFAULT_RTOL = 1e-7        # residual bound; the solve also has to meet the velocity-error estimate 3e-8 (as pylamp_stokes.solve)
DEFAULT_MAXIT = 600


def gidx(idxs, nx):
    """iz*nx*ny*4 + ix*ny*4 + iy*4 (+ IZ / IX / IY / IP3 for the equation): the order of the comment at pylamp_stokes.py:24."""
    if len(idxs) != 3:
        raise Exception("num of idxs != dimensions")
    return idxs[IZ] * nx[IX] * nx[IY] * 4 + idxs[IX] * nx[IY] * 4 + idxs[IY] * 4


def x2vp(x, nx):
    """([vz, vx, vy], P), each (nz, nx, ny); pressure in Kcont-scaled units, ghosts retained (cf. pylamp_stokes.py:86-101)."""
    X = np.asarray(x).reshape(int(nx[0]), int(nx[1]), int(nx[2]), 4)
    return [X[..., 0], X[..., 1], X[..., 2]], X[..., 3]


def x2t(x, nx):
    return np.asarray(x).reshape(int(nx[0]), int(nx[1]), int(nx[2]))


class Context3:
    def __init__(self, nx, grid, device=0):
        lib = _lib.load()
        self.lib = lib
        self.nx = [int(v) for v in nx]
        self.grid = [np.array(g, dtype=np.float64) for g in grid]
        if [g.size for g in self.grid] != self.nx:
            raise Exception("grid arrays do not match nx")
        h = C.c_void_p()
        rc = lib.pl3_create(C.byref(h), int(device), *self.nx, *[_lib.dptr(g) for g in self.grid])
        if rc != 0:
            msg = lib.pl3_last_error(None)
            raise Exception(msg.decode() if msg else "pl3_create failed")
        self.h = h
        self._fin = weakref.finalize(self, lib.pl3_destroy, h)

    def attach_comm(self, comm_ctx, Pz, Px, Py):
        """Make this context one block of a Pz x Px x Py decomposition (right after construction).  comm_ctx: a 2-D
        pylamp_amd Context whose communicator is set (torch.distributed under torchrun, or Context.attach_local for virtual
        ranks) with Pz * Px * Py ranks; it is kept alive with this context."""
        self.check(self.lib.pl3_set_comm(self.handle(), comm_ctx.handle(), int(Pz), int(Px), int(Py)))
        self._comm_ctx = comm_ctx
        self.layout = (int(Pz), int(Px), int(Py))

    def comm_stats(self, reset=False):
        v = (C.c_int64 * 2)()
        self.check(self.lib.pl3_comm_stats(self.handle(), v, 1 if reset else 0))
        return int(v[0]), int(v[1])

    def handle(self):
        """Native handle for a library call; a closed context raises instead of handing NULL to C."""
        if self.h is None:
            raise Exception("pylamp_amd: this 3-D context has been closed")
        return self.h

    def check(self, rc):
        if self.h is None:
            raise Exception("pylamp_amd: this 3-D context has been closed")
        if rc != 0:
            msg = self.lib.pl3_last_error(self.h)
            raise Exception(msg.decode() if msg else "libpylamp_hip error %d" % rc)

    def close(self):
        if self.h is not None:
            self._fin()
            self.h = None


def _f3(a, shp):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if a.shape != tuple(shp):
        raise Exception("field shape does not match nx")
    return a


class StokesOperator3:
    def __init__(self, ctx):
        self._ctx = ctx
        n = 4 * int(np.prod(ctx.nx))
        self.shape = (n, n)
        self.dtype = np.dtype(np.float64)
        self.last_stats = None
        kc = C.c_double(); kb = C.c_double()
        ctx.check(ctx.lib.pl3_stokes_get_scaling(ctx.handle(), C.byref(kc), C.byref(kb)))
        self.Kcont, self.Kbond = kc.value, kb.value

    def matvec(self, x):
        x = _lib.f64(x).reshape(-1)
        if x.size != self.shape[0]:
            raise Exception("dimension mismatch")
        y = np.empty_like(x)
        self._ctx.check(self._ctx.lib.pl3_stokes_apply(self._ctx.handle(), _lib.dptr(x), _lib.dptr(y)))
        return y

    dot = matvec

    def __matmul__(self, x):
        return self.matvec(x)

    def rhs(self):
        r = np.empty(self.shape[0])
        self._ctx.check(self._ctx.lib.pl3_stokes_rhs(self._ctx.handle(), _lib.dptr(r)))
        return r

    def apply_bench(self, reps=20, scaled=True):
        ms = C.c_double()
        self._ctx.check(self._ctx.lib.pl3_stokes_apply_bench(self._ctx.handle(), 1 if scaled else 0, int(reps), C.byref(ms)))
        return ms.value

    def mg_info(self):
        n = C.c_int(); lm = (C.c_double * 32)()
        self._ctx.check(self._ctx.lib.pl3_stokes_mg_info(self._ctx.handle(), C.byref(n), lm, 32))
        return n.value, [lm[k] for k in range(n.value)]


def makeStokesMatrix(nx, grid, f_etas, f_etan, f_rho, bc=None, grav=None, device=0, ctx=None, strict_reference=True):
    """3-D counterpart of pylamp_stokes.makeStokesMatrix: f_etas at the NODES (averaged onto the edges by the kernels),
    f_etan at the cell centres, f_rho at the nodes; all walls free-slip (bc, if given, must say so).
    strict_reference=True keeps the reference's wall rows (outermost in-domain tangential velocities slaved to their
    inner neighbours: free slip imposed half a cell inside the wall, first-order accurate); False uses natural mirror
    rows (second-order accurate)."""
    if bc is not None and any(int(b) != BC_TYPE_FREESLIP for b in bc):
        raise Exception("3-D Stokes: all walls are free-slip")
    ctx = ctx or Context3(nx, grid, device)
    shp = ctx.nx
    es, en, rho = _f3(f_etas, shp), _f3(f_etan, shp), _f3(f_rho, shp)
    g = None if grav is None else (C.c_double * 3)(*[float(v) for v in grav])
    ctx.check(ctx.lib.pl3_stokes_set_coeffs(ctx.handle(), _lib.dptr(es), _lib.dptr(en), _lib.dptr(rho), g))
    ctx.check(ctx.lib.pl3_stokes_set_wall_rows(ctx.handle(), 1 if strict_reference else 0))
    A = StokesOperator3(ctx)
    return A, A.rhs()


def solve(A, rhs=None, x0=None, rtol=DEFAULT_RTOL, maxit=DEFAULT_MAXIT, resident=False, warm=False):
    """x = A^-1 rhs (rhs None: the operator's own right-hand side): multigrid-preconditioned BiCGStab on the GPU.
    resident=True: nothing crosses PCIe -- the operator's own right-hand side, the solution stays on the device (returns None;
    solution(A) fetches it), warm=True starts from the previous resident solution."""
    ctx = A._ctx
    if resident:
        if rhs is not None or x0 is not None:
            raise Exception("solve(resident=True) works on the operator's own right-hand side and the resident solution")
        st = _lib.SolveStats()
        ctx.check(ctx.lib.pl3_stokes_solve(ctx.handle(), None, None, 1 if warm else 0, float(rtol), int(maxit), C.byref(st)))
        A.last_stats = st.as_dict()
        return None
    x = np.zeros(A.shape[0]) if x0 is None else _lib.f64(x0).reshape(-1).copy()
    st = _lib.SolveStats()
    r = None if rhs is None else _lib.dptr(_lib.f64(rhs).reshape(-1))
    ctx.check(ctx.lib.pl3_stokes_solve(ctx.handle(), r, _lib.dptr(x), 0 if x0 is None else 1, float(rtol), int(maxit), C.byref(st)))
    A.last_stats = st.as_dict()
    return x


class HeatOperator3:
    def __init__(self, ctx):
        self._ctx = ctx
        n = int(np.prod(ctx.nx))
        self.shape = (n, n)
        self.last_stats = None

    def matvec(self, x):
        x = _lib.f64(x).reshape(-1)
        y = np.empty_like(x)
        self._ctx.check(self._ctx.lib.pl3_heat_apply(self._ctx.handle(), _lib.dptr(x), _lib.dptr(y)))
        return y

    def __matmul__(self, x):
        return self.matvec(x)

    def rhs(self):
        r = np.empty(self.shape[0])
        self._ctx.check(self._ctx.lib.pl3_heat_rhs(self._ctx.handle(), _lib.dptr(r)))
        return r


def makeDiffusionMatrix(nx, grid, gridmp, f_T, f_k, f_Cp, f_rho, f_H, bc, bcvalue, tstep, device=0, ctx=None):
    """3-D counterpart of pylamp_diff.makeDiffusionMatrix: f_k = [kz, kx, ky] on the faces normal to z, x, y;
    bc / bcvalue = [z0, x0, y0, zL, xL, yL]."""
    ctx = ctx or Context3(nx, grid, device)
    shp = ctx.nx
    arrs = [_f3(a, shp) for a in (f_T, f_k[0], f_k[1], f_k[2], f_Cp, f_rho, f_H)]
    mp = [np.ascontiguousarray(m, dtype=np.float64) for m in gridmp]
    bc_arr = (C.c_int * 6)(*[int(b) for b in bc]); bv = (C.c_double * 6)(*[float(b) for b in bcvalue])
    ctx.check(ctx.lib.pl3_heat_set_coeffs(ctx.handle(), *[_lib.dptr(m) for m in mp], *[_lib.dptr(a) for a in arrs], bc_arr, bv, float(tstep)))
    A = HeatOperator3(ctx)
    return A, A.rhs()


def solution(A, heat=False):
    """The device-resident solution of the last solve(A, resident=True) / solve_heat(A, resident=True)."""
    ctx = A._ctx
    x = np.zeros(A.shape[0])
    ctx.check(ctx.lib.pl3_get_solution(ctx.handle(), 1 if heat else 0, _lib.dptr(x)))
    return x


def solve_heat(A, rtol=1e-12, maxit=2000, resident=False):
    ctx = A._ctx
    if resident:
        st = _lib.SolveStats()
        ctx.check(ctx.lib.pl3_heat_solve(ctx.handle(), None, None, float(rtol), int(maxit), C.byref(st)))
        A.last_stats = st.as_dict()
        return None
    x = np.zeros(A.shape[0])
    st = _lib.SolveStats()
    ctx.check(ctx.lib.pl3_heat_solve(ctx.handle(), None, _lib.dptr(x), float(rtol), int(maxit), C.byref(st)))
    A.last_stats = st.as_dict()
    return x


class VirtualCluster3:
    """Pz x Px x Py virtual ranks in ONE process on one GPU, each a Context3 block with its own host thread, joined by the
    library's in-process transport (as driver.VirtualCluster for the 2-D step): the rehearsal of BASELINE config 5 on several
    GPUs -- the same halo pack / unpack kernels, block-wise multigrid levels and all-reduced dot products, only the wire is a
    device-to-device copy."""

    def __init__(self, nx, grid, Pz, Px, Py, device=0):
        import concurrent.futures
        from ._context import Context
        lib = _lib.load()
        self.lib = lib
        self.size = int(Pz) * int(Px) * int(Py)
        g = C.c_void_p()
        if lib.pl_local_group_create(C.byref(g), self.size) != 0:
            raise Exception("pl_local_group_create failed")
        self.group = g
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=self.size)
        ncol = 16 * self.size + 1                      # the carrier context only carries the transport: any grid its 1 x size layout divides
        tiny = [np.linspace(0, 1, 17), np.linspace(0, 1, ncol)]
        self.comms, self.ctxs = [], []
        for r in range(self.size):
            c2 = Context([17, ncol], tiny, device=device, attach_dist=False)
            c2.attach_local(g, r, 1, self.size)
            c3 = Context3(nx, grid, device)
            c3.attach_comm(c2, Pz, Px, Py)
            self.comms.append(c2); self.ctxs.append(c3)

    def all(self, fn, timeout=1800):
        """fn(ctx3, rank) on every rank at once (the calls are collective); results in rank order."""
        import concurrent.futures as cf
        futs = [self.pool.submit(fn, c, r) for r, c in enumerate(self.ctxs)]
        done, pending = cf.wait(futs, timeout=timeout, return_when=cf.FIRST_EXCEPTION)
        if pending:
            self.lib.pl_local_group_abort(self.group)
            cf.wait(futs, timeout=60)
        errs = [(r, f.exception()) for r, f in enumerate(futs) if f.done() and f.exception() is not None]
        if errs:       # every rank's message: the first one in rank order is often only the consequence of another rank's failure
            raise Exception("virtual ranks failed: " + " | ".join("rank %d: %s" % (r, e) for r, e in errs)) from errs[0][1]
        return [f.result(timeout=1) for f in futs]

    def close(self):
        for c in self.ctxs:
            c.close()
        for c in self.comms:
            c.close()
        self.pool.shutdown(wait=False)
        self.lib.pl_local_group_destroy(self.group)
