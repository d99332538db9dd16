"""Drop-in for the reference's pylamp_stokes.py, backed by HIP kernels.

Same names and signatures as pylamp_stokes.py:17-20 (BC constants), 22-35 (gidx),
86-101 (x2vp), 104-563 (makeStokesMatrix).  makeStokesMatrix returns (A, rhs) where A is a
matrix-free StokesOperator living on the GPU instead of a lil_matrix; `solve(A, rhs)`
replaces the driver's spsolve(csc_matrix(A), rhs) (pylamp2.py:360,394).  The DOF order of
x / rhs is the reference's, so x2vp is unchanged.
"""
import ctypes as C

import numpy as np

from .pylamp_const import *  # noqa: F401,F403
from .pylamp_const import DIM, IZ, IX
from . import _lib
from ._context import get_context

BC_TYPE_NOSLIP = 0
BC_TYPE_FREESLIP = 1
BC_TYPE_CYCLIC = 2
BC_TYPE_FLOWTHRU = 4

# default stopping rule of solve(): true-residual ||b-Ax||/||b||
DEFAULT_RTOL = 1e-7        # residual bound; the solve also has to meet the velocity-error estimate 3e-8 (include/pylamp_hip.h, pl_stokes_solve)
DEFAULT_MAXIT = 400


def gidx(idxs, nx, dim):
    """Global matrix index of node(s) idxs=[i, j]; add IZ/IX/IP for the equation
    (pylamp_stokes.py:22-35)."""
    if len(idxs) != dim:
        raise Exception("num of idxs != dimensions")
    if dim == 2:
        ret = idxs[IZ] * nx[IX] * (dim + 1) + idxs[IX] * (dim + 1)
    else:
        print("!!! NOT IMPLEMENTED")
        raise Exception("gidx: only 2D is implemented")
    return ret


def x2vp(x, nx):
    """Split a solution vector into ([vz, vx], P), ghosts retained, pressure left in
    Kcont-scaled units (pylamp_stokes.py:86-101)."""
    dof = int(np.prod(nx)) * (DIM + 1)
    x = np.asarray(x)
    newvel = [[]] * DIM
    for d in range(DIM):
        newvel[d] = x[d:dof:DIM + 1].reshape(nx)
    newpres = x[DIM:dof:DIM + 1].reshape(nx)
    return (newvel, newpres)


class StokesOperator:
    """Matrix-free A of the Stokes system, resident on the GPU.

    Offers what the reference's callers use of the sparse matrix: .shape, A @ x / .dot /
    .matvec, and .tocsc() (materialised by 27 coloured probe applies; for tests).
    """

    def __init__(self, ctx, nx, coeffs):
        self._ctx = ctx
        self.nx = [int(nx[0]), int(nx[1])]
        n = 3 * self.nx[0] * self.nx[1]
        self.shape = (n, n)
        self.dtype = np.dtype(np.float64)
        self.last_stats = None
        # The operator owns host copies of its coefficients: the device holds ONE coefficient set per context, so
        # a later makeStokesMatrix on the same grid replaces it; this operator then uploads its own again before
        # it is applied or solved (the reference returns independent matrices).
        self._coeffs = coeffs
        self._gen = -1
        self._activate()
        kc = C.c_double(); kb = C.c_double()
        ctx.check(ctx.lib.pl_stokes_get_scaling(ctx.handle(), C.byref(kc), C.byref(kb)))
        self.Kcont, self.Kbond = kc.value, kb.value
        ctx.operators.add(self)

    def _activate(self):
        ctx = self._ctx
        if self._gen == ctx.stokes_gen and self._gen >= 0:
            return
        es, en, rho, bc, surfstab, tstep, theta = self._coeffs
        bc_arr = (C.c_int * 4)(*bc)
        ctx.check(ctx.lib.pl_stokes_set_coeffs(ctx.handle(), _lib.dptr(es), _lib.dptr(en), _lib.dptr(rho), bc_arr,
                                               surfstab, tstep, theta))
        ctx.stokes_gen += 1
        self._gen = ctx.stokes_gen

    def matvec(self, x):
        x = _lib.f64(x).reshape(-1)
        if x.size != self.shape[0]:
            raise Exception("dimension mismatch")
        self._activate()
        y = np.empty_like(x)
        self._ctx.check(self._ctx.lib.pl_stokes_apply(self._ctx.handle(), _lib.dptr(x), _lib.dptr(y)))
        return y

    dot = matvec

    def __matmul__(self, x):
        return self.matvec(x)

    def rhs(self):
        self._activate()
        r = np.empty(self.shape[0])
        self._ctx.check(self._ctx.lib.pl_stokes_rhs(self._ctx.handle(), _lib.dptr(r)))
        return r

    def precond(self, r):
        """z = M^-1 r of the solver's preconditioner (diagnostic; r unscaled)."""
        r = _lib.f64(r).reshape(-1)
        z = np.empty_like(r)
        self._activate()
        self._ctx.check(self._ctx.lib.pl_stokes_precond_apply(self._ctx.handle(), _lib.dptr(r), _lib.dptr(z)))
        return z

    def mg_info(self):
        n = C.c_int(); lm = (C.c_double * 32)()
        self._ctx.check(self._ctx.lib.pl_stokes_mg_info(self._ctx.handle(), C.byref(n), lm, 32))
        return n.value, [lm[k] for k in range(n.value)]

    def mg_precision(self):
        """(levels, levels in FP32) of the multigrid hierarchy of the last solve."""
        n = C.c_int(); nf = C.c_int()
        self._ctx.check(self._ctx.lib.pl_stokes_mg_precision(self._ctx.handle(), C.byref(n), C.byref(nf)))
        return n.value, nf.value

    def set_mg_precision(self, fp32, min_nodes=0):
        """True: the large multigrid levels store and sweep in FP32 (default: all FP64; BiCGStab is FP64 either way).
        min_nodes > 0: smallest level that runs in FP32 (default 200000 nodes)."""
        self._ctx.check(self._ctx.lib.pl_stokes_set_mg_precision(self._ctx.handle(), int(bool(fp32)), int(min_nodes)))

    def tocsc(self):
        """Explicit scipy CSC equal to the reference's matrix (probing: every row reaches
        only nodes within +-1 in i and j, so 3x3x3 colours separate all its entries)."""
        import scipy.sparse as sp
        nz, nxx = self.nx
        ii, jj = np.meshgrid(np.arange(nz), np.arange(nxx), indexing='ij')
        rows, cols, vals = [], [], []
        node = (ii * nxx + jj)
        for ci in range(3):
            for cj in range(3):
                mask = ((ii % 3) == ci) & ((jj % 3) == cj)
                for q in range(3):
                    e = np.zeros((nz, nxx, 3))
                    e[mask, q] = 1.0
                    y = self.matvec(e.reshape(-1)).reshape(nz, nxx, 3)
                    ri, rj, rq = np.nonzero(y)
                    # the probed column seen by row (ri,rj): the node of colour (ci,cj) within +-1
                    pi = ri + ((ci - ri % 3 + 1) % 3 - 1)
                    pj = rj + ((cj - rj % 3 + 1) % 3 - 1)
                    ok = (pi >= 0) & (pi < nz) & (pj >= 0) & (pj < nxx)
                    rows.append((node[ri, rj] * 3 + rq)[ok])
                    cols.append((pi * nxx + pj)[ok] * 3 + q)
                    vals.append(y[ri, rj, rq][ok])
        A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))), shape=self.shape)
        return A.tocsc()


def makeStokesMatrix(nx, grid, f_etas, f_etan, f_rho, bc, surfstab=False, tstep=None, surfstab_theta=0.5,
                     strict_reference=True):
    """Set up the matrix-free Stokes operator and its rhs (pylamp_stokes.py:104-563).

    strict_reference=False flips the sign of the free-surface stabilisation terms: as written in the
    reference (pylamp_stokes.py:422-426,483-487) they are ADDED to the negative diagonal, which amplifies
    interface velocities instead of damping them; the flipped sign is the
    stabilisation of Duretz et al. (2011).  It has no effect when surfstab is off."""
    ctx = get_context(nx, grid)
    if surfstab and tstep is None:
        raise Exception("surface stabilization needs predetermined tstep")
    es, en, rho = (np.array(a, dtype=np.float64, order="C") for a in (f_etas, f_etan, f_rho))    # own copies
    for a in (es, en, rho):
        if a.shape != (int(nx[0]), int(nx[1])):
            raise Exception("field shape does not match nx")
    coeffs = (es, en, rho, [int(b) for b in bc], 1 if surfstab else 0, float(tstep) if tstep is not None else 0.0,
              float(surfstab_theta) if strict_reference else -float(surfstab_theta))
    A = StokesOperator(ctx, nx, coeffs)
    return (A, A.rhs())


def solve(A, rhs, x0=None, rtol=DEFAULT_RTOL, maxit=DEFAULT_MAXIT):
    """x = A^-1 rhs on the GPU; stands in for spsolve(csc_matrix(A), rhs) (pylamp2.py:360)."""
    ctx = A._ctx
    A._activate()
    rhs = _lib.f64(rhs).reshape(-1)
    x = np.zeros_like(rhs) if x0 is None else _lib.f64(x0).reshape(-1).copy()
    st = _lib.SolveStats()
    ctx.check(ctx.lib.pl_stokes_solve(ctx.handle(), _lib.dptr(rhs), _lib.dptr(x), 0 if x0 is None else 1, float(rtol),
                                      int(maxit), C.byref(st)))
    A.last_stats = st.as_dict()
    return x
