"""Drop-in for the reference's pylamp_diff.py (pylamp_diff.py:12-13,15-28,78-83,85-183)."""
import ctypes as C

import numpy as np

from .pylamp_const import *  # noqa: F401,F403
from .pylamp_const import IZ, IX
from . import _lib
from ._context import get_context

BC_TYPE_FIXTEMP = 0
BC_TYPE_FIXFLOW = 1

DEFAULT_RTOL = 1e-12
DEFAULT_MAXIT = 2000


def gidx(idxs, nx):
    """Row index of node(s) [i, j] (pylamp_diff.py:15-28)."""
    if len(idxs) != 2:
        raise Exception("num of idxs != dimensions")
    return idxs[IZ] * nx[IX] + idxs[IX]


def x2t(x, nx):
    """pylamp_diff.py:78-83."""
    return np.asarray(x).reshape(nx)


class HeatOperator:
    def __init__(self, ctx, nx, coeffs):
        self._ctx = ctx
        self.nx = [int(nx[0]), int(nx[1])]
        n = self.nx[0] * self.nx[1]
        self.shape = (n, n)
        self.dtype = np.dtype(np.float64)
        self.last_stats = None
        self._coeffs = coeffs          # host copies: re-uploaded when another operator took the context's device slot
        self._gen = -1
        self._activate()
        ctx.operators.add(self)

    def _activate(self):
        ctx = self._ctx
        if self._gen == ctx.heat_gen and self._gen >= 0:
            return
        zm, xm, arrs, bc, bcvalue, tstep = self._coeffs
        bc_arr = (C.c_int * 4)(*bc)
        bv = (C.c_double * 4)(*bcvalue)
        ctx.check(ctx.lib.pl_heat_set_coeffs(ctx.handle(), _lib.dptr(zm), _lib.dptr(xm), *[_lib.dptr(a) for a in arrs],
                                             bc_arr, bv, tstep))
        ctx.heat_gen += 1
        self._gen = ctx.heat_gen

    def matvec(self, x):
        x = _lib.f64(x).reshape(-1)
        if x.size != self.shape[0]:
            raise Exception("dimension mismatch")
        self._activate()
        y = np.empty_like(x)
        self._ctx.check(self._ctx.lib.pl_heat_apply(self._ctx.handle(), _lib.dptr(x), _lib.dptr(y)))
        return y

    dot = matvec

    def __matmul__(self, x):
        return self.matvec(x)

    def rhs(self):
        self._activate()
        r = np.empty(self.shape[0])
        self._ctx.check(self._ctx.lib.pl_heat_rhs(self._ctx.handle(), _lib.dptr(r)))
        return r

    def tocsc(self):
        import scipy.sparse as sp
        nz, nxx = self.nx
        ii, jj = np.meshgrid(np.arange(nz), np.arange(nxx), indexing='ij')
        rows, cols, vals = [], [], []
        for ci in range(3):
            for cj in range(3):
                e = (((ii % 3) == ci) & ((jj % 3) == cj)).astype(np.float64)
                y = self.matvec(e.reshape(-1)).reshape(nz, nxx)
                ri, rj = np.nonzero(y)
                pi = ri + ((ci - ri % 3 + 1) % 3 - 1)
                pj = rj + ((cj - rj % 3 + 1) % 3 - 1)
                ok = (pi >= 0) & (pi < nz) & (pj >= 0) & (pj < nxx)
                rows.append((ri * nxx + rj)[ok]); cols.append((pi * nxx + pj)[ok]); vals.append(y[ri, rj][ok])
        return sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                             shape=self.shape).tocsc()


def makeDiffusionMatrix(nx, grid, gridmp, f_T, f_k, f_Cp, f_rho, f_H, bc, bcvalue, tstep):
    """Set up the matrix-free heat operator and rhs (pylamp_diff.py:85-183)."""
    ctx = get_context(nx, grid)
    shp = (int(nx[0]), int(nx[1]))
    arrs = [np.array(a, dtype=np.float64, order="C") for a in (f_T, f_k[IZ], f_k[IX], f_Cp, f_rho, f_H)]   # own copies
    for a in arrs:
        if a.shape != shp:
            raise Exception("field shape does not match nx")
    zm, xm = np.array(gridmp[IZ], dtype=np.float64), np.array(gridmp[IX], dtype=np.float64)
    coeffs = (zm, xm, arrs, [int(b) for b in bc], [float(b) for b in bcvalue], float(tstep))
    A = HeatOperator(ctx, nx, coeffs)
    return (A, A.rhs())


def solve(A, rhs, rtol=DEFAULT_RTOL, maxit=DEFAULT_MAXIT):
    """x = A^-1 rhs on the GPU; stands in for spsolve at pylamp2.py:419."""
    ctx = A._ctx
    A._activate()
    rhs = _lib.f64(rhs).reshape(-1)
    x = np.zeros_like(rhs)
    st = _lib.SolveStats()
    ctx.check(ctx.lib.pl_heat_solve(ctx.handle(), _lib.dptr(rhs), _lib.dptr(x), float(rtol), int(maxit), C.byref(st)))
    A.last_stats = st.as_dict()
    return x
