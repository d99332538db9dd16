"""Drop-in for the reference's pylamp_trac.py: trac2grid, grid2trac, RK
(pylamp_trac.py:11-22 constants, 30-158, 161-318, 321-388) on HIP kernels."""
import ctypes as C

import numpy as np

from .pylamp_const import *  # noqa: F401,F403
from .pylamp_const import DIM, IZ, IX
from . import _lib
from ._context import get_context

INTERP_AVG_ARITHMETIC = 1
INTERP_AVG_GEOMETRIC = 2
INTERP_AVG_WEIGHTED = 4
INTERP_AVG_ARITHW = INTERP_AVG_ARITHMETIC + INTERP_AVG_WEIGHTED
INTERP_AVG_GEOMW = INTERP_AVG_GEOMETRIC + INTERP_AVG_WEIGHTED

INTERP_METHOD_IDW = 1
INTERP_METHOD_GRIDDATA = 2
INTERP_METHOD_ELEM = 4
INTERP_METHOD_NEAREST = 8
INTERP_METHOD_LINEAR = 16
INTERP_METHOD_VELDIV = 32

_MAXF = 8


def _is_uniform(c):
    """True for a regular coordinate array (the only kind the reference's marker code handles)."""
    c = np.asarray(c, dtype=np.float64)
    if c.size < 3:
        return True
    d = np.diff(c)
    return bool(np.all(np.abs(d - d.mean()) <= 1e-9 * abs(d.mean())))


def _ctx_for(nz, nxx):
    # MIC kernels need a device context only for memory/stream; key it by the node counts
    return get_context([nz, nxx], [np.arange(nz, dtype=np.float64), np.arange(nxx, dtype=np.float64)])


def grid2trac(tr_x, tr_f, grid, gridfield, nx, defval=np.nan, method=INTERP_METHOD_LINEAR, stopOnError=False):
    """Interpolate gridfield (list of 2-D arrays) to tracer positions, writing tr_f in
    place (pylamp_trac.py:30-158).  On a non-uniform grid -- which the reference does not support: it
    picks the cell with the regular-grid formula -- cells are located by per-axis search."""
    assert len(gridfield) == tr_f.shape[1]
    assert tr_x.shape[0] == tr_f.shape[0] and tr_x.shape[1] == 2      # the reference fails on broadcasting here
    assert (method & INTERP_METHOD_LINEAR) or (method & INTERP_METHOD_NEAREST) or (method & INTERP_METHOD_VELDIV)
    nf = len(gridfield)
    gnz, gnx = int(nx[IZ]), int(nx[IX])
    ctx = _ctx_for(max(gnz, 5), max(gnx, 5))
    n = tr_x.shape[0]
    txc = _lib.f64(tr_x)
    gz, gx = _lib.f64(grid[IZ]), _lib.f64(grid[IX])
    out = np.empty((n, nf))
    nout = C.c_int64(0)
    ctx.check(ctx.lib.pl_mic_set_search(ctx.handle(), 0 if (_is_uniform(gz) and _is_uniform(gx)) else 1))
    for k0 in range(0, nf, _MAXF):
        k1 = min(nf, k0 + _MAXF)
        fl = [_lib.f64(gridfield[k]) for k in range(k0, k1)]
        fp = (_lib.c_double_p * (k1 - k0))(*[_lib.dptr(a) for a in fl])
        sub = np.empty((n, k1 - k0))
        ctx.check(ctx.lib.pl_grid2trac(ctx.handle(), n, _lib.dptr(txc), k1 - k0, fp, gnz, gnx, _lib.dptr(gz),
                                       _lib.dptr(gx), int(method), float(defval), 1 if stopOnError else 0,
                                       _lib.dptr(sub), k1 - k0, C.byref(nout)))
        out[:, k0:k1] = sub
    if nout.value > 0:
        print("!!! Warning, grid2trac(): Using default value for extrapolation in ", nout.value, "tracers")
    tr_f[:, :] = out        # in-place write, also into strided views (pylamp2.py:445)
    return


def trac2grid(tr_x, tr_f, mesh, grid, gridfield, nx, distweight=None, avgscheme=None, method=INTERP_METHOD_ELEM,
              debug=False):
    """Average tracer values onto the node set `grid`, writing gridfield[k][:, :] in place
    (pylamp_trac.py:161-318, method ELEM).  `mesh`, `distweight`, `debug` are accepted for
    signature compatibility (the reference uses mesh only for its shape)."""
    assert len(gridfield) == tr_f.shape[1]
    assert tr_x.shape[0] == tr_f.shape[0] and tr_x.shape[1] == 2      # np.add.at raises on the mismatch in the reference
    if avgscheme is None:
        avgscheme = [INTERP_AVG_ARITHMETIC + INTERP_AVG_WEIGHTED for i in range(len(gridfield))]
    assert type(avgscheme) == type([])
    assert len(avgscheme) == len(gridfield)
    if not (method & INTERP_METHOD_ELEM):
        raise Exception("trac2grid: only INTERP_METHOD_ELEM is implemented on the GPU")
    nz, nxx = int(nx[IZ]), int(nx[IX])
    ctx = _ctx_for(nz, nxx)
    n = tr_x.shape[0]
    txc = _lib.f64(tr_x)
    gz, gx = np.asarray(grid[IZ], dtype=np.float64), np.asarray(grid[IX], dtype=np.float64)
    z0, x0 = float(gz[0]), float(gx[0])
    hz = float(gz[-1] - gz[0]) / (nz - 1)
    hx = float(gx[-1] - gx[0]) / (nxx - 1)
    rect = not (_is_uniform(gz) and _is_uniform(gx))       # beyond the reference: per-axis search
    if rect:
        gz, gx = np.ascontiguousarray(gz), np.ascontiguousarray(gx)
        assert gz.size == nz and gx.size == nxx
    nf = len(gridfield)
    for k0 in range(0, nf, _MAXF):
        k1 = min(nf, k0 + _MAXF)
        sub = _lib.f64(tr_f[:, k0:k1])
        outs = [np.empty((nz, nxx)) for _ in range(k1 - k0)]
        op = (_lib.c_double_p * (k1 - k0))(*[_lib.dptr(a) for a in outs])
        sch = (C.c_int * (k1 - k0))(*[int(s) for s in avgscheme[k0:k1]])
        if rect:
            ctx.check(ctx.lib.pl_trac2grid_rect(ctx.handle(), n, _lib.dptr(txc), _lib.dptr(sub), k1 - k0, k1 - k0, sch,
                                                _lib.dptr(gz), _lib.dptr(gx), op))
        else:
            ctx.check(ctx.lib.pl_trac2grid(ctx.handle(), n, _lib.dptr(txc), _lib.dptr(sub), k1 - k0, k1 - k0, sch, z0, hz,
                                           x0, hx, op))
        for k in range(k0, k1):
            gridfield[k][:, :] = outs[k - k0]
    return


def RK(tr_x, grids, vels, nx, tstep, order=4):
    """Runge-Kutta advection of tracers; returns (vel_final, tr_x_final)
    (pylamp_trac.py:321-388).  grids/vels live on the padded (nz+1, nx+1) cell-centre grid."""
    if order != 2 and order != 4:
        raise Exception("Sorry, don't know how to do that")
    if len(nx) != 2:
        raise Exception("Sorry, only 2D supported at the moment")
    if order == 2:
        # the reference's order=2 branch uses undefined names (pylamp_trac.py:332-345) and cannot run
        raise Exception("RK order 2 is not functional in the reference; use order=4")
    gnz, gnx = int(nx[IZ]) + 1, int(nx[IX]) + 1
    ctx = _ctx_for(gnz, gnx)
    n = tr_x.shape[0]
    txc = _lib.f64(tr_x)
    gz, gx = _lib.f64(grids[IZ]), _lib.f64(grids[IX])
    vz, vx = _lib.f64(vels[IZ]), _lib.f64(vels[IX])
    if vz.shape != (gnz, gnx) or vx.shape != (gnz, gnx) or gz.size != gnz or gx.size != gnx:
        raise Exception("RK: velocity grids must have shape (nz+1, nx+1)")
    v = np.empty((n, DIM)); xn = np.empty((n, DIM))
    ctx.check(ctx.lib.pl_mic_set_search(ctx.handle(), 0 if (_is_uniform(gz) and _is_uniform(gx)) else 1))
    ctx.check(ctx.lib.pl_rk4(ctx.handle(), n, _lib.dptr(txc), gnz, gnx, _lib.dptr(gz), _lib.dptr(gx), _lib.dptr(vz),
                             _lib.dptr(vx), float(tstep), _lib.dptr(v), _lib.dptr(xn)))
    return v, xn
