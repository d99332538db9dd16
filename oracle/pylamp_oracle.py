"""CPU oracle for the PyLamp per-time-step hot path.  TEST INFRASTRUCTURE ONLY.

This module is a NumPy/SciPy restatement of the reference algorithm
(larskaislaniemi/PyLamp: pylamp_stokes.py, pylamp_diff.py, pylamp_trac.py and the
time-loop body of pylamp2.py).  It is the *checker* for the HIP path:

  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it;
  * the product package (pylamp_amd/) never imports it and never falls back to it.

Parity status: PINNED.  tests/test_oracle_golden.py checks every function below against
fixtures under tests/golden/ that were produced by running the reference's own modules
(oracle/gen_golden.py, run once in the build container where /root/reference exists).

Everything here was written from the discretisation, not copied:
the reference assembles a lil_matrix by fancy-index assignment; this file builds COO
triplets per *row class* and also offers an independent matrix-free apply.

Array conventions (reference: pylamp_const.py:6-18, pylamp2.py:37,100-113):
  nx = [nz, nx_] node counts, arrays are (nz, nx_) float64 indexed [i=z, j=x];
  Stokes DOF order is interleaved (vz, vx, P) per node, nodes row-major
  (pylamp_stokes.py:22-35) so  row(i,j,q) = (i*nx_ + j)*3 + q.
"""
import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

# ---- constants mirrored from pylamp_const.py:6-46 -------------------------------------
DIM = 2
IZ, IX = 0, 1
IP = DIM
G = [9.81, 0.0]
SECINYR = 60 * 60 * 24 * 365.25
SECINKYR = SECINYR * 1e3
SECINMYR = SECINYR * 1e6
GASR = 8.31446
NFTRAC = 13
(TR_RHO, TR_ETA, TR_MRK, TR_TMP, TR_HCD, TR_HCP, TR_RH0, TR_ALP, TR_MAT, TR_ACE, TR_ET0,
 TR_IHT, TR__ID) = range(13)
EPS = 2.0 ** (-10)

# pylamp_stokes.py:17-20
BC_NOSLIP, BC_FREESLIP, BC_CYCLIC, BC_FLOWTHRU = 0, 1, 2, 4
# pylamp_diff.py:12-13
BC_FIXTEMP, BC_FIXFLOW = 0, 1
# pylamp_trac.py:11-22
AVG_ARITH, AVG_GEOM, AVG_WEIGHTED = 1, 2, 4
AVG_ARITHW, AVG_GEOMW = 5, 6
M_NEAREST, M_LINEAR, M_VELDIV = 8, 16, 32


# =======================================================================================
# Stokes
# =======================================================================================
def stokes_scaling(grid, etas, etan):
    """Kcont, Kbond as in pylamp_stokes.py:116-122 (python min(); span/len(x), sic)."""
    mineta = min(np.min(etas), np.min(etan))
    avgdx = (grid[IX][-1] - grid[IX][0]) / grid[IX].shape[0]
    avgdz = (grid[IZ][-1] - grid[IZ][0]) / grid[IZ].shape[0]
    return 2 * mineta / (avgdx + avgdz), 4 * mineta / (avgdx + avgdz) ** 2


def _check_stokes_bc(bc):
    # bc = [z0, x0, zL, xL] (index DIM*wall + dir, pylamp_stokes.py:163,202,242,289).
    # Only the combinations that give a non-singular system in the reference are restated.
    for w in (0, 2):
        if bc[w] not in (BC_NOSLIP, BC_FREESLIP):
            raise Exception("oracle: z-wall BC must be NOSLIP or FREESLIP")
    for w in (1, 3):
        if bc[w] != BC_FREESLIP:
            raise Exception("oracle: x-wall BC must be FREESLIP")


def stokes_row_class(nx, bc=None):
    """Integer class map for every DOF, shape (3, nz, nx_).

    0 identity (ghost / wall normal velocity / anchor), 1 interior momentum or continuity,
    2 tangential wall row at the low wall, 3 tangential wall row at the high wall,
    4 corner pressure (left), 5 corner pressure (right).
    The reference's DEBUG `lc` counter (pylamp_stokes.py:112,555-561) proves the classes
    partition the rows.
    """
    nz, nxx = nx
    c = np.zeros((3, nz, nxx), dtype=np.int8)
    # vz
    c[0, 1:nz - 1, 1:nxx - 2] = 1
    c[0, 1:nz - 1, 0] = 2
    c[0, 1:nz - 1, nxx - 2] = 3
    # vx
    c[1, 1:nz - 2, 1:nxx - 1] = 1
    c[1, 0, 1:nxx - 1] = 2
    c[1, nz - 2, 1:nxx - 1] = 3
    # P
    c[2, 0:nz - 1, 0:nxx - 1] = 1
    for i in (0, nz - 2):
        c[2, i, 0] = 4
        c[2, i, nxx - 2] = 5
    c[2, 3, 2] = 0
    return c


def stokes_csr(nx, grid, etas, etan, rho, bc, surfstab=False, tstep=None, theta=0.5):
    """Explicit (A, rhs) equal to pylamp_stokes.makeStokesMatrix (pylamp_stokes.py:104-563).

    Returns scipy CSR and rhs in the reference DOF order.
    """
    _check_stokes_bc(bc)
    nz, nxx = int(nx[0]), int(nx[1])
    if nz < 5 or nxx < 5:
        raise Exception("oracle: grid too small")
    z = np.asarray(grid[IZ], dtype=np.float64)
    x = np.asarray(grid[IX], dtype=np.float64)
    Kc, Kb = stokes_scaling(grid, etas, etan)
    N = nz * nxx
    rows, cols, vals = [], [], []
    rhs = np.zeros(3 * N)

    def gid(i, j, q):
        return (np.asarray(i) * nxx + np.asarray(j)) * 3 + q

    def put(r, c, v):
        r = np.asarray(r).ravel()
        c = np.asarray(c).ravel()
        v = np.broadcast_to(np.asarray(v, dtype=np.float64), r.shape) if np.ndim(v) == 0 \
            else np.asarray(v, dtype=np.float64).ravel()
        rows.append(r); cols.append(c); vals.append(v)

    cls = stokes_row_class(nx)
    # --- identity rows (ghosts, wall-normal velocities, anchor): Kc * u = 0
    for q in range(3):
        ii, jj = np.nonzero(cls[q] == 0)
        put(gid(ii, jj, q), gid(ii, jj, q), Kc)

    # --- vz tangential rows at x-walls, FREESLIP (pylamp_stokes.py:249-255, 296-301)
    i = np.arange(1, nz - 1)
    put(gid(i, 0, IZ), gid(i, 0, IZ), Kc)
    put(gid(i, 0, IZ), gid(i, 1, IZ), -Kc)
    put(gid(i, nxx - 2, IZ), gid(i, nxx - 2, IZ), Kc)
    put(gid(i, nxx - 2, IZ), gid(i, nxx - 3, IZ), -Kc)

    # --- vx tangential rows at z-walls (pylamp_stokes.py:161-175, 200-214)
    j = np.arange(1, nxx - 1)
    if bc[0] == BC_FREESLIP:
        put(gid(0, j, IX), gid(0, j, IX), Kc)
        put(gid(0, j, IX), gid(1, j, IX), -Kc)
    else:
        put(gid(0, j, IX), gid(0, j, IX), Kc * (-1 / (z[2] - z[0]) - 1 / (z[1] - z[0])))
        put(gid(0, j, IX), gid(1, j, IX), Kc * (1 / (z[2] - z[0])))
    m = nz - 1
    if bc[2] == BC_FREESLIP:
        put(gid(m - 1, j, IX), gid(m - 1, j, IX), Kc)
        put(gid(m - 1, j, IX), gid(m - 2, j, IX), -Kc)
    else:
        put(gid(m - 1, j, IX), gid(m - 1, j, IX),
            Kc * (-1 / (z[m - 2] - z[m]) - 1 / (z[m - 1] - z[m])))
        put(gid(m - 1, j, IX), gid(m - 2, j, IX), Kc * (1 / (z[m - 2] - z[m])))

    # --- corner pressures (pylamp_stokes.py:358-369)
    for i0 in (0, nz - 2):
        put(gid(i0, 0, IP), gid(i0, 1, IP), Kb)
        put(gid(i0, 0, IP), gid(i0, 0, IP), -Kb)
        put(gid(i0, nxx - 2, IP), gid(i0, nxx - 3, IP), Kb)
        put(gid(i0, nxx - 2, IP), gid(i0, nxx - 2, IP), -Kb)

    # --- continuity on all other physical cells (pylamp_stokes.py:333-354, 496-518)
    ii, jj = np.nonzero(cls[2] == 1)
    r = gid(ii, jj, IP)
    dxj = x[jj + 1] - x[jj]
    dzi = z[ii + 1] - z[ii]
    put(r, gid(ii, jj + 1, IX), Kc / dxj)
    put(r, gid(ii, jj, IX), -Kc / dxj)
    put(r, gid(ii + 1, jj, IZ), Kc / dzi)
    put(r, gid(ii, jj, IZ), -Kc / dzi)

    # --- interior z-momentum (pylamp_stokes.py:376-429)
    ii, jj = np.nonzero(cls[0] == 1)
    r = gid(ii, jj, IZ)
    dz_i = z[ii + 1] - z[ii]
    dz_m = z[ii] - z[ii - 1]
    Dz = z[ii + 1] - z[ii - 1]
    dx_j = x[jj + 1] - x[jj]
    Dxp = x[jj + 2] - x[jj]
    Dxm = x[jj + 1] - x[jj - 1]
    cN = 4 * etan[ii, jj] / dz_i / Dz          # vz(i+1,j)
    cS = 4 * etan[ii - 1, jj] / dz_m / Dz      # vz(i-1,j)
    cE = 2 * etas[ii, jj + 1] / Dxp / dx_j     # vz(i,j+1)
    cW = 2 * etas[ii, jj] / Dxm / dx_j         # vz(i,j-1)
    xE = 2 * etas[ii, jj + 1] / Dz / dx_j      # cross terms on vx
    xW = 2 * etas[ii, jj] / Dz / dx_j
    diag = -(cN + cS + cE + cW)
    dvx = np.zeros_like(diag)
    if surfstab:
        if tstep is None:
            raise Exception("surface stabilization needs predetermined tstep")
        dvx = theta * tstep * G[IZ] * 0.5 * (rho[ii, jj + 1] + rho[ii + 1, jj + 1]
                                              - rho[ii, jj - 1] - rho[ii + 1, jj - 1]) / Dxm
        diag = diag + theta * tstep * G[IZ] * 0.5 * (rho[ii + 1, jj] + rho[ii + 1, jj + 1]
                                                     - rho[ii - 1, jj] - rho[ii - 1, jj + 1]) / Dz
    put(r, gid(ii, jj, IZ), diag)
    put(r, gid(ii + 1, jj, IZ), cN)
    put(r, gid(ii - 1, jj, IZ), cS)
    put(r, gid(ii, jj + 1, IZ), cE)
    put(r, gid(ii, jj - 1, IZ), cW)
    put(r, gid(ii, jj + 1, IX), xE)
    put(r, gid(ii - 1, jj + 1, IX), -xE)
    put(r, gid(ii, jj, IX), -xW + dvx)
    put(r, gid(ii - 1, jj, IX), xW)
    put(r, gid(ii, jj, IP), -2 * Kc / Dz)
    put(r, gid(ii - 1, jj, IP), 2 * Kc / Dz)
    rhs[r] = -0.5 * (rho[ii, jj] + rho[ii, jj + 1]) * G[IZ]

    # --- interior x-momentum (pylamp_stokes.py:435-490)
    ii, jj = np.nonzero(cls[1] == 1)
    r = gid(ii, jj, IX)
    dx_j = x[jj + 1] - x[jj]
    dx_m = x[jj] - x[jj - 1]
    Dx = x[jj + 1] - x[jj - 1]
    dz_i = z[ii + 1] - z[ii]
    Dzp = z[ii + 2] - z[ii]
    Dzm = z[ii + 1] - z[ii - 1]
    cE = 4 * etan[ii, jj] / dx_j / Dx
    cW = 4 * etan[ii, jj - 1] / dx_m / Dx
    cN = 2 * etas[ii + 1, jj] / Dzp / dz_i
    cS = 2 * etas[ii, jj] / Dzm / dz_i
    zN = 2 * etas[ii + 1, jj] / Dx / dz_i
    zS = 2 * etas[ii, jj] / Dx / dz_i
    diag = -(cE + cW + cN + cS)
    dvz = np.zeros_like(diag)
    if surfstab:
        diag = diag + theta * tstep * G[IX] * 0.5 * (rho[ii, jj + 1] + rho[ii + 1, jj + 1]
                                                     - rho[ii, jj - 1] - rho[ii + 1, jj - 1]) / Dx
        dvz = theta * tstep * G[IX] * 0.5 * (rho[ii + 1, jj] + rho[ii + 1, jj + 1]
                                              - rho[ii - 1, jj] - rho[ii - 1, jj + 1]) / Dzm
    put(r, gid(ii, jj, IX), diag)
    put(r, gid(ii, jj + 1, IX), cE)
    put(r, gid(ii, jj - 1, IX), cW)
    put(r, gid(ii + 1, jj, IX), cN)
    put(r, gid(ii - 1, jj, IX), cS)
    put(r, gid(ii + 1, jj, IZ), zN)
    put(r, gid(ii + 1, jj - 1, IZ), -zN)
    put(r, gid(ii, jj, IZ), -zS + dvz)
    put(r, gid(ii, jj - 1, IZ), zS)
    put(r, gid(ii, jj, IP), -2 * Kc / Dx)
    put(r, gid(ii, jj - 1, IP), 2 * Kc / Dx)
    rhs[r] = -0.5 * (rho[ii, jj] + rho[ii + 1, jj]) * G[IX]

    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(3 * N, 3 * N)).tocsr()
    return A, rhs


def stokes_apply(nx, grid, etas, etan, bc, xvec, rho=None, surfstab=False, tstep=None,
                 theta=0.5):
    """Matrix-free y = A x (independent restatement by array slicing)."""
    _check_stokes_bc(bc)
    nz, nxx = int(nx[0]), int(nx[1])
    z = np.asarray(grid[IZ], dtype=np.float64)
    x = np.asarray(grid[IX], dtype=np.float64)
    Kc, Kb = stokes_scaling(grid, etas, etan)
    X = np.asarray(xvec, dtype=np.float64).reshape(nz, nxx, 3)
    vz, vx, P = X[:, :, 0], X[:, :, 1], X[:, :, 2]
    yz = Kc * vz.copy()
    yx = Kc * vx.copy()
    yp = Kc * P.copy()
    dz = (z[1:] - z[:-1])[:, None]       # (nz-1,1)  dz[i] = z[i+1]-z[i]
    dx = (x[1:] - x[:-1])[None, :]
    Dz = (z[2:] - z[:-2])[:, None]       # Dz[i-1] = z[i+1]-z[i-1]
    Dx = (x[2:] - x[:-2])[None, :]

    # continuity on physical cells, then overwrite corners / anchor
    cont = Kc * ((vx[:-1, 1:] - vx[:-1, :-1]) / dx + (vz[1:, :-1] - vz[:-1, :-1]) / dz)
    yp[:-1, :-1] = cont
    for i0 in (0, nz - 2):
        yp[i0, 0] = Kb * (P[i0, 1] - P[i0, 0])
        yp[i0, nxx - 2] = Kb * (P[i0, nxx - 3] - P[i0, nxx - 2])
    yp[3, 2] = Kc * P[3, 2]

    # vz tangential rows
    yz[1:-1, 0] = Kc * (vz[1:-1, 0] - vz[1:-1, 1])
    yz[1:-1, nxx - 2] = Kc * (vz[1:-1, nxx - 2] - vz[1:-1, nxx - 3])
    # vx tangential rows
    if bc[0] == BC_FREESLIP:
        yx[0, 1:-1] = Kc * (vx[0, 1:-1] - vx[1, 1:-1])
    else:
        yx[0, 1:-1] = Kc * ((-1 / (z[2] - z[0]) - 1 / (z[1] - z[0])) * vx[0, 1:-1]
                            + vx[1, 1:-1] / (z[2] - z[0]))
    m = nz - 1
    if bc[2] == BC_FREESLIP:
        yx[m - 1, 1:-1] = Kc * (vx[m - 1, 1:-1] - vx[m - 2, 1:-1])
    else:
        yx[m - 1, 1:-1] = Kc * ((-1 / (z[m - 2] - z[m]) - 1 / (z[m - 1] - z[m])) * vx[m - 1, 1:-1]
                                + vx[m - 2, 1:-1] / (z[m - 2] - z[m]))

    # interior z-momentum: i in [1,nz-2], j in [1,nxx-3]
    I = slice(1, nz - 1); J = slice(1, nxx - 2)
    Ip = slice(2, nz); Im = slice(0, nz - 2)
    Jp = slice(2, nxx - 1); Jm = slice(0, nxx - 3)
    dzi = dz[1:nz - 1]; dzm = dz[0:nz - 2]; Dzi = Dz[0:nz - 2]
    dxj = dx[:, 1:nxx - 2]; Dxp = Dx[:, 1:nxx - 2]; Dxm = Dx[:, 0:nxx - 3]
    t = (4 * etan[I, J] / dzi / Dzi) * (vz[Ip, J] - vz[I, J]) \
        - (4 * etan[Im, J] / dzm / Dzi) * (vz[I, J] - vz[Im, J]) \
        + (2 * etas[I, Jp] / Dxp / dxj) * (vz[I, Jp] - vz[I, J]) \
        - (2 * etas[I, J] / Dxm / dxj) * (vz[I, J] - vz[I, Jm]) \
        + (2 * etas[I, Jp] / Dzi / dxj) * (vx[I, Jp] - vx[Im, Jp]) \
        - (2 * etas[I, J] / Dzi / dxj) * (vx[I, J] - vx[Im, J]) \
        - (2 * Kc / Dzi) * (P[I, J] - P[Im, J])
    if surfstab:
        t = t + theta * tstep * G[IZ] * 0.5 * (
            (rho[I, Jp] + rho[Ip, Jp] - rho[I, Jm] - rho[Ip, Jm]) / Dxm * vx[I, J]
            + (rho[Ip, J] + rho[Ip, Jp] - rho[Im, J] - rho[Im, Jp]) / Dzi * vz[I, J])
    yz[I, J] = t

    # interior x-momentum: i in [1,nz-3], j in [1,nxx-2]
    I = slice(1, nz - 2); J = slice(1, nxx - 1)
    Ip = slice(2, nz - 1); Im = slice(0, nz - 3)
    Jp = slice(2, nxx); Jm = slice(0, nxx - 2)
    dxj = dx[:, 1:nxx - 1]; dxm = dx[:, 0:nxx - 2]; Dxj = Dx[:, 0:nxx - 2]
    dzi = dz[1:nz - 2]; Dzp = Dz[1:nz - 2]; Dzm = Dz[0:nz - 3]
    t = (4 * etan[I, J] / dxj / Dxj) * (vx[I, Jp] - vx[I, J]) \
        - (4 * etan[I, Jm] / dxm / Dxj) * (vx[I, J] - vx[I, Jm]) \
        + (2 * etas[Ip, J] / Dzp / dzi) * (vx[Ip, J] - vx[I, J]) \
        - (2 * etas[I, J] / Dzm / dzi) * (vx[I, J] - vx[Im, J]) \
        + (2 * etas[Ip, J] / Dxj / dzi) * (vz[Ip, J] - vz[Ip, Jm]) \
        - (2 * etas[I, J] / Dxj / dzi) * (vz[I, J] - vz[I, Jm]) \
        - (2 * Kc / Dxj) * (P[I, J] - P[I, Jm])
    if surfstab:
        t = t + theta * tstep * G[IX] * 0.5 * (
            (rho[I, Jp] + rho[Ip, Jp] - rho[I, Jm] - rho[Ip, Jm]) / Dxj * vx[I, J]
            + (rho[Ip, J] + rho[Ip, Jp] - rho[Im, J] - rho[Im, Jp]) / Dzm * vz[I, J])
    yx[I, J] = t

    return np.stack([yz, yx, yp], axis=2).reshape(-1)


def stokes_rhs(nx, rho):
    """rhs of makeStokesMatrix (pylamp_stokes.py:429,490; everything else 0)."""
    nz, nxx = int(nx[0]), int(nx[1])
    r = np.zeros((nz, nxx, 3))
    r[1:nz - 1, 1:nxx - 2, 0] = -0.5 * (rho[1:nz - 1, 1:nxx - 2] + rho[1:nz - 1, 2:nxx - 1]) * G[IZ]
    r[1:nz - 2, 1:nxx - 1, 1] = -0.5 * (rho[1:nz - 2, 1:nxx - 1] + rho[2:nz - 1, 1:nxx - 1]) * G[IX]
    return r.reshape(-1)


def x2vp(xvec, nx):
    """pylamp_stokes.py:86-101."""
    X = np.asarray(xvec).reshape(int(nx[0]), int(nx[1]), 3)
    return [X[:, :, 0].copy(), X[:, :, 1].copy()], X[:, :, 2].copy()


def stokes_solve(nx, grid, etas, etan, rho, bc, **kw):
    """The reference's solve: spsolve(csc(A), rhs) (pylamp2.py:360)."""
    A, rhs = stokes_csr(nx, grid, etas, etan, rho, bc, **kw)
    return spla.spsolve(sp.csc_matrix(A), rhs)


def stokes_solve_refined(nx, grid, etas, etan, rho, bc, refinements=3, **kw):
    """The same system solved ACCURATELY: row / column equilibration, SuperLU, and iterative refinement with the residual
    evaluated in extended precision (np.longdouble).  Not what the reference does -- what its answer is measured against where
    the plain spsolve of pylamp2.py:360 is itself inexact: on the stock model 5 (viscosity contrast 1e10) the reference's
    velocities move by 5e-5 .. 1e-4 when the viscosities are perturbed by 1e-16, and lie 2.7e-5 from this solution, whose
    refinement steps change it by < 4e-10 (tests/test_oracle_golden.py::test_stock_model_reference_solution_accuracy)."""
    A, rhs = stokes_csr(nx, grid, etas, etan, rho, bc, **kw)
    A = sp.csr_matrix(A)
    dr = 1.0 / np.abs(A).max(axis=1).toarray().ravel()
    As = sp.diags(dr) @ A
    dc = 1.0 / np.abs(As).max(axis=0).toarray().ravel()
    lu = spla.splu((As @ sp.diags(dc)).tocsc())
    x = dc * lu.solve(dr * rhs)
    data = A.data.astype(np.longdouble); rl = rhs.astype(np.longdouble)
    for _ in range(refinements):
        r = rl - np.add.reduceat(data * x.astype(np.longdouble)[A.indices], A.indptr[:-1])
        x = x + dc * lu.solve((dr * r).astype(np.float64))
    return x


# =======================================================================================
# Heat
# =======================================================================================
def gridmp_of(grid):
    """Midpoint grids with one extrapolated extra entry (pylamp2.py:92-95)."""
    out = []
    for g in grid:
        m = (g[1:] + g[:-1]) / 2
        out.append(np.append(m, m[-1] + (m[-1] - m[-2])))
    return out


def heat_csr(nx, grid, gridmp, T, k, Cp, rho, H, bc, bcvalue, tstep):
    """(A, rhs) equal to pylamp_diff.makeDiffusionMatrix (pylamp_diff.py:85-183)."""
    nz, nxx = int(nx[0]), int(nx[1])
    z, x = np.asarray(grid[IZ]), np.asarray(grid[IX])
    zm, xm = np.asarray(gridmp[IZ]), np.asarray(gridmp[IX])
    kz, kx = k[IZ], k[IX]
    N = nz * nxx
    rows, cols, vals = [], [], []
    rhs = np.zeros(N)

    def gid(i, j):
        return np.asarray(i) * nxx + np.asarray(j)

    def put(r, c, v):
        r = np.asarray(r).ravel(); c = np.asarray(c).ravel()
        v = np.broadcast_to(np.asarray(v, dtype=np.float64), r.shape) if np.ndim(v) == 0 \
            else np.asarray(v, dtype=np.float64).ravel()
        rows.append(r); cols.append(c); vals.append(v)

    for b in bc:
        if b not in (BC_FIXTEMP, BC_FIXFLOW):
            raise Exception("oracle: heat BC must be FIXTEMP or FIXFLOW")
    j = np.arange(nxx)
    # z = 0 (pylamp_diff.py:99-110) and z = L (112-124): own the corners
    if bc[0] == BC_FIXTEMP:
        put(gid(0, j), gid(0, j), 1.0)
    else:
        c = kz[0, j] / (z[1] - z[0])
        put(gid(0, j), gid(1, j), c); put(gid(0, j), gid(0, j), -c)
    rhs[gid(0, j)] = bcvalue[0]
    m = nz - 1
    if bc[2] == BC_FIXTEMP:
        put(gid(m, j), gid(m, j), 1.0)
    else:
        c = kz[m - 1, j] / (z[m] - z[m - 1])
        put(gid(m, j), gid(m, j), c); put(gid(m, j), gid(m - 1, j), -c)
    rhs[gid(m, j)] = bcvalue[2]
    i = np.arange(1, nz - 1)
    if bc[1] == BC_FIXTEMP:
        put(gid(i, 0), gid(i, 0), 1.0)
    else:
        c = kx[i, 0] / (x[1] - x[0])
        put(gid(i, 0), gid(i, 1), c); put(gid(i, 0), gid(i, 0), -c)
    rhs[gid(i, 0)] = bcvalue[1]
    n = nxx - 1
    if bc[3] == BC_FIXTEMP:
        put(gid(i, n), gid(i, n), 1.0)
    else:
        c = kx[i, n - 1] / (x[n] - x[n - 1])
        put(gid(i, n), gid(i, n), c); put(gid(i, n), gid(i, n - 1), -c)
    rhs[gid(i, n)] = bcvalue[3]

    # interior (pylamp_diff.py:157-179)
    ii, jj = np.meshgrid(np.arange(1, nz - 1), np.arange(1, nxx - 1), indexing='ij')
    ii = ii.ravel(); jj = jj.ravel()
    r = gid(ii, jj)
    pre = tstep / (rho[ii, jj] * Cp[ii, jj])
    cE = pre * kx[ii, jj] / (x[jj + 1] - x[jj]) / (xm[jj] - xm[jj - 1])
    cW = pre * kx[ii, jj - 1] / (x[jj] - x[jj - 1]) / (xm[jj] - xm[jj - 1])
    cN = pre * kz[ii, jj] / (z[ii + 1] - z[ii]) / (zm[ii] - zm[ii - 1])
    cS = pre * kz[ii - 1, jj] / (z[ii] - z[ii - 1]) / (zm[ii] - zm[ii - 1])
    put(r, gid(ii, jj + 1), cE)
    put(r, gid(ii, jj - 1), cW)
    put(r, gid(ii + 1, jj), cN)
    put(r, gid(ii - 1, jj), cS)
    put(r, r, -(cE + cW + cN + cS) - 1)
    rhs[r] = -T[ii, jj] - tstep * H[ii, jj] / (rho[ii, jj] * Cp[ii, jj])
    A = sp.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                      shape=(N, N)).tocsr()
    return A, rhs


def heat_apply(nx, grid, gridmp, k, Cp, rho, bc, tstep, Tvec):
    """Matrix-free y = A T."""
    nz, nxx = int(nx[0]), int(nx[1])
    z, x = np.asarray(grid[IZ]), np.asarray(grid[IX])
    zm, xm = np.asarray(gridmp[IZ]), np.asarray(gridmp[IX])
    kz, kx = k[IZ], k[IX]
    T = np.asarray(Tvec, dtype=np.float64).reshape(nz, nxx)
    y = T.copy()
    dz = (z[1:] - z[:-1])[:, None]; dx = (x[1:] - x[:-1])[None, :]
    dzb = (zm[1:] - zm[:-1])[:, None]; dxb = (xm[1:] - xm[:-1])[None, :]
    I = slice(1, nz - 1); J = slice(1, nxx - 1)
    c = tstep / (rho[I, J] * Cp[I, J])
    fx = (kx[I, 1:nxx - 1] * (T[I, 2:] - T[I, J]) / dx[:, 1:]
          - kx[I, 0:nxx - 2] * (T[I, J] - T[I, 0:nxx - 2]) / dx[:, :-1]) / dxb[:, 0:nxx - 2]
    fz = (kz[1:nz - 1, J] * (T[2:, J] - T[I, J]) / dz[1:]
          - kz[0:nz - 2, J] * (T[I, J] - T[0:nz - 2, J]) / dz[:-1]) / dzb[0:nz - 2]
    y[I, J] = c * (fx + fz) - T[I, J]
    if bc[1] == BC_FIXFLOW:
        y[I, 0] = kx[I, 0] * (T[I, 1] - T[I, 0]) / (x[1] - x[0])
    if bc[3] == BC_FIXFLOW:
        n = nxx - 1
        y[I, n] = kx[I, n - 1] * (T[I, n] - T[I, n - 1]) / (x[n] - x[n - 1])
    if bc[0] == BC_FIXFLOW:
        y[0, :] = kz[0, :] * (T[1, :] - T[0, :]) / (z[1] - z[0])
    if bc[2] == BC_FIXFLOW:
        m = nz - 1
        y[m, :] = kz[m - 1, :] * (T[m, :] - T[m - 1, :]) / (z[m] - z[m - 1])
    return y.reshape(-1)


def heat_rhs(nx, T, Cp, rho, H, bc, bcvalue, tstep):
    nz, nxx = int(nx[0]), int(nx[1])
    r = -T - tstep * H / (rho * Cp)
    r = np.array(r, dtype=np.float64, copy=True)
    r[1:nz - 1, 0] = bcvalue[1]
    r[1:nz - 1, nxx - 1] = bcvalue[3]
    r[0, :] = bcvalue[0]
    r[nz - 1, :] = bcvalue[2]
    return r.reshape(-1)


def heat_solve(nx, grid, gridmp, T, k, Cp, rho, H, bc, bcvalue, tstep):
    """pylamp2.py:415-421."""
    A, rhs = heat_csr(nx, grid, gridmp, T, k, Cp, rho, H, bc, bcvalue, tstep)
    return spla.spsolve(sp.csc_matrix(A), rhs).reshape(int(nx[0]), int(nx[1]))


# =======================================================================================
# Marker-in-cell
# =======================================================================================
# Rectilinear (non-uniform) grids -- SURVEY section 8 f4.  The reference's marker code is regular-grid only: it
# finds the cell with the regular-grid formula and then reads the coordinates of THAT cell from the grid
# arrays (pylamp_trac.py:46-47,226-227,249-250), so on a non-uniform grid the weights leave [0,1].  With
# RECT_SEARCH the cell comes from a per-axis search instead; everything else is unchanged, and on a uniform
# grid both lookups agree.  This mode is DEFINED HERE (no reference behaviour to pin it to).
RECT_SEARCH = False


class rect_search:
    """with rect_search(): ... -- marker <-> grid transfers locate cells by per-axis search"""
    def __init__(self, on=True): self.on = on
    def __enter__(self):
        global RECT_SEARCH
        self.old = RECT_SEARCH; RECT_SEARCH = self.on
    def __exit__(self, *a):
        global RECT_SEARCH
        RECT_SEARCH = self.old


def _cell_lookup(tr_x, grid, nx):
    """Regular-grid cell formula shared by both directions (pylamp_trac.py:42-47,222-227)."""
    if RECT_SEARCH:
        ie = np.searchsorted(np.asarray(grid[IZ]), tr_x[:, IZ], side='right').astype(np.int64) - 1
        je = np.searchsorted(np.asarray(grid[IX]), tr_x[:, IX], side='right').astype(np.int64) - 1
        return ie, je
    Lmin = [grid[d][0] for d in range(DIM)]
    L = [grid[d][-1] - grid[d][0] for d in range(DIM)]
    ie = np.floor((nx[IZ] - 1) * (tr_x[:, IZ] - Lmin[IZ]) / L[IZ]).astype(np.int64)
    je = np.floor((nx[IX] - 1) * (tr_x[:, IX] - Lmin[IX]) / L[IX]).astype(np.int64)
    return ie, je


def trac2grid(tr_x, tr_f, grid, nx, avgscheme=None):
    """Tracer -> grid averaging; returns the list of (nz,nx_) fields.

    Follows pylamp_trac.trac2grid method ELEM (pylamp_trac.py:192-318) including the
    grid auto-extension (207-220) and the final crop (313-316).
    """
    nf = tr_f.shape[1]
    if avgscheme is None:
        avgscheme = [AVG_ARITHW] * nf
    g = [np.array(grid[d], dtype=np.float64, copy=True) for d in range(DIM)]
    n = [int(nx[0]), int(nx[1])]
    addl, addr = [0, 0], [0, 0]
    for d in range(DIM):
        while np.min(tr_x[:, d]) < g[d][0]:
            g[d] = np.concatenate([[g[d][0] - (g[d][1] - g[d][0])], g[d]])
            n[d] += 1; addl[d] += 1
        while np.max(tr_x[:, d]) > g[d][-1]:
            g[d] = np.concatenate([g[d], [g[d][-1] + (g[d][-1] - g[d][-2])]])
            n[d] += 1; addr[d] += 1
    # NOTE the accumulator shape in the reference is mesh[0].shape, i.e. the *extended*
    # shape when the grid was modified and the caller's mesh shape otherwise.
    ie, je = _cell_lookup(tr_x, g, n)
    if RECT_SEARCH:      # a marker exactly on the last coordinate belongs to the last cell (a = 1)
        ie = np.minimum(ie, n[0] - 2); je = np.minimum(je, n[1] - 2)
    a = (tr_x[:, IZ] - g[IZ][ie]) / (g[IZ][ie + 1] - g[IZ][ie])
    b = (tr_x[:, IX] - g[IX][je]) / (g[IX][je + 1] - g[IX][je])
    w = [(1 - b) * (1 - a), (1 - b) * a, b * (1 - a), b * a]
    corners = [(ie, je), (ie + 1, je), (ie, je + 1), (ie + 1, je + 1)]
    shape = (n[0], n[1])
    wsum = np.zeros(shape); cnt = np.zeros(shape)
    for (ci, cj), wk in zip(corners, w):
        np.add.at(wsum, (ci, cj), wk)
        np.add.at(cnt, (ci, cj), 1.0)
    out = []
    for f in range(nf):
        sch = avgscheme[f]
        acc = np.zeros(shape)
        if sch & AVG_ARITH:
            val = tr_f[:, f]
        elif sch & AVG_GEOM:
            with np.errstate(divide='ignore', invalid='ignore'):
                val = np.log(tr_f[:, f])
        else:
            raise Exception("invalid averaging scheme")
        for (ci, cj), wk in zip(corners, w):
            with np.errstate(invalid='ignore'):
                np.add.at(acc, (ci, cj), val * wk if (sch & AVG_WEIGHTED) else val)
        den = wsum if (sch & AVG_WEIGHTED) else cnt
        with np.errstate(divide='ignore', invalid='ignore'):
            if sch & AVG_ARITH:
                res = acc / den
            else:
                acc[np.isinf(acc)] = 0
                res = np.exp(acc / den)
        out.append(res[addl[0]:n[0] - addr[0], addl[1]:n[1] - addr[1]].copy())
    return out


def grid2trac(tr_x, grid, gridfield, nx, defval=np.nan, method=M_LINEAR, stop_on_error=False):
    """Grid -> tracer interpolation; returns (ntrac, nf) (pylamp_trac.py:30-158)."""
    nf = len(gridfield)
    nz, nxx = int(nx[0]), int(nx[1])
    ie, je = _cell_lookup(tr_x, grid, nx)
    bad = (ie < 0) | (ie > nz - 1) | (je < 0) | (je > nxx - 1)
    if RECT_SEARCH:      # at or beyond the last coordinate = outside (the strict path raises IndexError there)
        bad |= (ie > nz - 2) | (je > nxx - 2)
    if stop_on_error and bad.any():
        raise Exception("stopOnError in grid2trac")
    ie = np.where(bad, 0, ie); je = np.where(bad, 0, je)
    z, x = np.asarray(grid[IZ]), np.asarray(grid[IX])
    dz0 = tr_x[:, IZ] - z[ie]; dz1 = -(tr_x[:, IZ] - z[ie + 1])
    dx0 = tr_x[:, IX] - x[je]; dx1 = -(tr_x[:, IX] - x[je + 1])
    out = np.empty((tr_x.shape[0], nf))
    if method & M_NEAREST:
        d2 = np.stack([dz0 ** 2 + dx0 ** 2, dz0 ** 2 + dx1 ** 2,
                       dz1 ** 2 + dx0 ** 2, dz1 ** 2 + dx1 ** 2], axis=1)
        c = np.argmin(d2, axis=1)
        dj = c % 2; di = (c - dj) // 2
        for f in range(nf):
            out[:, f] = gridfield[f][ie + di, je + dj]
    else:
        b = dx0 / (dx0 + dx1)
        a = dz0 / (dz0 + dz1)

        def bil(F):
            return ((1 - b) * (1 - a) * F[ie, je] + b * (1 - a) * F[ie, je + 1]
                    + (1 - b) * a * F[ie + 1, je] + b * a * F[ie + 1, je + 1])
        if method & M_LINEAR:
            for f in range(nf):
                out[:, f] = bil(gridfield[f])
        elif method & M_VELDIV:
            if nf != 2:
                raise Exception("VELDIV expects (vz, vx)")
            Vz, Vx = gridfield[IZ], gridfield[IX]
            hz = (z[1:] - z[:-1])[ie]; hx = (x[1:] - x[:-1])[je]
            C10 = (0.5 * hx / hz) * (Vz[ie, je] - Vz[ie + 1, je] + Vz[ie + 1, je + 1] - Vz[ie, je + 1])
            C20 = (0.5 * hz / hx) * (Vx[ie, je] - Vx[ie, je + 1] + Vx[ie + 1, je + 1] - Vx[ie + 1, je])
            out[:, IX] = bil(Vx) + b * (1 - b) * C10
            out[:, IZ] = bil(Vz) + a * (1 - a) * C20
        else:
            raise Exception("unknown method")
    if method & M_VELDIV and not (method & (M_NEAREST | M_LINEAR)):
        # Reference quirk (pylamp_trac.py:83,98-156): the per-field loop recomputes BOTH
        # components on every pass and then resets only column `ifield`, so after the last
        # pass only the last field (vx) holds defval for out-of-grid tracers; vz keeps the
        # value extrapolated from cell (0,0) with the tracer's real coordinates.
        out[bad, nf - 1] = defval
    else:
        out[bad, :] = defval
    return out


def rk4(tr_x, grids, vels, nx, tstep):
    """pylamp_trac.RK order=4 (pylamp_trac.py:347-388). NB weights 1,1,1,1 (line 385)."""
    n2 = [int(nx[0]) + 1, int(nx[1]) + 1]
    k1 = grid2trac(tr_x, grids, vels, n2, defval=0, method=M_VELDIV)
    k2 = grid2trac(tr_x + 0.5 * tstep * k1, grids, vels, n2, defval=0, method=M_VELDIV)
    k3 = grid2trac(tr_x + 0.5 * tstep * k2, grids, vels, n2, defval=0, method=M_VELDIV)
    k4 = grid2trac(tr_x + tstep * k3, grids, vels, n2, defval=0, method=M_VELDIV)
    xnew = tr_x + (1 / 6) * tstep * (k1 + k2 + k3 + k4)
    return (xnew - tr_x) / tstep, xnew


# =======================================================================================
# Driver glue (pylamp2.py loop body)
# =======================================================================================
def advection_velocity(newvel, gridmp, nx, bc):
    """Cell-centred velocities on the (nz+1, nx_+1) padded grid (pylamp2.py:491-545)."""
    nz, nxx = int(nx[0]), int(nx[1])
    vz, vx = newvel
    Vz = np.zeros((nz + 1, nxx + 1)); Vx = np.zeros((nz + 1, nxx + 1))
    Vz[1:-1, 1:-1] = 0.5 * (vz[1:, :-1] + vz[:-1, :-1])
    Vx[1:-1, 1:-1] = 0.5 * (vx[:-1, 1:] + vx[:-1, :-1])
    gz = np.insert(gridmp[IZ], 0, gridmp[IZ][0] - (gridmp[IZ][1] - gridmp[IZ][0]))
    gx = np.insert(gridmp[IX], 0, gridmp[IX][0] - (gridmp[IX][1] - gridmp[IX][0]))

    def wall(b):
        if b & BC_FREESLIP:
            return 1.0
        if b == BC_NOSLIP:
            return None  # `& NOSLIP` is never true in the reference: ghost stays 0
        raise Exception("oracle: unsupported stokes BC for advection ghost fill")
    # order z0, x0, zL, xL exactly as the source
    s = wall(bc[0])
    if s is not None:
        Vx[0, :] = Vx[1, :]; Vz[0, :] = -Vz[1, :]
    s = wall(bc[1])
    if s is not None:
        Vz[:, 0] = Vz[:, 1]; Vx[:, 0] = -Vx[:, 1]
    s = wall(bc[2])
    if s is not None:
        Vx[-1, :] = Vx[-2, :]; Vz[-1, :] = -Vz[-2, :]
    s = wall(bc[3])
    if s is not None:
        Vz[:, -1] = Vz[:, -2]; Vx[:, -1] = -Vx[:, -2]
    return [gz, gx], [Vz, Vx]


def property_update(tr_f, tdep_rho, tdep_eta, Tref=1623.0, etamin=1e17, etamax=1e23):
    """pylamp2.py:291-303 (in place)."""
    if tdep_rho:
        tr_f[:, TR_RHO] = ((tr_f[:, TR_ALP] * (tr_f[:, TR_TMP] - Tref) + 1) / tr_f[:, TR_RH0]) ** (-1)
    else:
        tr_f[:, TR_RHO] = tr_f[:, TR_RH0]
    if tdep_eta:
        e = tr_f[:, TR_ET0] * np.exp(tr_f[:, TR_ACE] / (GASR * tr_f[:, TR_TMP])
                                     - tr_f[:, TR_ACE] / (GASR * Tref))
        tr_f[:, TR_ETA] = np.clip(e, etamin, etamax)
    else:
        tr_f[:, TR_ETA] = tr_f[:, TR_ET0]


class StepConfig:
    """Options of pylamp2.py:37-77 that the step below honours."""

    def __init__(self, **kw):
        self.do_heatdiff = True
        self.do_subgrid_heatdiff = True
        self.tdep_rho = True
        self.tdep_eta = True
        self.etamin, self.etamax, self.Tref = 1e17, 1e23, 1623.0
        self.tstep_adv_max = 50e9 * SECINYR; self.tstep_adv_min = 50e-9 * SECINYR
        self.tstep_dif_max = 50e9 * SECINYR; self.tstep_dif_min = 50e-9 * SECINYR
        self.tstep_modifier = 0.67
        self.bcstokes = [1, 1, 1, 1]
        self.bcheat = [BC_FIXTEMP, BC_FIXFLOW, BC_FIXTEMP, BC_FIXFLOW]
        self.bcheatvals = [273.0, 0.0, 1623.0, 0.0]
        self.surface_stabilization = False; self.surfstab_theta = 0.5; self.surfstab_tstep = -1
        # pylamp2.py:39-42.  tracdens_min = 0 never finds a deficient cell (the stock values are 45 / 25)
        self.tracdens = 0; self.tracdens_min = 0; self.tracs_fence_enabled = True
        self.__dict__.update(kw)


def fence_and_delete(tr_x, tr_f, trac_vel, L, fence_enabled=True):
    """pylamp2.py:557-581 for the supported (non-CYCLIC, non-FLOWTHRU) walls: tracers at or beyond a wall are put
    back EPS inside it, or -- fence off -- get TR__ID = -1 and are deleted from all three arrays.
    Returns (tr_x, tr_f, trac_vel, number removed); tr_x / tr_f are modified in place before the deletion."""
    for d in range(DIM):
        idx = tr_x[:, d] <= 0
        if fence_enabled:
            tr_x[idx, d] = EPS
        else:
            tr_f[idx, TR__ID] = -1
        idx = tr_x[:, d] >= L[d]
        if fence_enabled:
            tr_x[idx, d] = L[d] - EPS
        else:
            tr_f[idx, TR__ID] = -1
    out = tr_f[:, TR__ID] < 0
    n_out = int(np.sum(out))
    if n_out:
        keep = ~out
        tr_x, tr_f, trac_vel = tr_x[keep], tr_f[keep], trac_vel[keep]
    return tr_x, tr_f, trac_vel, n_out


def census(tr_x, nx, L):
    """Tracers per cell, pylamp2.py:588-594 (cell = floor((n-1) x / L), counted with np.bincount).
    Returns (ielem, jelem, counts[(nz-1)*(nx_-1)])."""
    ielem = np.floor((nx[IZ] - 1) * tr_x[:, IZ] / L[IZ]).astype(int)
    jelem = np.floor((nx[IX] - 1) * tr_x[:, IX] / L[IX]).astype(int)
    kelem = ielem * (nx[IX] - 1) + jelem
    ncell = (nx[IZ] - 1) * (nx[IX] - 1)
    return ielem, jelem, np.bincount(kelem, minlength=ncell)[:ncell]


def inject(tr_x, tr_f, grid, nx, L, tracdens, tracdens_min, rand=None):
    """Refill of depleted cells, pylamp2.py:595-633.  Cells holding fewer than tracdens_min tracers receive
    tracdens - count new ones, cell by cell in ascending cell number, appended behind the existing tracers:
    positions uniformly random inside the cell (np.random.rand(m, DIM), legacy global stream unless `rand` is
    given), every field except the ID = plain mean of the tracers that were in the cell BEFORE any injection
    (0/0 = NaN for an empty cell), IDs arange(max(ID), max(ID) + m) with the maximum taken over the array as it
    stands -- i.e. the first new ID of every cell repeats the last ID handed out (pylamp2.py:621-622).
    Returns (tr_x, tr_f, info) with info = dict(cells, n_missing, n_injected)."""
    rand = rand or np.random.rand
    ielem, jelem, cnt = census(tr_x, nx, L)
    few = cnt < tracdens_min
    cells = np.where(few)[0]
    info = dict(cells=cells, n_missing=tracdens - cnt[few], n_injected=0)
    if cells.size == 0:
        return tr_x, tr_f, info
    ci = cells // (nx[IX] - 1); cj = cells % (nx[IX] - 1)
    n_missing = info["n_missing"]
    prev_f = tr_f.copy()
    new_x, new_f = [], []
    maxid = np.max(tr_f[:, TR__ID])
    for k in range(cells.size):
        m = int(n_missing[k])
        x = rand(m, DIM)
        x[:, IX] = x[:, IX] * (grid[IX][cj[k] + 1] - grid[IX][cj[k]]) + grid[IX][cj[k]]
        x[:, IZ] = x[:, IZ] * (grid[IZ][ci[k] + 1] - grid[IZ][ci[k]]) + grid[IZ][ci[k]]
        f = np.zeros((m, NFTRAC))
        f[:, TR__ID] = np.arange(maxid, maxid + m)
        inel = (ielem == ci[k]) & (jelem == cj[k])
        with np.errstate(invalid="ignore", divide="ignore"):
            for q in range(NFTRAC):
                if q != TR__ID:
                    f[:, q] = np.sum(prev_f[inel, q]) / np.sum(inel)
        if m > 0:
            maxid = max(maxid, f[-1, TR__ID])
        new_x.append(x); new_f.append(f)
    info["n_injected"] = int(np.sum(n_missing))
    return np.concatenate([tr_x] + new_x), np.concatenate([tr_f] + new_f), info


def step(state, cfg, it):
    """One time step of the stock loop (pylamp2.py:273-633): everything up to the advection, the fence / deletion
    (574-581) and the census + injection (588-633; inert while cfg.tracdens_min == 0).

    state: dict with nx, L, grid, tr_x (n,2), tr_f (n,13) and, for it>1, newtemp.
    Returns a dict of the step's grid fields; replaces state['tr_x'], state['tr_f'] (after injection) and stores
    the pre-injection arrays the stock snapshot writes as out['snap_tr_x'], out['snap_tr_f'].
    """
    nx, L, grid = state['nx'], state['L'], state['grid']
    nz, nxx = nx
    gridmp = gridmp_of(grid)
    dx = [L[i] / (nx[i] - 1) for i in range(DIM)]
    tr_x, tr_f = state['tr_x'], state['tr_f']
    property_update(tr_f, cfg.tdep_rho, cfg.tdep_eta, cfg.Tref, cfg.etamin, cfg.etamax)
    out = {}
    if cfg.do_heatdiff:
        f_rho, f_etas, f_Cp, f_T, f_H, f_mat = trac2grid(
            tr_x, tr_f[:, [TR_RHO, TR_ETA, TR_HCP, TR_TMP, TR_IHT, TR_MAT]], grid, nx,
            [AVG_ARITHW, AVG_GEOMW, AVG_ARITHW, AVG_ARITHW, AVG_ARITHW, AVG_ARITHW])
        f_etan, = trac2grid(tr_x, tr_f[:, [TR_ETA]], gridmp, nx, [AVG_GEOMW])
        f_kz, = trac2grid(tr_x, tr_f[:, [TR_HCD]], [gridmp[IZ], grid[IX]], nx, [AVG_ARITHW])
        f_kx, = trac2grid(tr_x, tr_f[:, [TR_HCD]], [grid[IZ], gridmp[IX]], nx, [AVG_ARITHW])
        if it > 1:
            nt = state['newtemp']
            f_T[:, 0] = nt[:, 0]; f_T[:, -1] = nt[:, -1]; f_T[0, :] = nt[0, :]; f_T[-1, :] = nt[-1, :]
        diffusivity = f_kz / (f_rho * f_Cp)
        tstep_temp = cfg.tstep_modifier * np.min(dx) ** 2 / np.max(2 * diffusivity)
        tstep_temp = max(min(tstep_temp, cfg.tstep_dif_max), cfg.tstep_dif_min)
    else:
        f_rho, f_etas = trac2grid(tr_x, tr_f[:, [TR_RHO, TR_ETA]], grid, nx, [AVG_ARITHW, AVG_GEOMW])
        f_etan, = trac2grid(tr_x, tr_f[:, [TR_ETA]], gridmp, nx, [AVG_GEOM])

    ss = cfg.surface_stabilization
    if (not ss) or cfg.surfstab_tstep < 0:                       # pylamp2.py:352-355
        xsol = stokes_solve(nx, grid, f_etas, f_etan, f_rho, cfg.bcstokes)
    else:
        xsol = stokes_solve(nx, grid, f_etas, f_etan, f_rho, cfg.bcstokes, surfstab=True,
                            tstep=cfg.surfstab_tstep, theta=cfg.surfstab_theta)
    newvel, newpres = x2vp(xsol, nx)
    tstep_stokes = cfg.tstep_modifier * np.min(dx) / np.max(newvel)
    tstep_stokes = max(min(tstep_stokes, cfg.tstep_adv_max), cfg.tstep_adv_min)
    if cfg.surfstab_tstep > 0:                                   # pylamp2.py:368-372
        tstep_stokes = cfg.surfstab_tstep
    if cfg.do_heatdiff:
        limiter = "H" if tstep_temp < tstep_stokes else "S"
        tstep = min(tstep_temp, tstep_stokes)
    else:
        tstep, limiter = tstep_stokes, "S"
    nresolve = 0
    if ss and cfg.surfstab_tstep < 0:                            # pylamp2.py:387-405
        while True:
            xsol = stokes_solve(nx, grid, f_etas, f_etan, f_rho, cfg.bcstokes, surfstab=True, tstep=tstep,
                                theta=cfg.surfstab_theta)
            newvel, newpres = x2vp(xsol, nx)
            nresolve += 1
            check = cfg.tstep_modifier * np.min(dx) / np.max(newvel)
            if check < tstep:
                tstep = check; limiter = "Ss"
            else:
                break
    out['nresolve'] = nresolve
    out.update(velz=newvel[IZ], velx=newvel[IX], pres=newpres, rho=f_rho, etas=f_etas,
               etan=f_etan, tstep=tstep, limiter=limiter)

    if cfg.do_heatdiff:
        newtemp = heat_solve(nx, grid, gridmp, f_T, [f_kz, f_kx], f_Cp, f_rho, f_H,
                             cfg.bcheat, cfg.bcheatvals, tstep)
        old_T = tr_f[:, TR_TMP].copy()
        if it == 1:
            tr_f[:, TR_TMP] = grid2trac(tr_x, grid, [newtemp], nx, method=M_LINEAR,
                                        stop_on_error=True)[:, 0]
        else:
            tr_f[:, TR_TMP] += grid2trac(tr_x, grid, [newtemp - f_T], nx, method=M_LINEAR,
                                         stop_on_error=True)[:, 0]
            if cfg.do_subgrid_heatdiff:
                dt0 = tr_f[:, TR_HCP] * tr_f[:, TR_RHO] / (
                    tr_f[:, TR_HCD] * ((2 / dx[IX]) ** 2 + (2 / dx[IZ]) ** 2))
                Tsub = old_T - (old_T - tr_f[:, TR_TMP]) * np.exp(-0.5 * tstep / dt0)
                dTs = Tsub - tr_f[:, TR_TMP]
                f_sgc, = trac2grid(tr_x, dTs[:, None], grid, nx, [AVG_ARITHW])
                back = grid2trac(tr_x, grid, [f_sgc], nx, method=M_LINEAR, stop_on_error=True)[:, 0]
                tr_f[:, TR_TMP] = Tsub - back
        state['newtemp'] = newtemp
        out.update(temp=newtemp, f_T=f_T)

    grids, vels = advection_velocity(newvel, gridmp, nx, cfg.bcstokes)
    trac_vel, xnew = rk4(tr_x, grids, vels, nx, tstep)
    xnew, tr_f, trac_vel, n_removed = fence_and_delete(xnew, tr_f, trac_vel, L, cfg.tracs_fence_enabled)
    out.update(tr_v=trac_vel, n_removed=n_removed, snap_tr_x=xnew, snap_tr_f=tr_f)
    if cfg.tracdens_min > 0:
        xnew, tr_f, info = inject(xnew, tr_f, grid, nx, L, cfg.tracdens, cfg.tracdens_min, state.get('rand'))
        out.update(inject=info)
    state['tr_x'] = xnew; state['tr_f'] = tr_f
    return out
