"""3-D staggered Stokes + heat (BASELINE config 5).  The reference has no 3-D implementation, so there is no oracle to
pin (parity UNPINNED).  Validation follows the survey's recipe: (i) a y-invariant extrusion of a 2-D
problem must reproduce the 2-D oracle on every y-slice -- operator to 1e-12, solution to 1e-6; (ii) a manufactured
solution with genuinely 3-D structure converges at second order; (iii) true residuals."""
import numpy as np
import pytest

from conftest import relerr

pytestmark = pytest.mark.gpu


def _fields2d(oracle, nx2, L2, seed, contrast=3.0):
    rng = np.random.default_rng(seed)
    grid = [np.linspace(0, L2[0], nx2[0]), np.linspace(0, L2[1], nx2[1])]
    Z, X = np.meshgrid(*grid, indexing="ij")
    Zc, Xc = np.meshgrid(*oracle.gridmp_of(grid), indexing="ij")
    f = lambda z, x: 1e20 * 10 ** (0.5 * contrast * np.sin(2 * np.pi * x / L2[1]) * np.cos(np.pi * z / L2[0]))
    etas, etan = f(Z, X), f(Zc, Xc)
    rho = 3300 + 40 * np.sin(2 * np.pi * X / L2[1]) * np.sin(np.pi * Z / L2[0]) + rng.uniform(-1, 1, nx2)
    return grid, etas, etan, rho


def _ygrid(grid2, ny):
    # Kcont uses avgd = L / n (pylamp_stokes.py:119-120); avgdy = mean of the two others makes the 3-D Kcont equal the 2-D one
    avg = 0.5 * ((grid2[0][-1] - grid2[0][0]) / grid2[0].size + (grid2[1][-1] - grid2[1][0]) / grid2[1].size)
    return np.linspace(0, avg * ny, ny)


def test_extrusion_operator_matches_2d_oracle(oracle):
    from pylamp_amd import pylamp3d as P3
    nx2 = [12, 10]; L2 = [660e3, 500e3]; ny = 9
    grid2, etas, etan, rho = _fields2d(oracle, nx2, L2, 1)
    # non-uniform z and x: the coefficient tables are exercised as well
    rng = np.random.default_rng(2)
    for d in range(2):
        w = rng.uniform(0.7, 1.3, nx2[d] - 1); g = np.concatenate([[0.0], np.cumsum(w)]); grid2[d] = g * (L2[d] / g[-1])
    gy = _ygrid(grid2, ny)
    nx3 = nx2 + [ny]
    ext = lambda a: np.repeat(a[:, :, None], ny, axis=2)
    A, rhs = P3.makeStokesMatrix(nx3, grid2 + [gy], ext(etas), ext(etan), ext(rho))
    kc2, kb2 = oracle.stokes_scaling(grid2, etas, etan)
    assert A.Kcont == pytest.approx(kc2, rel=1e-13) and A.Kbond == pytest.approx(kb2, rel=1e-13)
    x2 = rng.standard_normal(3 * nx2[0] * nx2[1])
    X2 = x2.reshape(nx2[0], nx2[1], 3)
    X3 = np.zeros(nx3 + [4]); X3[..., 0] = X2[:, :, None, 0]; X3[..., 1] = X2[:, :, None, 1]; X3[..., 3] = X2[:, :, None, 2]
    y3 = (A @ X3.reshape(-1)).reshape(nx3 + [4])
    y2 = oracle.stokes_apply(nx2, grid2, etas, etan, [1, 1, 1, 1], x2).reshape(nx2[0], nx2[1], 3)
    r2 = oracle.stokes_rhs(nx2, rho).reshape(nx2[0], nx2[1], 3)
    R3 = rhs.reshape(nx3 + [4])
    scale = np.abs(y2).max(axis=(0, 1))
    for k in range(1, ny - 2):                      # slices whose rows are interior in y
        for q3, q2 in ((0, 0), (1, 1), (3, 2)):
            d = np.abs(y3[:, :, k, q3] - y2[:, :, q2])
            if q2 == 2 and k != 2:
                d[3, 2] = 0.0                       # the 3-D system anchors the single cell (3, 2, 2)
            assert d.max() < 1e-12 * scale[q2], (k, q3, d.max() / scale[q2])
        assert np.abs(y3[:, :, k, 2]).max() < 1e-12 * scale[1]            # y-momentum rows vanish identically
        assert np.allclose(R3[:, :, k, 0], r2[:, :, 0], rtol=1e-14, atol=0) and not R3[:, :, k, 1:].any()


def test_extrusion_solution_matches_2d_direct_solve(oracle):
    from pylamp_amd import pylamp3d as P3
    nx2 = [33, 41]; L2 = [660e3, 820e3]; ny = 17
    grid2, etas, etan, rho = _fields2d(oracle, nx2, L2, 3)
    gy = _ygrid(grid2, ny)
    nx3 = nx2 + [ny]
    ext = lambda a: np.repeat(a[:, :, None], ny, axis=2)
    A, rhs = P3.makeStokesMatrix(nx3, grid2 + [gy], ext(etas), ext(etan), ext(rho))
    x = P3.solve(A)
    st = A.last_stats
    assert st["converged"] == 1 and st["rel_residual"] <= P3.DEFAULT_RTOL, st
    (vz, vx, vy), p = P3.x2vp(x, nx3)
    (rz, rx), rp = oracle.x2vp(oracle.stokes_solve(nx2, grid2, etas, etan, rho, [1, 1, 1, 1]), nx2)
    vn = np.sqrt(np.sum(rz ** 2) + np.sum(rx ** 2))
    for k in range(ny - 1):                         # every physical y-slice
        ev = np.sqrt(np.sum((vz[:, :, k] - rz) ** 2) + np.sum((vx[:, :, k] - rx) ** 2)) / vn
        assert ev < 1e-6, (k, ev)
        assert relerr(p[:-1, :-1, k], rp[:-1, :-1]) < 1e-5
    assert np.abs(vy).max() < 1e-6 * max(np.abs(rz).max(), np.abs(rx).max())
    # the reported residual is the true one
    r = rhs - A @ x
    assert np.linalg.norm(r) / np.linalg.norm(rhs) < 1e-6


def _manufactured(n, strict):
    """Constant viscosity, rho = rho0 + drho sin(kz z) cos(kx x) cos(ky y) in a free-slip cube: the exact solution is
    vz = W sin cos cos, vx = U cos sin cos, vy = V cos cos sin with W = drho g (kx^2 + ky^2) / (eta k^4)."""
    from pylamp_amd import pylamp3d as P3
    L = [1.0e5, 1.3e5, 0.9e5]; eta = 1e20; drho = 30.0; g = 9.81
    grid = [np.linspace(0, L[d], n) for d in range(3)]
    kz, kx, ky = np.pi / L[0], np.pi / L[1], np.pi / L[2]
    k2 = kz * kz + kx * kx + ky * ky
    W = drho * g * (kx * kx + ky * ky) / (eta * k2 * k2)
    Pm = -eta * k2 * kz * W / (kx * kx + ky * ky)
    U, V = kx * Pm / (eta * k2), ky * Pm / (eta * k2)
    Z, X, Y = np.meshgrid(*grid, indexing="ij")
    rho = 3300.0 + drho * np.sin(kz * Z) * np.cos(kx * X) * np.cos(ky * Y)
    one = np.full((n, n, n), eta)
    A, rhs = P3.makeStokesMatrix([n, n, n], grid, one, one, rho, strict_reference=strict)
    x = P3.solve(A, rtol=1e-11)
    assert A.last_stats["converged"] == 1, A.last_stats
    (vz, vx, vy), p = P3.x2vp(x, [n, n, n])
    mid = [0.5 * (c[1:] + c[:-1]) for c in grid]
    Zz, Xz, Yz = np.meshgrid(grid[0], mid[1], mid[2], indexing="ij")           # vz at (z_i, x_j+1/2, y_k+1/2)
    ez = np.abs(vz[:, :-1, :-1] - W * np.sin(kz * Zz) * np.cos(kx * Xz) * np.cos(ky * Yz)).max() / abs(W)
    Zx, Xx, Yx = np.meshgrid(mid[0], grid[1], mid[2], indexing="ij")
    ex = np.abs(vx[:-1, :, :-1] - U * np.cos(kz * Zx) * np.sin(kx * Xx) * np.cos(ky * Yx)).max() / abs(W)
    Zy, Xy, Yy = np.meshgrid(mid[0], mid[1], grid[2], indexing="ij")
    ey = np.abs(vy[:-1, :-1, :] - V * np.cos(kz * Zy) * np.cos(kx * Xy) * np.sin(ky * Yy)).max() / abs(W)
    h = [L[d] / (n - 1) for d in range(3)]
    div = (vz[1:, :-1, :-1] - vz[:-1, :-1, :-1]) / h[0] + (vx[:-1, 1:, :-1] - vx[:-1, :-1, :-1]) / h[1] + (vy[:-1, :-1, 1:] - vy[:-1, :-1, :-1]) / h[2]
    A._ctx.close()
    return max(ez, ex, ey), np.abs(div[1:-1, 1:-1, 1:-1]).max() * h[0] / abs(W), A.last_stats["iterations"]


def test_manufactured_solution_convergence_order():
    """Natural wall rows: second order.  The reference's slaved wall rows (strict mode, the default) impose free slip half a
    cell inside the wall -- the same treatment as in 2-D (pylamp_stokes.py:170-175) -- and converge at first order."""
    e17, d17, it17 = _manufactured(17, False)
    e33, d33, it33 = _manufactured(33, False)
    assert e33 < 0.01 and e17 / e33 > 3.0, (e17, e33)        # second order: the error drops ~4x per halving of h
    assert d17 < 1e-7 and d33 < 1e-7                          # discretely divergence-free
    assert it33 <= 2 * it17 + 10                              # multigrid: iterations do not grow with the grid
    s17, _, _ = _manufactured(17, True)
    s33, ds, _ = _manufactured(33, True)
    assert 1.6 < s17 / s33 < 2.6 and ds < 1e-7, (s17, s33)   # first order


def test_heat_extrusion_matches_2d_oracle(oracle):
    from pylamp_amd import pylamp3d as P3
    nx2 = [17, 21]; L2 = [660e3, 800e3]; ny = 9
    rng = np.random.default_rng(7)
    grid2 = [np.linspace(0, L2[0], nx2[0]), np.linspace(0, L2[1], nx2[1])]
    gy = np.linspace(0, 300e3, ny)
    gm2 = oracle.gridmp_of(grid2); gm3 = gm2 + oracle.gridmp_of([gy, gy])[:1]
    kz = rng.uniform(2, 5, nx2); kx = rng.uniform(2, 5, nx2); Cp = rng.uniform(1000, 1250, nx2); rho = rng.uniform(3200, 3400, nx2)
    H = rng.uniform(0, 1e-9, nx2) * 3300; T0 = rng.uniform(273, 1623, nx2)
    dt = 0.67 * (L2[0] / (nx2[0] - 1)) ** 2 / np.max(2 * kz / (rho * Cp))
    nx3 = nx2 + [ny]
    ext = lambda a: np.repeat(a[:, :, None], ny, axis=2)
    bc3 = [0, 1, 1, 0, 1, 1]; bv3 = [273.0, 0.0, 0.0, 1623.0, 0.0, 0.0]          # y-walls insulating
    A, rhs = P3.makeDiffusionMatrix(nx3, grid2 + [gy], gm3, ext(T0), [ext(kz), ext(kx), ext(rng.uniform(2, 5, nx2))], ext(Cp), ext(rho),
                                    ext(H), bc3, bv3, dt)
    x2 = rng.standard_normal(nx2[0] * nx2[1])
    y3 = (A @ ext(x2.reshape(nx2)).reshape(-1)).reshape(nx3)
    y2 = oracle.heat_apply(nx2, grid2, gm2, [kz, kx], Cp, rho, [0, 1, 0, 1], dt, x2).reshape(nx2)
    r2 = oracle.heat_rhs(nx2, T0, Cp, rho, H, [0, 1, 0, 1], [273.0, 0.0, 1623.0, 0.0], dt).reshape(nx2)
    for k in range(1, ny - 1):
        assert np.abs(y3[:, :, k] - y2).max() < 1e-12 * np.abs(y2).max()
        assert np.allclose(rhs.reshape(nx3)[:, :, k], r2, rtol=1e-14, atol=0)
    T3 = P3.x2t(P3.solve_heat(A), nx3)
    assert A.last_stats["converged"] == 1, A.last_stats
    T2 = oracle.heat_solve(nx2, grid2, gm2, T0, [kz, kx], Cp, rho, H, [0, 1, 0, 1], [273.0, 0.0, 1623.0, 0.0], dt).reshape(nx2)
    for k in range(ny):
        assert relerr(T3[:, :, k], T2) < 1e-6


def test_config5_solve_129_cubed():
    """A 129^3 instance of BASELINE config 5 (257^3 itself runs in `bench.py --config 3d257`): T-dependent viscosity over
    three decades with 3-D structure; converged at the default tolerance, true residual recomputed on the host side of the
    C ABI, discretely divergence-free, multigrid hierarchy of 6 levels."""
    from pylamp_amd import pylamp3d as P3
    n = 129; L = [660e3] * 3
    grid = [np.linspace(0, L[d], n) for d in range(3)]
    mid = [np.append(0.5 * (g[1:] + g[:-1]), g[-1] + 0.5 * (g[-1] - g[-2])) for g in grid]

    def field(c):
        Z, X, Y = np.meshgrid(*c, indexing="ij", sparse=True)
        return 273 + 1350 * np.clip(Z / L[0], 0, 1) + 60 * np.sin(3 * np.pi * X / L[1]) * np.sin(np.pi * Z / L[0]) * np.cos(2 * np.pi * Y / L[2])
    eta = lambda T: np.clip(1e20 * np.exp(120e3 / (8.31446 * T) - 120e3 / (8.31446 * 1623)), 1e17, 1e23)
    Tn = field(grid)
    A, rhs = P3.makeStokesMatrix([n, n, n], grid, eta(Tn), eta(field(mid)), 3300 / (3.5e-5 * (Tn - 1623) + 1))
    x = P3.solve(A)
    st = A.last_stats
    assert st["converged"] == 1 and st["rel_residual"] <= P3.DEFAULT_RTOL and st["iterations"] < 120, st
    assert A.mg_info()[0] == 6
    r = rhs - A @ x
    assert np.linalg.norm(r) / np.linalg.norm(rhs) < 1e-6
    (vz, vx, vy), p = P3.x2vp(x, [n, n, n])
    h = L[0] / (n - 1)
    div = (vz[1:, :-1, :-1] - vz[:-1, :-1, :-1]) / h + (vx[:-1, 1:, :-1] - vx[:-1, :-1, :-1]) / h + (vy[:-1, :-1, 1:] - vy[:-1, :-1, :-1]) / h
    vmax = max(np.abs(vz).max(), np.abs(vx).max(), np.abs(vy).max())
    # (RMS: the pressure-anchor cell has no continuity row -- its divergence is minus the sum of all others' -- and would dominate a maximum)
    dv = div[1:-1, 1:-1, 1:-1]
    assert h * np.sqrt(np.mean(dv ** 2)) < 1e-6 * np.sqrt(np.mean(vz ** 2 + vx ** 2 + vy ** 2))
    assert np.abs(vy).max() > 1e-3 * vmax                                                                     # genuinely 3-D flow
    assert 0 < st["error_estimate"] <= 3e-8, st
    A._ctx.close()


def test_config5_257_cubed_resident():
    """BASELINE config 5 at its stated size, 257^3 nodes (68 M unknowns), as the bench runs it: device-resident Stokes and heat
    solves (nothing but the coefficients crosses PCIe), warm-started second solve.  No reference exists in 3-D (parity unpinned):
    the solution is checked through the host side of the C ABI -- true residual of the downloaded vector, discrete divergence,
    error estimate -- and the resident temperature against the operator it solves."""
    from pylamp_amd import pylamp3d as P3
    n = 257; L = [660e3] * 3
    grid = [np.linspace(0, L[d], n) for d in range(3)]
    mid = [np.append(0.5 * (g[1:] + g[:-1]), g[-1] + 0.5 * (g[-1] - g[-2])) for g in grid]

    def field(c):
        Z, X, Y = np.meshgrid(*c, indexing="ij", sparse=True)
        return 273 + 1350 * np.clip(Z / L[0], 0, 1) + 60 * np.sin(3 * np.pi * X / L[1]) * np.sin(np.pi * Z / L[0]) * np.cos(2 * np.pi * Y / L[2])
    eta = lambda T: np.clip(1e20 * np.exp(120e3 / (8.31446 * T) - 120e3 / (8.31446 * 1623)), 1e17, 1e23)
    Tn = field(grid)
    rho = 3300 / (3.5e-5 * (Tn - 1623) + 1)
    ctx = P3.Context3([n, n, n], grid)
    A, rhs = P3.makeStokesMatrix([n, n, n], grid, eta(Tn), eta(field(mid)), rho, ctx=ctx)
    assert P3.solve(A, resident=True) is None
    st = dict(A.last_stats)
    assert st["converged"] == 1 and st["rel_residual"] <= P3.DEFAULT_RTOL and st["iterations"] < 80 and 0 < st["error_estimate"] <= 3e-8, st
    x = P3.solution(A)
    r = rhs - A @ x
    assert np.linalg.norm(r) / np.linalg.norm(rhs) < 1e-6
    (vz, vx, vy), p = P3.x2vp(x, [n, n, n])
    h = L[0] / (n - 1)
    dv = ((vz[1:, :-1, :-1] - vz[:-1, :-1, :-1]) + (vx[:-1, 1:, :-1] - vx[:-1, :-1, :-1]) + (vy[:-1, :-1, 1:] - vy[:-1, :-1, :-1]))[1:-1, 1:-1, 1:-1]
    assert np.sqrt(np.mean(dv ** 2)) < 1e-6 * np.sqrt(np.mean(vz ** 2 + vx ** 2 + vy ** 2))
    del dv
    # a warm-started second solve of the same system ends at once
    P3.solve(A, resident=True, warm=True)
    assert A.last_stats["converged"] == 1 and A.last_stats["iterations"] <= 2, A.last_stats
    # heat on the same grid, resident
    k = np.full((n, n, n), 4.0); cp = np.full((n, n, n), 1250.0); H = np.full((n, n, n), 0.02e-6 / 3300)
    dt = 0.67 * h ** 2 / np.max(2 * 4.0 / (rho * 1250.0))
    Ah, rh = P3.makeDiffusionMatrix([n, n, n], grid, mid, Tn, [k, k, k], cp, rho, H, [0, 1, 1, 0, 1, 1], [273.0, 0, 0, 1623.0, 0, 0], dt, ctx=ctx)
    assert P3.solve_heat(Ah, resident=True) is None and Ah.last_stats["converged"] == 1, Ah.last_stats
    T = P3.solution(Ah, heat=True)
    assert np.linalg.norm(Ah @ T - rh) / np.linalg.norm(rh) < 1e-10
    ctx.close()


def _mantle3(n, L):
    grid = [np.linspace(0, L[d], n[d]) for d in range(3)]
    mid = [np.append(0.5 * (g[1:] + g[:-1]), g[-1] + 0.5 * (g[-1] - g[-2])) for g in grid]

    def field(c):
        Z, X, Y = np.meshgrid(*c, indexing="ij", sparse=True)
        return 273 + 1350 * np.clip(Z / L[0], 0, 1) + 60 * np.sin(3 * np.pi * X / L[1]) * np.sin(np.pi * Z / L[0]) * np.cos(2 * np.pi * Y / L[2])
    eta = lambda T: np.clip(1e20 * np.exp(120e3 / (8.31446 * T) - 120e3 / (8.31446 * 1623)), 1e17, 1e23)
    Tn = field(grid)
    return grid, mid, Tn, eta(Tn), eta(field(mid)), 3300 / (3.5e-5 * (Tn - 1623) + 1)


@pytest.mark.parametrize("n,layout", [([33, 49, 65], (2, 2, 2)), ([65, 65, 65], (2, 2, 2)), ([33, 33, 65], (1, 1, 4)), ([49, 33, 33], (3, 2, 1)),
                                      ([33, 41, 49], (2, 2, 2)), ([129, 65, 65], (2, 1, 1))])     # (the last: blocks large enough for the marching LDS kernels)
def test_blocks_3d_equal_one_rank(n, layout):
    """BASELINE config 5 on several ranks, rehearsed with Pz x Px x Py virtual ranks on one GPU (pylamp3d.VirtualCluster3: the
    multi-GPU code path except for the wire): operator, right-hand side, Stokes solve and heat solve of the block-decomposed
    contexts against the one-rank context -- operator to rounding, the same multigrid hierarchy and iteration counts, solutions
    to 1e-8.  Parity stays UNPINNED (no 3-D reference exists); this pins the decomposition against the one-rank code."""
    from pylamp_amd import pylamp3d as P3
    L = [660e3, 660e3 * (n[1] - 1) / (n[0] - 1), 660e3 * (n[2] - 1) / (n[0] - 1)]
    grid, mid, Tn, es, en, rho = _mantle3(n, L)
    rng = np.random.default_rng(11)
    x = rng.standard_normal(4 * int(np.prod(n)))
    k = np.full(n, 4.0); cp = np.full(n, 1250.0); H = np.full(n, 0.02e-6 / 3300)
    dt = 0.67 * (L[0] / (n[0] - 1)) ** 2 / np.max(2 * 4.0 / (rho * 1250.0)) * 5
    hbc, hbv = [0, 1, 1, 0, 1, 1], [273.0, 0, 0, 1623.0, 0, 0]
    # one rank
    c1 = P3.Context3(n, grid)
    A1, rhs1 = P3.makeStokesMatrix(n, grid, es, en, rho, ctx=c1)
    y1 = A1 @ x
    x1 = P3.solve(A1)
    st1 = A1.last_stats
    H1, _ = P3.makeDiffusionMatrix(n, grid, mid, Tn, [k, k, k], cp, rho, H, hbc, hbv, dt, ctx=c1)
    T1 = P3.solve_heat(H1); ht1 = H1.last_stats
    lev1 = A1.mg_info()[0]
    c1.close()
    assert st1["converged"] == 1 and ht1["converged"] == 1
    # blocks
    vc = P3.VirtualCluster3(n, grid, *layout)

    def run(ctx, rank):
        A, rhs = P3.makeStokesMatrix(n, grid, es, en, rho, ctx=ctx)
        y = A @ x
        xs = P3.solve(A)
        Hm, _ = P3.makeDiffusionMatrix(n, grid, mid, Tn, [k, k, k], cp, rho, H, hbc, hbv, dt, ctx=ctx)
        T = P3.solve_heat(Hm)
        return rhs, y, xs, A.last_stats, T, Hm.last_stats, A.mg_info()[0], ctx.comm_stats()
    res = vc.all(run)
    vc.close()
    for rhs, y, xs, st, T, ht, lev, cs in res:
        assert np.array_equal(rhs, rhs1)
        assert np.abs(y - y1).max() <= 1e-13 * np.abs(y1).max()
        # blocks are halved with the grid: the hierarchy is the one-rank one as long as every block keeps an even number of cells
        # (all cases but the last, whose 5-cell blocks end it one level early: more sweeps on the coarsest level, a few more iterations)
        assert lev <= lev1 and (lev == lev1 or n == [33, 41, 49]) and st["converged"] == 1 and ht["converged"] == 1
        assert abs(st["iterations"] - st1["iterations"]) <= (3 if lev == lev1 else 8), (st, st1)       # (BiCGStab counts move by one or two with the summation order of the dot products)
        (v, p), (v1, p1) = P3.x2vp(xs, n), P3.x2vp(x1, n)
        ev = np.sqrt(sum(np.sum((a - b) ** 2) for a, b in zip(v, v1)) / sum(np.sum(b ** 2) for b in v1))
        assert ev < 5e-8 and relerr(p, p1) < 1e-7, (ev, relerr(p, p1))     # (both solves stop at an estimated error of 3e-8)
        assert relerr(T, T1) < 1e-10 and abs(ht["iterations"] - ht1["iterations"]) <= 1
        assert cs[0] > 0 and cs[1] > 0                    # halo exchanges and all-reduces happened
