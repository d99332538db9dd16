"""The marker kernels the TIMED step launches, pinned directly.

The module-level trac2grid / grid2trac / RK of pylamp_amd.pylamp_trac take host arrays and run the generic kernels
(k_scatter_atomic, k_gather<false,0>, k_rk4<false>).  The resident step runs other kernels on its cell-sorted tracers:
    k_scatter_cells<6,1,true,IND> / <2,2,false> / <1,0,false>   (fused: all four staggered target sets in one pass; IND: the constant
                                                                 columns read through the slot index of the epoch layout)
    k_gather<true,0|1|2>                                     (regular-grid bilinear gather, subgrid-diffusion epilogues)
    k_rk4<true>
Simulation.scatter_fields / temp_to_tracers / advect (C ABI pl_resident_*) run exactly those stage functions one at a
time, and this file compares them with the REFERENCE's fixtures (tests/golden/trac2grid.npz, grid2trac.npz, rk4.npz:
outputs of pylamp_trac.py:161-318, :30-158, :321-388) and, at 2049^2 with 4.2 M tracers, with the oracle.
"""
import os

import numpy as np
import pytest

from conftest import golden, maxrel, pointrel, relerr

pytestmark = pytest.mark.gpu

TR_RHO, TR_ETA, TR_MRK, TR_TMP, TR_HCD, TR_HCP, TR_RH0, TR_ALP, TR_MAT, TR_ACE, TR_ET0, TR_IHT, TR__ID = range(13)
# GEOM-mean viscosities: exp(sum w ln eta / sum w) at EVERY node to this relative error (|ln eta| ~ 50, so a few 1e-14 are
# inherent); the other fields keep conftest.maxrel (relative to the field's maximum)
ETA_TOL = 1e-11


def _close(name, a, b, tol):
    return (pointrel(a, b) < max(tol, ETA_TOL)) if name in ("etas", "etan") else (maxrel(a, b) < tol)


def _tracers(tr_x, cols):
    tr_f = np.zeros((tr_x.shape[0], 13))
    tr_f[:, TR__ID] = np.arange(tr_x.shape[0])
    for k, v in cols.items():
        tr_f[:, k] = v
    return tr_f


def _targets(oracle, nx, L):
    grid = [np.linspace(0, L[i], nx[i]) for i in range(2)]
    gmp = oracle.gridmp_of(grid)
    return {"nodes": grid, "centres": gmp, "zmid": [gmp[0], grid[1]], "xmid": [grid[0], gmp[1]]}


def _sim(nx, L, tr_x, tr_f, **kw):
    from pylamp_amd import driver
    opt = driver.Options(tdep_rho=False, tdep_eta=False, **kw)          # rho = RH0, eta = ET0: the fixture's columns as they are
    return driver.Simulation(nx, L, tr_x, tr_f, opt)


@pytest.mark.parametrize("case", ["dense", "sparse", "outside"])
def test_fused_scatter_vs_reference_fixture(oracle, case):
    """k_scatter_cells<6,1,true> on the reference's trac2grid fixture: columns 2 / 3 of the fixture carry schemes 5 (ARITHW) and
    6 (GEOMW) -- the two the step uses -- with expected values for all four target sets.  'sparse' has empty nodes (NaN masks
    must be identical), 'outside' has markers beyond the walls (they leave their sort cell: the kernel's list path)."""
    g = golden("trac2grid")
    nx = [int(v) for v in g["nx"]]; L = [float(v) for v in g["L"]]
    assert [int(s) for s in g["schemes"]] == [1, 2, 5, 6]
    X = g[case + "_tr_x"]; F = g[case + "_tr_f"]
    tr_f = _tracers(X, {TR_RH0: F[:, 2], TR_ET0: F[:, 3], TR_HCD: F[:, 2], TR_HCP: F[:, 0], TR_TMP: 300 + F[:, 0], TR_IHT: 1e-9 * F[:, 2] * F[:, 0],
                        TR_MAT: np.round(F[:, 0])})
    sim = _sim(nx, L, X, tr_f)
    got = sim.scatter_fields()
    tg = _targets(oracle, nx, L)
    # against the reference's own output
    assert maxrel(got["rho"], g[case + "_nodes"][2]) < 1e-12
    assert pointrel(got["etas"], g[case + "_nodes"][3]) < ETA_TOL           # (pointwise: the viscosities span decades)
    assert pointrel(got["etan"], g[case + "_centres"][3]) < ETA_TOL
    assert maxrel(got["kz"], g[case + "_zmid"][2]) < 1e-12
    assert maxrel(got["kx"], g[case + "_xmid"][2]) < 1e-12
    # the other node fields against the (pinned) oracle
    ref = oracle.trac2grid(X, tr_f[:, [TR_HCP, TR_TMP, TR_IHT, TR_MAT]], tg["nodes"], nx, [5, 5, 5, 5])
    for name, r in zip(("cp", "f_T", "H", "mat"), ref):
        assert maxrel(got[name], r) < 1e-12, name
    # the one-set-per-pass kernels (k_scatter_binned) give the same fields
    os.environ["PYLAMP_SCATTER"] = "0"
    try:
        old = sim.scatter_fields()
    finally:
        del os.environ["PYLAMP_SCATTER"]
    for k in got:
        assert _close(k, got[k], old[k], 1e-12), k
    sim.close()


@pytest.mark.parametrize("case", ["dense", "sparse"])
def test_fused_scatter_heat_off_vs_reference_fixture(oracle, case):
    """k_scatter_cells<2,2,false>: the advect-only branch (pylamp2.py:316-319) takes the UNWEIGHTED geometric mean for the
    centre viscosity -- scheme 2 of the fixture, column 1."""
    g = golden("trac2grid")
    nx = [int(v) for v in g["nx"]]; L = [float(v) for v in g["L"]]
    X = g[case + "_tr_x"]; F = g[case + "_tr_f"]
    sim = _sim(nx, L, X, _tracers(X, {TR_RH0: F[:, 2], TR_ET0: F[:, 1]}), do_heatdiff=False)
    got = sim.scatter_fields()
    assert maxrel(got["rho"], g[case + "_nodes"][2]) < 1e-12
    assert pointrel(got["etan"], g[case + "_centres"][1]) < ETA_TOL               # unweighted GEOM, the reference's output
    etas, = oracle.trac2grid(X, F[:, [1]], _targets(oracle, nx, L)["nodes"], nx, [6])
    assert pointrel(got["etas"], etas) < ETA_TOL
    sim.close()


def test_fast_gather_vs_reference_fixture():
    """k_gather<true,0> (it == 1 branch of pylamp2.py:436-455) on the reference's grid2trac fixture, LINEAR."""
    g = golden("grid2trac")
    nx = [int(v) for v in g["nx"]]; L = [float(v) for v in g["L"]]
    X = g["inside"]
    sim = _sim(nx, L, X, _tracers(X, {TR_RH0: 1.0, TR_ET0: 1.0, TR_HCP: 1.0, TR_HCD: 1.0}))
    for k in range(2):
        T = sim.temp_to_tracers(g["F"][k], 1.0, first=True)
        assert maxrel(T, g["inside_linear"][:, k]) < 1e-13
    sim.close()
    # markers outside the grid: the step interpolates with stopOnError=True (pylamp2.py:445) -> exception
    Xm = g["mixed"]
    sim = _sim(nx, L, Xm, _tracers(Xm, {TR_RH0: 1.0, TR_ET0: 1.0}))
    with pytest.raises(Exception, match="stopOnError"):
        sim.temp_to_tracers(g["F"][0], 1.0, first=True)
    sim.close()


def _subgrid_expected(oracle, X, tr_f, grid, nx, L, f_T, newtemp, tstep, subgrid):
    """pylamp2.py:453-480 restated with the oracle's (pinned) trac2grid / grid2trac."""
    dx = [L[i] / (nx[i] - 1) for i in range(2)]
    old_T = tr_f[:, TR_TMP].copy()
    T = old_T + oracle.grid2trac(X, grid, [newtemp - f_T], nx, stop_on_error=True)[:, 0]
    if not subgrid:
        return T
    dt0 = tr_f[:, TR_HCP] * tr_f[:, TR_RHO] / (tr_f[:, TR_HCD] * ((2 / dx[1]) ** 2 + (2 / dx[0]) ** 2))
    Tsub = old_T - (old_T - T) * np.exp(-0.5 * tstep / dt0)
    dTs = Tsub - T
    sgc, = oracle.trac2grid(X, dTs[:, None], grid, nx, [5])
    back = oracle.grid2trac(X, grid, [sgc], nx, stop_on_error=True)[:, 0]
    return Tsub - back


@pytest.mark.parametrize("subgrid", [True, False])
def test_subgrid_epilogue_gathers_vs_oracle(oracle, subgrid):
    """k_gather<true,1> and <true,2> with the fused subgrid-diffusion formulas (pylamp2.py:471-480) and the one-field
    fused scatter k_scatter_cells<1,0,false> between them."""
    rng = np.random.default_rng(7)
    nx = [65, 49]; L = [660e3, 495e3]
    n = nx[0] * nx[1] * 12
    X = rng.random((n, 2)) * np.array(L)
    cols = {TR_RH0: rng.uniform(3200, 3400, n), TR_ET0: 10 ** rng.uniform(19, 22, n), TR_HCP: rng.uniform(1000, 1300, n),
            TR_HCD: rng.uniform(2, 5, n), TR_TMP: 273 + 1350 * X[:, 0] / L[0] + rng.uniform(-30, 30, n), TR_IHT: 1e-11, TR_MAT: 1.0}
    tr_f = _tracers(X, cols)
    sim = _sim(nx, L, X, tr_f, do_subgrid_heatdiff=subgrid)
    fields = sim.scatter_fields()
    tr_f[:, TR_RHO] = tr_f[:, TR_RH0]                               # what the property update left on the device
    grid = [np.linspace(0, L[0], nx[0]), np.linspace(0, L[1], nx[1])]
    Z, Xg = np.meshgrid(*grid, indexing="ij")
    newtemp = fields["f_T"] + 15 * np.sin(3 * np.pi * Xg / L[1]) * np.sin(2 * np.pi * Z / L[0])
    tstep = 0.3 * (L[0] / (nx[0] - 1)) ** 2 * 3300 * 1250 / 4.0     # a diffusive step: exp(-dt/dt0) well inside (0, 1)
    got = sim.temp_to_tracers(newtemp, tstep, first=False)
    exp = _subgrid_expected(oracle, X, tr_f, grid, nx, L, fields["f_T"], newtemp, tstep, subgrid)
    assert maxrel(got, exp) < 1e-12
    sim.close()


def test_fast_rk4_vs_reference_fixture():
    """k_rk4<true> on the reference's RK fixture (pylamp_trac.py:321-388; the second data set has stages that leave the grid)."""
    g = golden("rk4")
    nx = [int(v) for v in g["nx"]]; L = [float(v) for v in g["L"]]
    X = g["tr"]
    for Vz, Vx, dt, x_ref, v_ref in ((g["Vz"], g["Vx"], float(g["tstep"]), g["x1"], g["v1"]),
                                      (g["Vz2"], g["Vx2"], 4 * float(g["tstep"]), g["x2"], g["v2"])):
        sim = _sim(nx, L, X, _tracers(X, {TR_RH0: 1.0, TR_ET0: 1.0}))
        gz = np.concatenate([[-0.5 * L[0] / (nx[0] - 1)], 0.5 * (sim.grid[0][1:] + sim.grid[0][:-1]), [L[0] + 0.5 * L[0] / (nx[0] - 1)]])
        assert np.allclose(gz, g["gz"], rtol=1e-14, atol=1e-9)      # the padded centre grid the device builds is the fixture's
        v, x = sim.advect(Vz, Vx, dt, fence=False)
        assert maxrel(x, x_ref) < 1e-14 and maxrel(v, v_ref) < 1e-9
        sim.close()


def test_resident_mic_kernels_2049_vs_oracle(oracle):
    """The same kernels on the bench grid (2049^2) with 4.2 M tracers at the bench density (16 per cell inside a 512^2-cell
    window, nothing elsewhere: dense strips, empty strips and NaN nodes in one case) against the oracle: all four target
    sets, the subgrid epilogue and RK4."""
    rng = np.random.default_rng(2049)
    n = 2049; nx = [n, n]; L = [660e3, 660e3]
    h = L[0] / (n - 1)
    m = 4_200_000
    X = (700 + 512 * rng.random((m, 2))) * h                           # cells 700 .. 1211 in both directions
    T0 = 273 + 1350 * X[:, 0] / L[0] + 20 * np.sin(40 * np.pi * X[:, 1] / L[1])
    cols = {TR_RH0: 3300 + 50 * np.sin(X[:, 0] / 3e3), TR_ET0: 1e19 * 10 ** (2 * np.cos(X[:, 1] / 5e3) ** 2), TR_HCP: 1250 + 50 * np.cos(X[:, 0] / 7e3),
            TR_HCD: 4 + np.sin(X[:, 1] / 2e3), TR_TMP: T0, TR_IHT: 6e-12 * (1 + 0.1 * np.sin(X[:, 0] / 1e3)), TR_MAT: 1.0 + (X[:, 0] > 0.45 * L[0])}
    tr_f = _tracers(X, cols)
    sim = _sim(nx, L, X, tr_f)
    got = sim.scatter_fields()
    tg = _targets(oracle, nx, L)
    ref = dict(zip(("rho", "etas", "cp", "f_T", "H", "mat"),
                   oracle.trac2grid(X, tr_f[:, [TR_RH0, TR_ET0, TR_HCP, TR_TMP, TR_IHT, TR_MAT]], tg["nodes"], nx, [5, 6, 5, 5, 5, 5])))
    ref["etan"], = oracle.trac2grid(X, tr_f[:, [TR_ET0]], tg["centres"], nx, [6])
    ref["kz"], = oracle.trac2grid(X, tr_f[:, [TR_HCD]], tg["zmid"], nx, [5])
    ref["kx"], = oracle.trac2grid(X, tr_f[:, [TR_HCD]], tg["xmid"], nx, [5])
    for k, r in ref.items():
        assert _close(k, got[k], r, 1e-11), k
    # temperature to tracers with subgrid diffusion (the NaN nodes outside the window are never read: every marker sits inside)
    tr_f[:, TR_RHO] = tr_f[:, TR_RH0]
    grid = tg["nodes"]
    f_T = got["f_T"]
    Z, Xg = np.meshgrid(*grid, indexing="ij")
    newtemp = np.where(np.isnan(f_T), 0.0, f_T) + 10 * np.sin(60 * np.pi * Xg / L[1]) * np.sin(50 * np.pi * Z / L[0])
    f_T0 = np.where(np.isnan(f_T), 0.0, f_T)
    tstep = 0.3 * h ** 2 * 3300 * 1250 / 4.0
    # the device keeps the NaN of the empty nodes in its f_T plane; newtemp - f_T is NaN there as well, and no marker reads it
    Tg = sim.temp_to_tracers(newtemp, tstep, first=False)
    exp = _subgrid_expected(oracle, X, tr_f, grid, nx, L, f_T0, newtemp, tstep, True)
    assert maxrel(Tg, exp) < 1e-11
    # RK4 through a smooth solenoidal field on the padded centre grid
    gz = np.concatenate([[-0.5 * h], 0.5 * (grid[0][1:] + grid[0][:-1]), [L[0] + 0.5 * h]])
    gx = np.concatenate([[-0.5 * h], 0.5 * (grid[1][1:] + grid[1][:-1]), [L[1] + 0.5 * h]])
    Zp, Xp = np.meshgrid(gz, gx, indexing="ij")
    Vz = 1e-9 * np.sin(np.pi * Zp / L[0]) * np.cos(7 * np.pi * Xp / L[1]); Vx = -1e-9 / 7 * np.cos(np.pi * Zp / L[0]) * np.sin(7 * np.pi * Xp / L[1])
    dt = 0.67 * h / 1e-9
    v, x = sim.advect(Vz, Vx, dt, fence=True)
    v0, x0 = oracle.rk4(X, [gz, gx], [Vz, Vx], nx, dt)
    assert relerr(x, x0) < 1e-14 and relerr(v, v0) < 1e-9
    # after the sort the census is that of the moved markers
    ci = np.minimum((x[:, 0] / h).astype(np.int64), n - 2); cj = np.minimum((x[:, 1] / h).astype(np.int64), n - 2)
    assert np.array_equal(sim.census(), np.bincount(ci * (n - 1) + cj, minlength=(n - 1) ** 2).reshape(n - 1, n - 1))
    # ---- the same stages in the EPOCH layout, which is what the timed step runs from its second step on: after a sort that is not
    # followed by a download the constant columns stay where they were and k_property_update, k_scatter_cells<6,1,true,true> and the
    # gather epilogue read them through the slot index (pl_step.hip)
    assert sim.layout() == (0, 0)                                     # the downloads above have closed the epoch
    sim.advect(Vz, Vx, 0.5 * dt, fence=True, download=False)
    assert sim.layout()[0] == 1 and sim.layout()[1] == 1
    got = sim.scatter_fields()
    assert sim.layout()[0] == 1                                       # (the scatter itself leaves the layout alone)
    X1, F1 = sim.tracers()
    ref = dict(zip(("rho", "etas", "cp", "f_T", "H", "mat"),
                   oracle.trac2grid(X1, F1[:, [TR_RH0, TR_ET0, TR_HCP, TR_TMP, TR_IHT, TR_MAT]], tg["nodes"], nx, [5, 6, 5, 5, 5, 5])))
    ref["etan"], = oracle.trac2grid(X1, F1[:, [TR_ET0]], tg["centres"], nx, [6])
    ref["kz"], = oracle.trac2grid(X1, F1[:, [TR_HCD]], tg["zmid"], nx, [5])
    ref["kx"], = oracle.trac2grid(X1, F1[:, [TR_HCD]], tg["xmid"], nx, [5])
    for k, r in ref.items():
        assert _close(k, got[k], r, 1e-11), k
    assert np.array_equal(F1[:, [TR_HCD, TR_HCP, TR_RH0, TR_ET0, TR_IHT, TR_MAT, TR__ID]], tr_f[:, [TR_HCD, TR_HCP, TR_RH0, TR_ET0, TR_IHT, TR_MAT, TR__ID]])
    sim.advect(0 * Vz, 0 * Vx, dt, fence=True, download=False)       # a sort that moves nobody: a new epoch on the same positions
    assert sim.layout()[0] == 1
    got = sim.scatter_fields()
    f_T1 = np.where(np.isnan(got["f_T"]), 0.0, got["f_T"])
    newtemp1 = f_T1 + 10 * np.sin(60 * np.pi * Xg / L[1]) * np.sin(50 * np.pi * Z / L[0])
    Tg = sim.temp_to_tracers(newtemp1, tstep, first=False)
    F1[:, TR_RHO] = F1[:, TR_RH0]
    exp = _subgrid_expected(oracle, X1, F1, grid, nx, L, f_T1, newtemp1, tstep, True)
    assert maxrel(Tg, exp) < 1e-11
    sim.close()
