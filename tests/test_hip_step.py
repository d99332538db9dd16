"""GPU parity of the device-resident time step against the trajectories produced by the
reference's own driver (pylamp2.py exec'd by oracle/gen_golden.py)."""
import os

import numpy as np
import pytest

from conftest import golden, relerr

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,heat", [("traj_block41", False), ("traj_mantle33x41", True)])
def test_trajectory_vs_reference_driver(name, heat, tmp_path):
    from pylamp_amd import driver
    g = golden(name)
    gz, gx = g["gz"], g["gx"]
    nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
    # default tolerances: the same ones bench.py times
    opt = driver.Options(do_heatdiff=heat, tdep_rho=heat, tdep_eta=heat)
    sim = driver.Simulation(nx, L, g["init_tr_x"], g["init_tr_f"], opt)
    for it in range(1, int(g["nsteps"]) + 1):
        rep = sim.step()
        p = "s%d_" % it
        assert rep["stokes"]["converged"] == 1, rep
        # velocity/temperature: the BASELINE tolerance 1e-6 vs the reference's direct solves
        ez, ex = relerr(sim.field("velz"), g[p + "velz"]), relerr(sim.field("velx"), g[p + "velx"])
        assert ez < 1e-6 and ex < 1e-6, (it, ez, ex, rep["stokes"])
        assert relerr(sim.field("rho"), g[p + "rho"]) < 1e-7   # tracer positions carry the solver tolerance
        assert abs(sim.totaltime - float(g[p + "time"])) < 1e-6 * sim.totaltime
        tr_x, tr_f = sim.tracers()
        assert relerr(tr_x, g[p + "tr_x"]) < 1e-7
        assert relerr(sim.tracer_velocity(), g[p + "tr_v"]) < 1e-5
        if heat:
            assert relerr(sim.field("temp"), g[p + "temp"]) < 1e-6
            assert relerr(tr_f[:, 3], g[p + "tr_T"]) < 1e-6
            assert rep["heat"]["converged"] == 1
    # snapshot files carry the reference's keys (pylamp2.py:637-650)
    sim.write_snapshot(str(tmp_path))
    gd = np.load(os.path.join(str(tmp_path), "griddata.%06d.npz" % sim.it))
    assert sorted(gd.files) == sorted(["gridz", "gridx", "velz", "velx", "pres", "rho", "temp", "tstep", "time"])
    tc = np.load(os.path.join(str(tmp_path), "tracs.%06d.npz" % sim.it))
    assert sorted(tc.files) == sorted(["tr_x", "tr_f", "tr_v", "tstep", "time"])
    sim.close()


def test_step_vs_oracle_midsize(oracle):
    """129x97 mantle model, 2 steps, every field against the oracle's step()."""
    from pylamp_amd import driver
    nx = [129, 97]; L = [660e3, 495e3]
    rng = np.random.default_rng(3)
    tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng)
    sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options())
    st = dict(nx=nx, L=L, grid=[np.linspace(0, L[0], nx[0]), np.linspace(0, L[1], nx[1])], tr_x=tr_x.copy(),
              tr_f=tr_f.copy())
    cfg = oracle.StepConfig()
    for it in (1, 2):
        rep = sim.step()
        out = oracle.step(st, cfg, it)
        assert rep["limiter"] == out["limiter"]
        assert rep["tstep"] == pytest.approx(out["tstep"], rel=1e-6)
        assert relerr(sim.field("velz"), out["velz"]) < 1e-6 and relerr(sim.field("velx"), out["velx"]) < 1e-6
        assert relerr(sim.field("temp"), out["temp"]) < 1e-6
        assert relerr(sim.field("etas"), out["etas"]) < 1e-6
        X, F = sim.tracers()
        assert relerr(X, st["tr_x"]) < 1e-7 and relerr(F[:, 3], st["tr_f"][:, 3]) < 1e-6
    sim.close()


def _cells(X, nx, L):
    ci = np.floor((nx[0] - 1) * X[:, 0] / L[0]).astype(int); cj = np.floor((nx[1] - 1) * X[:, 1] / L[1]).astype(int)
    return ci * (nx[1] - 1) + cj


@pytest.mark.parametrize("name,heat,dens,dmin", [("traj_inject41", False, 12, 9), ("traj_inject_mantle25x33", True, 14, 10)])
def test_census_and_injection_vs_reference_driver(name, heat, dens, dmin):
    """pylamp2.py:588-633 against the stock driver run with tracdens_min > 0 (oracle/gen_golden.py, state captured
    after the injection).  Everything but the random positions is compared: number injected, which cells are
    refilled and by how many, append order, the reference's ID rule, the cell-mean fields, the census afterwards.
    The positions come from a different generator, so every step starts from the reference's post-injection state
    of the step before (block model; the heat model compares its first step)."""
    from pylamp_amd import driver
    g = golden(name)
    gz, gx = g["gz"], g["gx"]
    nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
    opt = driver.Options(do_heatdiff=heat, tdep_rho=heat, tdep_eta=heat, tracdens=dens, tracdens_min=dmin)
    start_x, start_f = g["init_tr_x"], g["init_tr_f"]
    total = 0
    for it in range(1, (1 if heat else int(g["nsteps"])) + 1):
        sim = driver.Simulation(nx, L, start_x, start_f, opt)
        rep = sim.step()
        p, q = "s%d_" % it, "p%d_" % it
        ref_x, ref_f = g[q + "tr_x"], g[q + "tr_f"]
        n_old = g[p + "tr_x"].shape[0]
        assert rep["stokes"]["converged"] == 1
        assert relerr(sim.field("velz"), g[p + "velz"]) < 1e-6 and relerr(sim.field("velx"), g[p + "velx"]) < 1e-6
        assert rep["ninjected"] == ref_x.shape[0] - n_old and rep["ntrac"] == ref_x.shape[0] and rep["nremoved"] == 0
        total += rep["ninjected"]
        X, F = sim.tracers()
        assert relerr(X[:n_old], ref_x[:n_old]) < 1e-7                           # resident tracers: the advection
        assert np.allclose(F[:n_old], ref_f[:n_old], rtol=1e-6, atol=0)
        # injected tracers: appended in the reference's order (ascending cell number), same cells, same IDs
        assert np.array_equal(_cells(X[n_old:], nx, L), _cells(ref_x[n_old:], nx, L))
        assert np.array_equal(F[n_old:, 12], ref_f[n_old:, 12])
        ids = F[n_old:, 12]
        assert ids.size - np.unique(ids).size == np.unique(_cells(ref_x[n_old:], nx, L)).size - 1   # one repeat per cell
        cols = [c for c in range(13) if c != 12]
        assert np.allclose(F[n_old:][:, cols], ref_f[n_old:][:, cols], rtol=1e-6, atol=0, equal_nan=True)
        cen = sim.census().ravel()
        assert np.array_equal(cen, np.bincount(_cells(ref_x, nx, L), minlength=cen.size))
        assert cen.min() >= dmin
        # the snapshot holds the pre-injection state, like the reference's (pylamp2.py:599-600,648)
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            sim.write_snapshot(td)
            tc = np.load(os.path.join(td, "tracs.%06d.npz" % sim.it))
            assert tc["tr_x"].shape[0] == n_old and np.array_equal(tc["tr_f"][:, 12], g[p + "tr_id"])
        sim.close()
        start_x, start_f = ref_x, ref_f
    assert total > 100


def test_deletion_vs_reference_driver():
    """pylamp2.py:563-581 with the fence off: 70 tracers start beyond the low walls, get TR__ID = -1 after the first
    advection and are deleted; the survivors keep their order.  Two steps against the stock driver."""
    from pylamp_amd import driver
    g = golden("traj_delete41")
    gz, gx = g["gz"], g["gx"]
    nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
    opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, tracs_fence_enabled=False)
    sim = driver.Simulation(nx, L, g["init_tr_x"], g["init_tr_f"], opt)
    n0 = g["init_tr_x"].shape[0]
    for it, nrem in ((1, 70), (2, 0)):
        rep = sim.step()
        p = "s%d_" % it
        assert rep["stokes"]["converged"] == 1 and rep["nremoved"] == nrem and rep["ntrac"] == n0 - 70
        assert relerr(sim.field("velz"), g[p + "velz"]) < 1e-6 and relerr(sim.field("velx"), g[p + "velx"]) < 1e-6
        X, F = sim.tracers()
        assert X.shape == g[p + "tr_x"].shape and np.array_equal(F[:, 12], g[p + "tr_id"])       # order kept
        assert relerr(X, g[p + "tr_x"]) < 1e-7
        assert relerr(sim.tracer_velocity(), g[p + "tr_v"]) < 1e-5
    sim.close()


@pytest.mark.parametrize("graded", [False, True])
def test_census_and_injection(graded):
    """Cells that fall below tracdens_min are refilled to tracdens with tracers carrying the cell
    mean of every field (pylamp2.py:588-633); IDs stay unique; nothing else is touched."""
    from pylamp_amd import driver
    nx = [41, 41]; L = [660e3, 660e3]
    rng = np.random.default_rng(21)
    tr_x, tr_f = driver.falling_block_tracers(nx, L, 8, rng)
    if graded:          # refined towards the top-left: cells differ 2.5x in size, located by per-axis search
        hh = np.linspace(1.0, 2.5, 40); c = np.concatenate([[0.0], np.cumsum(hh)]); c *= L[0] / c[-1]; c[-1] = L[0]
        grid = [c, c.copy()]
    else:
        grid = [np.linspace(0, L[0], 41), np.linspace(0, L[1], 41)]
    cell = lambda x, d: np.clip(np.searchsorted(grid[d], x, side="right") - 1, 0, 39)
    # deplete a patch of cells: keep at most one tracer per cell there
    ci = cell(tr_x[:, 0], 0); cj = cell(tr_x[:, 1], 1)
    patch = (ci >= 5) & (ci < 12) & (cj >= 20) & (cj < 30)
    key = ci * 40 + cj
    first = np.zeros(tr_x.shape[0], bool)
    _, idx = np.unique(key[patch], return_index=True)
    first[np.nonzero(patch)[0][idx]] = True
    keep = ~patch | first
    tr_x, tr_f = tr_x[keep], tr_f[keep]
    n0 = tr_x.shape[0]
    opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, tracdens=8, tracdens_min=3,
                         inject_unique_ids=True)
    sim = driver.Simulation(nx, L, tr_x, tr_f, opt, grid=grid if graded else None)
    rep = sim.step()
    assert rep["ninjected"] > 0 and rep["ntrac"] == n0 + rep["ninjected"]
    X, F = sim.tracers()
    assert X.shape[0] == rep["ntrac"]
    old, new = slice(0, n0), slice(n0, None)               # injected tracers are appended
    assert np.array_equal(F[old, 12], tr_f[:, 12])          # resident tracers keep order and identity
    ids = F[:, 12]
    assert np.unique(ids).size == ids.size and ids[new].min() > tr_f[:, 12].max()
    ci = cell(X[:, 0], 0); cj = cell(X[:, 1], 1)
    cnt_all = np.bincount(ci * 40 + cj, minlength=1600)
    cnt_old = np.bincount(ci[old] * 40 + cj[old], minlength=1600)
    deficient = cnt_old < 3
    assert deficient.any()
    assert np.all(cnt_all[deficient] == 8) and np.array_equal(cnt_all[~deficient], cnt_old[~deficient])
    # fields of the new tracers = plain mean of the resident tracers of their cell
    knew = ci[new] * 40 + cj[new]
    for col in (6, 10, 8):                                   # RH0, ET0, MAT
        s = np.bincount(ci[old] * 40 + cj[old], weights=F[old, col], minlength=1600)
        with np.errstate(invalid="ignore", divide="ignore"):
            mean = s / cnt_old
        got = F[new, col]
        exp = mean[knew]
        ok = ~np.isnan(exp)
        assert np.allclose(got[ok], exp[ok], rtol=1e-13) and np.isnan(got[~ok]).all()
    # a second step runs on the refilled set
    rep2 = sim.step()
    assert rep2["stokes"]["converged"] == 1
    sim.close()


def test_surface_stabilisation_loop_vs_reference_driver():
    """Strict (reference-sign) surfstab loop (pylamp2.py:387-405) against the trajectory of the reference's own driver
    (model 3: rising block under sticky air, dynamic stabilisation time step).  With the reference's sign the velocity
    block is indefinite at the Courant step: the multigrid-preconditioned iteration does not
    converge there, and the solve is finished by the banded-LU fallback for small systems (pl_direct.hip) -- the
    reference itself uses SuperLU.  Every step of the fixture is reproduced to the BASELINE tolerance."""
    from pylamp_amd import driver
    g = golden("traj_surfstab41")
    gz, gx = g["gz"], g["gx"]
    nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
    opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, surface_stabilization=True)
    sim = driver.Simulation(nx, L, g["init_tr_x"], g["init_tr_f"], opt)
    used = 0
    for it in range(1, int(g["nsteps"]) + 1):
        rep = sim.step()
        p = "s%d_" % it
        assert rep["stokes"]["converged"] == 1 and rep["stokes_resolves"] >= 1, rep
        used += rep["stokes"]["used_direct"]
        assert relerr(sim.field("velz"), g[p + "velz"]) < 1e-6 and relerr(sim.field("velx"), g[p + "velx"]) < 1e-6
        assert abs(sim.totaltime - float(g[p + "time"])) < 1e-6 * sim.totaltime
        X, _ = sim.tracers()
        assert relerr(X, g[p + "tr_x"]) < 1e-7
    assert used >= 1                                   # the indefinite systems did need the direct fallback
    sim.close()


def _surfstab_fields(oracle):
    g = golden("traj_surfstab41")
    gz, gx = g["gz"], g["gx"]
    nx = [gz.size, gx.size]; grid = [gz, gx]
    tr_x = g["init_tr_x"]; tr_f = g["init_tr_f"].copy()
    oracle.property_update(tr_f, False, False)
    rho, etas = oracle.trac2grid(tr_x, tr_f[:, [0, 1]], grid, nx, [5, 6])
    etan, = oracle.trac2grid(tr_x, tr_f[:, [1]], oracle.gridmp_of(grid), nx, [2])
    (vz, vx), _ = oracle.x2vp(oracle.stokes_solve(nx, grid, etas, etan, rho, [1, 1, 1, 1]), nx)
    courant = 0.67 * (gz[1] - gz[0]) / max(vz.max(), vx.max())
    return nx, grid, etas, etan, rho, courant


@pytest.mark.parametrize("strict,dtfac", [(False, 1.0), (False, 0.25), (True, 0.05)])
def test_stabilised_solve_vs_oracle(oracle, strict, dtfac):
    """Sticky-air model 3 (air 1e18 Pa s / 1000 kg m^-3 over rock): the stabilised system against the oracle's
    direct solve.  strict=False is the damping sign; the reference's own sign is only solvable iteratively while
    the terms stay below the viscous diagonal (here 5 % of the Courant step) -- at the full Courant step 38 rows
    of the 41^2 velocity block change sign and the block is indefinite (see the next test)."""
    from pylamp_amd import pylamp_stokes as S
    nx, grid, etas, etan, rho, courant = _surfstab_fields(oracle)
    bc = [1, 1, 1, 1]; tstep = dtfac * courant
    A, rhs = S.makeStokesMatrix(nx, grid, etas, etan, rho, bc, surfstab=True, tstep=tstep, surfstab_theta=0.5,
                                strict_reference=strict)
    x = S.solve(A, rhs)
    assert A.last_stats["converged"] == 1, A.last_stats
    xr = oracle.stokes_solve(nx, grid, etas, etan, rho, bc, surfstab=True, tstep=tstep, theta=0.5 if strict else -0.5)
    (vz, vx), _ = S.x2vp(x, nx); (rz, rx), _ = oracle.x2vp(xr, nx)
    err = np.sqrt((np.sum((vz - rz) ** 2) + np.sum((vx - rx) ** 2)) / (np.sum(rz ** 2) + np.sum(rx ** 2)))
    assert err < 1e-6, err


def test_reference_sign_amplifies_instead_of_damping(oracle):
    """Why the reference-sign loop cannot be handed to an iterative solver: with the sign of pylamp_stokes.py:
    422-426 the 'stabilised' solution is FASTER than the unstabilised one (x230 at the Courant step in this
    model), the flipped sign damps it.  Pure oracle arithmetic, kept next to the GPU tests it explains."""
    nx, grid, etas, etan, rho, courant = _surfstab_fields(oracle)
    bc = [1, 1, 1, 1]
    v = {}
    for name, kw in (("off", {}), ("ref", dict(surfstab=True, tstep=courant, theta=0.5)),
                     ("flip", dict(surfstab=True, tstep=courant, theta=-0.5))):
        (vz, vx), _ = oracle.x2vp(oracle.stokes_solve(nx, grid, etas, etan, rho, bc, **kw), nx)
        v[name] = np.abs(vz).max()
    assert v["ref"] > 50 * v["off"] and v["flip"] < v["off"]


def test_corrected_stabilisation_loop_vs_oracle(oracle):
    """The re-solve loop (pylamp2.py:387-405) with the damping sign, every step against the oracle's step()."""
    from pylamp_amd import driver
    g = golden("traj_surfstab41")
    gz, gx = g["gz"], g["gx"]
    nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
    opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, surface_stabilization=True,
                         surfstab_strict_reference=False)
    sim = driver.Simulation(nx, L, g["init_tr_x"], g["init_tr_f"], opt)
    st = dict(nx=nx, L=L, grid=[gz, gx], tr_x=g["init_tr_x"].copy(), tr_f=g["init_tr_f"].copy())
    cfg = oracle.StepConfig(do_heatdiff=False, tdep_rho=False, tdep_eta=False, surface_stabilization=True,
                            surfstab_theta=-0.5)
    for it in (1, 2):
        rep = sim.step()
        out = oracle.step(st, cfg, it)
        assert rep["stokes"]["converged"] == 1, rep
        assert rep["stokes_resolves"] == out["nresolve"]
        assert rep["tstep"] == pytest.approx(out["tstep"], rel=1e-6)
        assert relerr(sim.field("velz"), out["velz"]) < 1e-6 and relerr(sim.field("velx"), out["velx"]) < 1e-6
        X, _ = sim.tracers()
        assert relerr(X, st["tr_x"]) < 1e-7
    sim.close()


def test_step_on_nonuniform_grid_vs_oracle(oracle):
    """SURVEY 8 f4 end to end: the resident step on a rectilinear grid (refined towards the top and the left),
    every field against the oracle's step() with per-axis cell search."""
    from pylamp_amd import driver
    nx = [65, 49]; L = [660e3, 495e3]
    rng = np.random.default_rng(4)

    def graded(n, Lx):
        h = np.linspace(1.0, 2.5, n - 1)
        c = np.concatenate([[0.0], np.cumsum(h)])
        c *= Lx / c[-1]; c[-1] = Lx
        return c
    grid = [graded(nx[0], L[0]), graded(nx[1], L[1])]
    tr_x, tr_f = driver.mantle_tracers(nx, L, 20, rng)
    sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options(tracdens=0), grid=grid)
    st = dict(nx=nx, L=L, grid=grid, tr_x=tr_x.copy(), tr_f=tr_f.copy())
    cfg = oracle.StepConfig()
    for it in (1, 2):
        rep = sim.step()
        with oracle.rect_search():
            out = oracle.step(st, cfg, it)
        assert rep["stokes"]["converged"] == 1 and rep["heat"]["converged"] == 1, rep
        assert rep["limiter"] == out["limiter"]
        assert rep["tstep"] == pytest.approx(out["tstep"], rel=1e-6)
        assert relerr(sim.field("etas"), out["etas"]) < 1e-9 and relerr(sim.field("rho"), out["rho"]) < 1e-9
        assert relerr(sim.field("velz"), out["velz"]) < 1e-6 and relerr(sim.field("velx"), out["velx"]) < 1e-6
        assert relerr(sim.field("temp"), out["temp"]) < 1e-6
        X, F = sim.tracers()
        assert relerr(X, st["tr_x"]) < 1e-7 and relerr(F[:, 3], st["tr_f"][:, 3]) < 1e-6
    sim.close()


def test_randomised_step_campaign():
    """tools/fuzz_step.py: 16 seeded random configurations (grid sizes incl. non-coarsenable ones, NOSLIP/FREESLIP
    z-walls, heat and T-dependence on/off, uniform and graded grids, block / mantle models, cell aspect 0.7-1.4),
    two resident steps each against the oracle's step(): velocities, time step, positions and temperature to 1e-5."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_step.py"), "16", "11"], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "failures: 0" in r.stdout, r.stdout[-3000:]


def test_step_with_unreachable_nodes_fails_loudly():
    """Far too few markers: some node has none in reach, its averages are 0/0 = NaN and so is the time step.  The
    reference would silently carry the NaN into the marker positions; the resident step stops with a message."""
    from pylamp_amd import driver
    nx = [41, 41]; L = [660e3, 660e3]
    rng = np.random.default_rng(2)
    tr_x, tr_f = driver.mantle_tracers(nx, L, 1, rng)
    sim = driver.Simulation(nx, L, tr_x[:300], tr_f[:300], driver.Options())
    with pytest.raises(Exception, match="time step is not finite"):
        sim.step()
    sim.close()


def test_pressure_anchor_deflation_same_fields_fewer_iterations(tmp_path):
    """The rank-1 deflation of the pressure-anchor mode in the Stokes preconditioner changes the
    iteration count, not the answer: two resident runs of the mantle model at 257 x 257, with and without it
    (PYLAMP_DEFLATE is read when a context's solver is created, hence the subprocesses)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from pylamp_amd import driver
n = 257; nx = [n, n]; L = [660e3, 660e3]
tr_x, tr_f = driver.mantle_tracers(nx, L, 12, np.random.default_rng(11))
sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options())
its = []
for k in range(3):
    rep = sim.step(); its.append(rep["stokes"]["iterations"]); assert rep["stokes"]["converged"] == 1, rep
np.save(sys.argv[1], np.stack([sim.field("velz"), sim.field("velx"), sim.field("temp")]))
print("RESULT", json.dumps(dict(its=its, est=rep["stokes"]["error_estimate"])))
''' % root
    out = {}
    for name, env in (("deflated", {}), ("plain", {"PYLAMP_DEFLATE": "0"})):
        path = str(tmp_path / ("defl_%s.npy" % name))
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, (name, r.stderr[-1500:])
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0]
        out[name] = (json.loads(line[7:]), np.load(path))
        os.remove(path)
    (sd, fd), (sp, fp) = out["deflated"], out["plain"]
    for q in range(3):
        assert relerr(fd[q], fp[q]) < 1e-6, q                  # both within the solver's error bound of the same solution
    assert sum(sd["its"]) < 0.85 * sum(sp["its"]), (sd, sp)
    assert sd["est"] <= 3e-8 and sp["est"] <= 3e-8


def test_warm_starts_same_fields_fewer_iterations(tmp_path):
    """The initial guesses of the time-step loop -- Stokes: quadratic extrapolation in model time of the last three solutions with
    the initial residual as BiCGStab's shadow vector; heat: old nodal temperature + scaled last increment --
    change the iteration counts, not the answer: two resident runs of the mantle model at 257 x 257, with and without them
    (the knobs are read once per process, hence the subprocesses)."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
from pylamp_amd import driver
n = 257; nx = [n, n]; L = [660e3, 660e3]
tr_x, tr_f = driver.mantle_tracers(nx, L, 12, np.random.default_rng(11))
sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options())
its, hits = [], []
for k in range(7):
    rep = sim.step(); its.append(rep["stokes"]["iterations"]); hits.append(rep["heat"]["iterations"])
    assert rep["stokes"]["converged"] == 1 and rep["heat"]["converged"] == 1, rep
np.save(sys.argv[1], np.stack([sim.field("velz"), sim.field("velx"), sim.field("temp")]))
print("RESULT", json.dumps(dict(its=its, hits=hits, est=rep["stokes"]["error_estimate"], time=sim.totaltime)))
''' % root
    out = {}
    off = {"PYLAMP_X0_EXTRAP": "0", "PYLAMP_HEAT_X0": "0", "PYLAMP_SHADOW": "0"}
    for name, env in (("warm", {}), ("plain", off)):
        path = str(tmp_path / ("warm_%s.npy" % name))
        r = subprocess.run([sys.executable, "-c", code, path], capture_output=True, text=True, timeout=600, env=dict(os.environ, **env))
        assert r.returncode == 0, (name, r.stderr[-1500:])
        line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0]
        out[name] = (json.loads(line[7:]), np.load(path))
        os.remove(path)
    (sw, fw), (sp, fp) = out["warm"], out["plain"]
    for q in range(3):
        assert relerr(fw[q], fp[q]) < 1e-6, q                  # both within the solvers' error bounds of the same trajectory
    assert abs(sw["time"] - sp["time"]) <= 1e-6 * sp["time"]
    assert sum(sw["its"][3:]) < sum(sp["its"][3:]), (sw, sp)   # from the fourth step on the history is complete
    assert sum(sw["hits"][2:]) < sum(sp["hits"][2:]), (sw, sp)
    assert sw["est"] <= 3e-8 and sp["est"] <= 3e-8


@pytest.mark.gpu
@pytest.mark.parametrize("heat,dens,dmin", [(True, 9, 8), (False, 9, 8), (True, 16, 15)])
def test_epoch_layout_and_lazy_columns_change_nothing(heat, dens, dmin, monkeypatch):
    """The end-of-step sort of the resident step moves only positions, temperature and a 4-byte slot per tracer
    (epoch layout: the ten columns no stage writes stay where they were, RHO / ETA / tracer velocities are moved only
    when somebody asks for them; pl_step.hip).  Against PYLAMP_EPOCH=0 (every column moves in every sort): the constant
    columns of the uploaded tracers come back IDENTICAL bit for bit, in the caller's order -- across epoch boundaries (epoch
    length 3), with injection in every step and with downloads in the middle of an epoch; everything that went through a
    Stokes solve agrees to the solver's own reproducibility (the order of the tracers inside a cell, hence the last bits
    of the marker sums, depends on the order in which the sort's atomics land).  The third case refills nearly every cell
    from 9 to 16 markers in the first step: the tracer arrays are re-allocated in the middle of the sort that opens the epoch."""
    from pylamp_amd import driver
    nx = [49, 65]; L = [660e3, 880e3]
    runs = {}
    const_cols = [2, 4, 5, 6, 7, 8, 9, 10, 11, 12]
    for epoch in ("0", "3"):
        monkeypatch.setenv("PYLAMP_EPOCH", epoch)
        tr_x, tr_f = driver.mantle_tracers(nx, L, 9, np.random.default_rng(5))
        rng = np.random.default_rng(6)                         # per-tracer constants: the indirection has something to get wrong
        for col in (2, 4, 5, 6, 7, 9, 10, 11):
            tr_f[:, col] *= rng.uniform(0.9, 1.1, tr_f.shape[0])
        tr_f[:, 2] = rng.uniform(0, 1, tr_f.shape[0])
        n0 = tr_x.shape[0]
        opt = driver.Options(do_heatdiff=heat, do_subgrid_heatdiff=heat, tracdens=dens, tracdens_min=dmin, inject_unique_ids=True)
        sim = driver.Simulation(nx, L, tr_x, tr_f, opt)
        out = []
        for it in range(8):
            rep = sim.step()
            assert rep["stokes"]["converged"] == 1
            if it in (1, 4, 7):                                # downloads at epoch age 2, 2 (after a relayout at 3) and 2
                X, F = sim.tracers()
                assert np.array_equal(F[:n0, const_cols], tr_f[:, const_cols]), (epoch, it)
                out.append((it, rep["ninjected"], X, F, sim.tracer_velocity(), sim.field("rho"), sim.field("etas")))
        assert sum(o[1] for o in out) > 0
        runs[epoch] = out
        sim.close()
    for a, b in zip(runs["0"], runs["3"]):
        assert a[0] == b[0] and a[1] == b[1]
        Fa, Fb = a[3], b[3]
        assert Fa.shape == Fb.shape
        assert np.array_equal(Fa[:, 12], Fb[:, 12])                               # IDs incl. the injected ones
        nan = np.isnan(Fa)
        assert np.array_equal(nan, np.isnan(Fb))
        assert np.allclose(Fa[~nan], Fb[~nan], rtol=1e-7, atol=0)                 # means of the injected tracers, T / rho / eta
        assert relerr(b[2], a[2]) < 1e-9 and relerr(b[4], a[4]) < 1e-5
        assert relerr(b[5], a[5]) < 1e-9 and relerr(b[6], a[6]) < 1e-7


def test_stock_model5_trajectory_vs_reference_driver(oracle):
    """Three steps of the UNMODIFIED stock driver (choose_model = 5: 201 x 41 nodes, 370 845 tracers, viscosity contrast 1e10,
    census 45 / 25) against the resident step.  The reference's own direct solve of this system is only reproducible to ~1e-4
    and the difference grows through the markers at the sphere's rim (tests/test_oracle_golden.py measures 5e-5, 5e-4, 1.8e-3
    over the three steps for the oracle that runs the SAME SuperLU): the fixture bars are those; the 1e-6 bar is against the
    accurate solve (oracle.stokes_solve_refined) of the step's OWN nodal fields.  The injected tracers' positions come from
    another generator than the reference's: every step continues from the library's own state, only the count, the cells and the
    cell-mean fields of the injection are compared."""
    from pylamp_amd import driver, pylamp_stokes as S
    g = golden("traj_model5")
    gz, gx = g["gz"], g["gx"]
    nx = [gz.size, gx.size]; L = [float(gz[-1]), float(gx[-1])]
    tr_x, tr_f = driver.sphere_tracers(nx, L, int(g["tracdens"]), int(g["seed"]))
    opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, tracdens=int(g["tracdens"]), tracdens_min=int(g["tracdens_min"]))
    sim = driver.Simulation(nx, L, tr_x, tr_f, opt)
    k = int(g["stride"])
    for it in range(1, int(g["nsteps"]) + 1):
        rep = sim.step()
        p, q = "s%d_" % it, "p%d_" % it
        assert rep["stokes"]["converged"] == 1, rep
        vz, vx = sim.field("velz"), sim.field("velx")
        # (a) the accurate solution of the system the step assembled from ITS markers
        xr = oracle.stokes_solve_refined(nx, [gz, gx], sim.field("etas"), sim.field("etan"), sim.field("rho"), [1, 1, 1, 1], refinements=4)
        (rz, rx), _ = S.x2vp(xr, nx)
        ev = np.sqrt((np.sum((vz - rz) ** 2) + np.sum((vx - rx) ** 2)) / (np.sum(rz ** 2) + np.sum(rx ** 2)))
        # (FP64 floor of this system: the banded LU + BiCGStab refinement lands 4e-7 .. 1.3e-6 from the extended-precision solution;
        #  the estimate must vouch for whatever it is)
        assert ev < 2e-6 and rep["stokes"]["error_estimate"] >= ev / 4, (it, ev, rep["stokes"])
        # (b) the reference's trajectory, to its own reproducibility
        vtol = (2e-4, 2e-3, 8e-3)[it - 1]
        assert relerr(vz, g[p + "velz"]) < vtol and relerr(vx, g[p + "velx"]) < vtol, (it, relerr(vz, g[p + "velz"]), relerr(vx, g[p + "velx"]))
        assert relerr(sim.field("rho"), g[p + "rho"]) < 1e-6
        assert abs(sim.totaltime - float(g[p + "time"])) < vtol * sim.totaltime
        n_old = int(g[p + "n"])
        assert rep["ntrac"] - rep["ninjected"] == n_old or it > 1
        assert rep["ninjected"] == g[q + "inj_x"].shape[0] and rep["nremoved"] == 0
        X, F = sim.tracers()
        if it == 1:
            assert relerr(X[:n_old][::k], g[p + "tr_x_sub"]) < 1e-5
            assert relerr(sim.tracer_velocity()[:n_old][::k], g[p + "tr_v_sub"]) < vtol
            if rep["ninjected"]:
                assert np.array_equal(_cells(X[n_old:], nx, L), _cells(g[q + "inj_x"], nx, L))
                cols = [c for c in range(13) if c != 12]
                assert np.allclose(F[n_old:][:, cols], g[q + "inj_f"][:, cols], rtol=1e-6, atol=0, equal_nan=True)
                assert np.array_equal(F[n_old:, 12], g[q + "inj_f"][:, 12])
    sim.close()
