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
    # the mantle fixture is almost hydrostatic (velocities are a response to tracer-sampling noise):
    # its velocity error sits close to 1e-6 at the default 1e-10, so the test solves one digit deeper
    opt = driver.Options(do_heatdiff=heat, tdep_rho=heat, tdep_eta=heat, stokes_rtol=1e-11 if heat else 1e-10)
    sim = driver.Simulation(nx, L, g["init_tr_x"], g["init_tr_f"], opt)
    for it in range(1, int(g["nsteps"]) + 1):
        rep = sim.step()
        p = "s%d_" % it
        assert rep["stokes"]["converged"] == 1, rep
        # velocity/temperature: the BASELINE tolerance 1e-6 vs the reference's direct solves
        assert relerr(sim.field("velz"), g[p + "velz"]) < 1e-6
        assert relerr(sim.field("velx"), g[p + "velx"]) < 1e-6
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


def test_census_and_injection():
    """Cells that fall below tracdens_min are refilled to tracdens with tracers carrying the cell
    mean of every field (pylamp2.py:588-633); IDs stay unique; nothing else is touched."""
    from pylamp_amd import driver
    nx = [41, 41]; L = [660e3, 660e3]
    rng = np.random.default_rng(21)
    tr_x, tr_f = driver.falling_block_tracers(nx, L, 8, rng)
    h = L[0] / 40
    # deplete a patch of cells: keep at most one tracer per cell there
    ci = np.floor(tr_x[:, 0] / h).astype(int); cj = np.floor(tr_x[:, 1] / h).astype(int)
    patch = (ci >= 5) & (ci < 12) & (cj >= 20) & (cj < 30)
    key = ci * 40 + cj
    first = np.zeros(tr_x.shape[0], bool)
    _, idx = np.unique(key[patch], return_index=True)
    first[np.nonzero(patch)[0][idx]] = True
    keep = ~patch | first
    tr_x, tr_f = tr_x[keep], tr_f[keep]
    n0 = tr_x.shape[0]
    opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, tracdens=8, tracdens_min=3)
    sim = driver.Simulation(nx, L, tr_x, tr_f, opt)
    rep = sim.step()
    assert rep["ninjected"] > 0 and rep["ntrac"] == n0 + rep["ninjected"]
    X, F = sim.tracers()
    assert X.shape[0] == rep["ntrac"]
    old, new = slice(0, n0), slice(n0, None)               # injected tracers are appended
    assert np.array_equal(F[old, 12], tr_f[:, 12])          # resident tracers keep order and identity
    ids = F[:, 12]
    assert np.unique(ids).size == ids.size and ids[new].min() > tr_f[:, 12].max()
    ci = np.clip(np.floor(X[:, 0] / h).astype(int), 0, 39); cj = np.clip(np.floor(X[:, 1] / h).astype(int), 0, 39)
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


def test_surface_stabilisation_loop_reports_honestly():
    """The surfstab re-solve loop (pylamp2.py:387-405) runs on the device, but the stabilised system
    is NOT yet within reach of the preconditioner (it ignores the stabilisation terms; a re-solve
    exceeds maxit).  Until that is fixed the step must say so instead of returning silently wrong
    fields: converged == 0 is reported.  (The oracle reproduces the reference trajectory:
    tests/test_oracle_golden.py::test_trajectory_surface_stabilisation.)"""
    from pylamp_amd import driver
    g = golden("traj_surfstab41")
    gz, gx = g["gz"], g["gx"]
    nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
    opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, surface_stabilization=True)
    sim = driver.Simulation(nx, L, g["init_tr_x"], g["init_tr_f"], opt)
    rep = sim.step()
    assert rep["stokes_resolves"] >= 1
    ok = relerr(sim.field("velz"), g["s1_velz"]) < 1e-6
    assert ok or rep["stokes"]["converged"] == 0, rep
    sim.close()
