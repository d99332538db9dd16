"""The 2-D block decomposition (BASELINE config 4's 2 x 4 layout) rehearsed on ONE GPU: Pz x Px virtual
ranks in one process (driver.VirtualCluster, the library's in-process transport).  Everything except the wire is the
multi-GPU code path: pack / unpack kernels of the 8-neighbour halo exchange with corners, replicated coarse multigrid
levels gathered from blocks, the hydrostatic prefix over block columns, reverse halo of the scatter accumulators,
velocity windows for RK4 and the 8-neighbour tracer migration."""
import numpy as np
import pytest

from conftest import golden, relerr

pytestmark = pytest.mark.gpu


def _one_rank(nx, L, tr_x, tr_f, opt, nsteps, grid=None):
    from pylamp_amd import driver
    sim = driver.Simulation(nx, L, tr_x, tr_f, opt, grid=grid)
    reps = [sim.step() for _ in range(nsteps)]
    out = dict(reps=reps, fields={k: sim.field(k) for k in ("velz", "velx", "pres", "rho", "etas", "etan", "temp")})
    X, F = sim.tracers()
    out.update(X=X, F=F, V=sim.tracer_velocity())
    sim.close()
    return out


@pytest.mark.parametrize("Pz,Px", [(2, 4), (1, 2), (2, 1), (4, 2)])
def test_blocks_equal_one_rank_mantle(Pz, Px):
    """Mantle model with heat and subgrid diffusion, 129 x 257 nodes, 3 steps: every layout reproduces the one-rank run
    (fields to solver tolerance, tracers one by one), nobody is lost or duplicated in the 8-neighbour migration."""
    from pylamp_amd import driver
    nx = [129, 257]; L = [660e3, 1320e3]
    rng = np.random.default_rng(5)
    tr_x, tr_f = driver.mantle_tracers(nx, L, 12, rng, perturb=60.0)
    opt = driver.Options()
    ref = _one_rank(nx, L, tr_x, tr_f, opt, 3)
    vc = driver.VirtualCluster(nx, L, Pz, Px, tr_x, tr_f, opt)
    vc.comm_stats(reset=True)
    for it in range(3):
        reps = vc.step()
        r0 = reps[0]
        assert all(r["stokes"]["converged"] == 1 and r["heat"]["converged"] == 1 for r in reps), reps
        assert all(r["tstep"] == r0["tstep"] and r["limiter"] == r0["limiter"] for r in reps)          # same scalars everywhere
        assert sum(r["ntrac"] for r in reps) == tr_x.shape[0]
        assert r0["limiter"] == ref["reps"][it]["limiter"] and r0["tstep"] == pytest.approx(ref["reps"][it]["tstep"], rel=1e-7)
    stats = vc.comm_stats()
    assert all(s[0] > 0 for s in stats)                          # halo exchanges happened on every rank
    assert all(s[3] == 0 for s in stats), stats                  # ... and not one host all-reduce inside the time steps (round 4)
    # the ranks run the one-rank tracer layout: epoch-ordered constants, lazy columns (migration before the ONE sort of a step)
    lay = vc.all(lambda s: s.layout())
    assert all(a > 0 and b == 1 for a, b in lay), lay
    for k, f in ref["fields"].items():
        tol = 1e-9 if k in ("rho", "etas", "etan") else 1e-6
        got = vc.field(k)
        m = np.isfinite(f)
        assert np.array_equal(np.isfinite(got), m), k
        assert relerr(got[m], f[m]) < tol, (k, relerr(got[m], f[m]))
    X, F, V = vc.tracers()
    assert np.array_equal(F[:, 12], ref["F"][:, 12])
    assert relerr(X, ref["X"]) < 1e-8 and relerr(F[:, 3], ref["F"][:, 3]) < 1e-7 and relerr(V, ref["V"]) < 1e-5
    vc.close()


def test_blocks_2x2_vs_reference_driver_trajectory():
    """Falling block 41 x 41 on 2 x 2 blocks against the trajectory of the reference's own driver."""
    from pylamp_amd import driver
    g = golden("traj_block41")
    gz, gx = g["gz"], g["gx"]
    nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
    opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False)
    vc = driver.VirtualCluster(nx, L, 2, 2, g["init_tr_x"], g["init_tr_f"], opt)
    for it in range(1, int(g["nsteps"]) + 1):
        reps = vc.step()
        p = "s%d_" % it
        assert all(r["stokes"]["converged"] == 1 for r in reps)
        assert relerr(vc.field("velz"), g[p + "velz"]) < 1e-6 and relerr(vc.field("velx"), g[p + "velx"]) < 1e-6
        assert relerr(vc.field("rho"), g[p + "rho"]) < 1e-7
        X, F, V = vc.tracers()
        assert X.shape == g[p + "tr_x"].shape and relerr(X, g[p + "tr_x"]) < 1e-7 and relerr(V, g[p + "tr_v"]) < 1e-5
    vc.close()


def test_blocks_injection_deletion_and_graded_grid(oracle):
    """2 x 2 blocks: census + refill (cell-mean fields, per-cell counts) and fence-off deletion agree with one rank;
    a graded grid (blocks of unequal physical size, per-axis cell search) agrees with the oracle."""
    from pylamp_amd import driver
    nx = [65, 81]; L = [660e3, 820e3]
    rng = np.random.default_rng(9)
    tr_x, tr_f = driver.falling_block_tracers(nx, L, 10, rng)
    tr_x[7:60, 0] = -2000.0; tr_x[100:140, 1] = -1500.0                    # beyond the low walls: deleted when the fence is off
    opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, tracdens=10, tracdens_min=7, inject_unique_ids=True,
                         tracs_fence_enabled=False)
    ref = _one_rank(nx, L, tr_x, tr_f, opt, 2)
    vc = driver.VirtualCluster(nx, L, 2, 2, tr_x, tr_f, opt)
    for it in range(2):
        reps = vc.step()
        assert sum(r["ninjected"] for r in reps) == ref["reps"][it]["ninjected"]
        assert it > 0 or ref["reps"][it]["ninjected"] > 1000
        assert sum(r["nremoved"] for r in reps) == ref["reps"][it]["nremoved"]
        assert sum(r["ntrac"] for r in reps) == ref["reps"][it]["ntrac"]
    assert ref["reps"][0]["nremoved"] == 93
    X, F, V = vc.tracers()
    # injected tracers get layout-independent positions (the generator is keyed by the global cell) but IDs in
    # rank-major order: compare as sets of (position, fields without ID)
    key = lambda X_, F_: np.lexsort((X_[:, 1], X_[:, 0]))
    o1, o2 = key(X, F), key(ref["X"], ref["F"])
    assert relerr(X[o1], ref["X"][o2]) < 1e-7
    assert np.allclose(F[o1][:, :12], ref["F"][o2][:, :12], rtol=1e-7, atol=0, equal_nan=True)
    assert relerr(vc.field("velz"), ref["fields"]["velz"]) < 1e-6
    vc.close()

    def graded(n, Lx):
        h = np.linspace(1.0, 2.5, n - 1)
        c = np.concatenate([[0.0], np.cumsum(h)]); c *= Lx / c[-1]; c[-1] = Lx
        return c
    grid = [graded(nx[0], L[0]), graded(nx[1], L[1])]
    tr_x, tr_f = driver.mantle_tracers(nx, L, 20, np.random.default_rng(8))
    vc = driver.VirtualCluster(nx, L, 2, 2, tr_x, tr_f, driver.Options(), grid=grid)
    st = dict(nx=nx, L=L, grid=grid, tr_x=tr_x.copy(), tr_f=tr_f.copy())
    cfg = oracle.StepConfig()
    for it in (1, 2):
        reps = vc.step()
        with oracle.rect_search():
            out = oracle.step(st, cfg, it)
        assert reps[0]["tstep"] == pytest.approx(out["tstep"], rel=1e-6)
        assert relerr(vc.field("velz"), out["velz"]) < 1e-6 and relerr(vc.field("temp"), out["temp"]) < 1e-6
        X, F, V = vc.tracers()
        assert relerr(X, st["tr_x"]) < 1e-7 and relerr(F[:, 3], st["tr_f"][:, 3]) < 1e-6
    vc.close()


def test_blocks_communication_budget(monkeypatch):
    """Three distributed multigrid levels on 2 x 4 blocks (replication threshold lowered so that a 257 x 513 grid has
    them): with deep halos / tile kernels a preconditioner application costs at most 8 neighbour exchanges (one per smoothing
    sequence of a level instead of one per sweep) and a BiCGStab iteration TWO all-reduces -- the pressure-anchor deflation's
    scalars ride in them (round 4) --; the exchange-per-sweep mode (PYLAMP_MG_DEEP=0, staged kernels on the distributed levels)
    gives the same iterates at several times the exchanges."""
    from pylamp_amd import driver
    nx = [257, 513]; L = [660e3, 1320e3]
    tr_x, tr_f = driver.mantle_tracers(nx, L, 12, np.random.default_rng(6), perturb=60.0)
    res = {}
    for deep in ("1", "0"):
        monkeypatch.setenv("PYLAMP_MG_REPL_NODES", "3000")
        monkeypatch.setenv("PYLAMP_MG_DEEP", deep)
        monkeypatch.setenv("PYLAMP_MG_FUSED_DIST", deep)
        vc = driver.VirtualCluster(nx, L, 2, 4, tr_x, tr_f, driver.Options(do_heatdiff=False, tdep_rho=True, tdep_eta=True))
        vc.comm_stats(reset=True)
        rep = vc.step()[0]
        st = np.array(vc.comm_stats()).max(axis=0)
        res[deep] = dict(velz=vc.field("velz"), its=rep["stokes"]["iterations"], nprec=rep["stokes"]["precond_applies"],
                         napply=rep["stokes"]["operator_applies"], exchanges=int(st[0]), allreduces=int(st[2] + st[3]))
        assert rep["stokes"]["converged"] == 1
        vc.close()
    d, l = res["1"], res["0"]
    other = 80                                   # set-up (coefficient halos, power iterations), scatter, advection
    assert d["exchanges"] <= 8 * d["nprec"] + d["napply"] + other, d
    # two per BiCGStab iteration (= per two preconditioner applications; the short solve for the deflation vector included) plus
    # the deflation scalar of every application
    assert d["allreduces"] <= d["nprec"] + other, d            # (2 per iteration = 1 per preconditioner application)
    assert l["exchanges"] > 2.0 * d["exchanges"], (d, l)
    assert relerr(d["velz"], l["velz"]) < 1e-7
