"""Worker: the resident time step on N ranks (gloo, one shared GPU) against the reference-driver
trajectory fixture and against the oracle's step()."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
os.environ["PYLAMP_DEVICE"] = "0"
dist.init_process_group(backend="gloo")
rank, size = dist.get_rank(), dist.get_world_size()

from pylamp_amd import driver                      # noqa: E402
from oracle import pylamp_oracle as O              # noqa: E402


def relerr(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b))


# ---- 1. falling block 41x41 vs the trajectory of the reference driver (golden fixture)
g = np.load(os.path.join(ROOT, "tests", "golden", "traj_block41.npz"))
gz, gx = g["gz"], g["gx"]
nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
sim = driver.Simulation(nx, L, g["init_tr_x"], g["init_tr_f"], driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False))
assert sim.ctx.nranks == size
ntot = g["init_tr_x"].shape[0]
for it in range(1, int(g["nsteps"]) + 1):
    rep = sim.step()
    p = "s%d_" % it
    assert rep["stokes"]["converged"] == 1, rep
    assert relerr(sim.field("velz"), g[p + "velz"]) < 1e-6 and relerr(sim.field("velx"), g[p + "velx"]) < 1e-6
    assert relerr(sim.field("rho"), g[p + "rho"]) < 1e-7
    X, F, V = sim.gather_tracers()
    assert X.shape[0] == ntot, (X.shape, ntot)               # nobody lost or duplicated in migration
    assert relerr(X, g[p + "tr_x"]) < 1e-7 and relerr(V, g[p + "tr_v"]) < 1e-5
sim.close()
if rank == 0:
    print("PASS block trajectory", flush=True)

# ---- 2. mantle model with heat, 129x97, 3 steps vs oracle.step
nx = [129, 97]; L = [660e3, 495e3]
rng = np.random.default_rng(3)
tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng)
sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options())
st = dict(nx=nx, L=L, grid=[np.linspace(0, L[0], nx[0]), np.linspace(0, L[1], nx[1])], tr_x=tr_x.copy(), tr_f=tr_f.copy())
cfg = O.StepConfig()
for it in (1, 2, 3):
    rep = sim.step()
    out = O.step(st, cfg, it)
    assert rep["limiter"] == out["limiter"] and abs(rep["tstep"] / out["tstep"] - 1) < 1e-6
    assert relerr(sim.field("velz"), out["velz"]) < 1e-6 and relerr(sim.field("velx"), out["velx"]) < 1e-6
    assert relerr(sim.field("temp"), out["temp"]) < 1e-6
    X, F, V = sim.gather_tracers()
    assert X.shape[0] == tr_x.shape[0]
    assert relerr(X, st["tr_x"]) < 1e-7 and relerr(F[:, 3], st["tr_f"][:, 3]) < 1e-6
# snapshot files are written once (rank 0) and hold ALL tracers
import tempfile                                      # noqa: E402
snapdir = os.path.join(tempfile.gettempdir(), "pylamp_snap_%d" % size)
sim.write_snapshot(snapdir)
dist.barrier()
if rank == 0:
    tc = np.load(os.path.join(snapdir, "tracs.%06d.npz" % sim.it))
    assert tc["tr_x"].shape[0] == tr_x.shape[0] and tc["tr_v"].shape == tc["tr_x"].shape
    gd = np.load(os.path.join(snapdir, "griddata.%06d.npz" % sim.it))
    assert gd["velz"].shape == tuple(nx)
sim.close()
if rank == 0:
    print("PASS mantle steps", flush=True)

# ---- 3. graded (non-uniform) grid: per-axis cell search in every marker kernel, slabs of unequal thickness
nx = [97, 81]; L = [660e3, 550e3]


def graded(n, Lx):
    h = np.linspace(1.0, 2.5, n - 1)
    c = np.concatenate([[0.0], np.cumsum(h)]); c *= Lx / c[-1]; c[-1] = Lx
    return c


grid = [graded(nx[0], L[0]), graded(nx[1], L[1])]
rng = np.random.default_rng(8)
tr_x, tr_f = driver.mantle_tracers(nx, L, 20, rng)
sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options(), grid=grid)
st = dict(nx=nx, L=L, grid=grid, tr_x=tr_x.copy(), tr_f=tr_f.copy())
for it in (1, 2):
    rep = sim.step()
    with O.rect_search():
        out = O.step(st, cfg, it)
    assert rep["stokes"]["converged"] == 1 and abs(rep["tstep"] / out["tstep"] - 1) < 1e-6, rep
    assert relerr(sim.field("velz"), out["velz"]) < 1e-6 and relerr(sim.field("velx"), out["velx"]) < 1e-6
    assert relerr(sim.field("temp"), out["temp"]) < 1e-6
    X, F, V = sim.gather_tracers()
    assert X.shape[0] == tr_x.shape[0]
    assert relerr(X, st["tr_x"]) < 1e-7 and relerr(F[:, 3], st["tr_f"][:, 3]) < 1e-6
sim.close()
if rank == 0:
    print("PASS graded grid steps", flush=True)

# ---- 4. free-surface stabilisation loop (damping sign): stabilisation-aware multigrid levels across ranks
g = np.load(os.path.join(ROOT, "tests", "golden", "traj_surfstab41.npz"))
gz, gx = g["gz"], g["gx"]
nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, surface_stabilization=True, surfstab_strict_reference=False)
sim = driver.Simulation(nx, L, g["init_tr_x"], g["init_tr_f"], opt)
st = dict(nx=nx, L=L, grid=[gz, gx], tr_x=g["init_tr_x"].copy(), tr_f=g["init_tr_f"].copy())
cfg2 = O.StepConfig(do_heatdiff=False, tdep_rho=False, tdep_eta=False, surface_stabilization=True, surfstab_theta=-0.5)
for it in (1, 2):
    rep = sim.step()
    out = O.step(st, cfg2, it)
    assert rep["stokes"]["converged"] == 1 and rep["stokes_resolves"] == out["nresolve"], rep
    assert abs(rep["tstep"] / out["tstep"] - 1) < 1e-6
    assert relerr(sim.field("velz"), out["velz"]) < 1e-6 and relerr(sim.field("velx"), out["velx"]) < 1e-6
sim.close()
if rank == 0:
    print("PASS stabilisation loop", flush=True)
dist.barrier()
dist.destroy_process_group()
