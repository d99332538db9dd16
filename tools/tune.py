"""Tuning experiment (not part of the product): solver settings vs accuracy / time."""
import sys, os, time
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import pylamp_stokes as S, driver
from oracle import pylamp_oracle as O

def fields(n, model):
    nx = [n, n]; L = [660e3, 660e3]
    grid = [np.linspace(0, L[0], n), np.linspace(0, L[1], n)]
    rng = np.random.default_rng(5)
    if model == "block":
        tr_x, tr_f = driver.falling_block_tracers(nx, L, 16, rng)
        O.property_update(tr_f, False, False)
    else:
        tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng)
        O.property_update(tr_f, True, True)
    frho, fes = O.trac2grid(tr_x, tr_f[:, [0, 1]], grid, nx, [5, 6])
    fen, = O.trac2grid(tr_x, tr_f[:, [1]], O.gridmp_of(grid), nx, [6])
    return nx, grid, fes, fen, frho

mode = sys.argv[1]
if mode == "rtol":
    for model in ("block", "mantle"):
        nx, grid, es, en, rho = fields(257, model)
        bc = [1, 1, 1, 1]
        xref = O.stokes_solve(nx, grid, es, en, rho, bc)
        (rz, rx), rp = O.x2vp(xref, nx)
        A, rhs = S.makeStokesMatrix(nx, grid, es, en, rho, bc)
        for rtol in (1e-6, 1e-7, 1e-8, 1e-9, 1e-10, 1e-11):
            x = S.solve(A, rhs, rtol=rtol)
            (vz, vx), p = S.x2vp(x, nx)
            ev = np.sqrt((np.sum((vz - rz) ** 2) + np.sum((vx - rx) ** 2)) / (np.sum(rz ** 2) + np.sum(rx ** 2)))
            st = A.last_stats
            print(model, "rtol %.0e its %d relres %.2e vel err %.2e p err %.2e" % (rtol, st["iterations"], st["rel_residual"], ev, np.linalg.norm(p - rp) / np.linalg.norm(rp)), flush=True)
else:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2049
    nx = [n, n]; L = [660e3, 660e3]
    rng = np.random.default_rng(1)
    tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng)
    sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options())
    for k in range(3):
        r = sim.step()
        print(os.environ.get("PYLAMP_MG_NU"), os.environ.get("PYLAMP_MG_COARSE"), "step", k, "stokes ms %.1f its %d res %.2e" % (r["ms_stokes"], r["stokes"]["iterations"], r["stokes"]["rel_residual"]), flush=True)
