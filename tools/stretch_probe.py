"""Stretched / graded grids: the multigrid-preconditioned Stokes solve against the oracle's direct solve.
    python tools/stretch_probe.py [case ...]      cases: flat (513 x 129 nodes on a 1 x 1 domain), flat16 (1025 x 65), tall16 (65 x 1025), tall (129 x 513), graded (513^2, z spacing
                                                  growing 30x towards the bottom), graded2 (both axes graded)"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import pylamp_stokes as S
from oracle import pylamp_oracle as O


def graded(n, ratio):
    w = np.geomspace(1.0, ratio, n - 1)
    return np.concatenate([[0.0], np.cumsum(w)]) / w.sum()


def case(name):
    if name == "flat": nx = [513, 129]; grid = [np.linspace(0, 1, 513), np.linspace(0, 1, 129)]
    elif name == "flat16": nx = [1025, 65]; grid = [np.linspace(0, 1, 1025), np.linspace(0, 1, 65)]
    elif name == "iso": nx = [257, 257]; grid = [np.linspace(0, 1, 257), np.linspace(0, 1, 257)]
    elif name == "iso_isovisc": nx = [257, 257]; grid = [np.linspace(0, 1, 257), np.linspace(0, 1, 257)]
    elif name == "tall_isovisc": nx = [129, 513]; grid = [np.linspace(0, 1, 129), np.linspace(0, 1, 513)]
    elif name == "tall16_isovisc": nx = [65, 1025]; grid = [np.linspace(0, 1, 65), np.linspace(0, 1, 1025)]
    elif name == "flat_isovisc": nx = [513, 129]; grid = [np.linspace(0, 1, 513), np.linspace(0, 1, 129)]
    elif name == "flat16_isovisc": nx = [1025, 65]; grid = [np.linspace(0, 1, 1025), np.linspace(0, 1, 65)]
    elif name == "tall16": nx = [65, 1025]; grid = [np.linspace(0, 1, 65), np.linspace(0, 1, 1025)]
    elif name == "tall": nx = [129, 513]; grid = [np.linspace(0, 1, 129), np.linspace(0, 1, 513)]
    elif name == "graded": nx = [513, 513]; grid = [graded(513, 30.0), np.linspace(0, 1, 513)]
    elif name == "graded2": nx = [513, 513]; grid = [graded(513, 30.0), graded(513, 10.0)]
    elif name == "graded1025": nx = [1025, 1025]; grid = [graded(1025, 30.0), np.linspace(0, 1, 1025)]
    else: raise SystemExit("unknown case " + name)
    Z, X = np.meshgrid(grid[0], grid[1], indexing="ij")
    eta = 10 ** (1.5 * np.sin(3 * np.pi * X) * np.cos(2 * np.pi * Z))
    if name.endswith("_isovisc"): eta = np.ones_like(eta)
    etan = eta
    rho = 1.0 + 0.1 * np.exp(-((Z - 0.4) ** 2 + (X - 0.55) ** 2) / 0.02)
    return nx, grid, eta, etan, rho


for name in (sys.argv[1:] or ["flat", "tall", "graded", "graded2"]):
    nx, grid, es, en, rho = case(name)
    bc = [1, 1, 1, 1]
    A, rhs = S.makeStokesMatrix(nx, grid, es, en, rho, bc)
    t = time.time(); x = S.solve(A, rhs); tg = time.time() - t
    st = A.last_stats
    line = "%-10s %4dx%-4d iterations %3d converged %d direct %d rel_residual %.2e estimate %.2e solve_ms %.1f" % (
        name, nx[0], nx[1], st["iterations"], st["converged"], st["used_direct"], st["rel_residual"], st["error_estimate"], st["solve_ms"])
    if nx[0] * nx[1] <= 300000 or os.environ.get("STRETCH_ORACLE"):
        t = time.time(); xo = O.stokes_solve(nx, grid, es, en, rho, bc); to = time.time() - t
        (vz, vx), _ = S.x2vp(x, nx); (rz, rx), _ = S.x2vp(xo, nx)
        ev = np.sqrt((np.sum((vz - rz) ** 2) + np.sum((vx - rx) ** 2)) / (np.sum(rz ** 2) + np.sum(rx ** 2)))
        line += "  velocity error vs direct solve %.2e (oracle %.1f s)" % (ev, to)
    print(line, flush=True)
