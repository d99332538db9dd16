"""The solver's velocity-error estimate against the true error (oracle direct solve) on the cross-check problem of
tests/test_hip_solve.py (129^2, smooth random viscosity over 2 decades, density noise of +-50 per node)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import pylamp_stokes as S
from oracle import pylamp_oracle as O
rng = np.random.default_rng(9)
n = 129; nx = [n, n]; grid = [np.linspace(0, 660e3, n), np.linspace(0, 660e3, n)]
def fld():
    a = rng.uniform(0, 2, nx)
    for _ in range(8):
        p = np.pad(a, 1, mode="edge")
        a = (p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] + 4 * a) / 8
    return 1e19 * 10 ** ((a - a.min()) / (a.max() - a.min()) * 2)
etas = fld(); etan = fld(); rho = 3300 + rng.uniform(-50, 50, nx)
A, rhs = S.makeStokesMatrix(nx, grid, etas, etan, rho, [1, 1, 1, 1])
xr = O.stokes_solve(nx, grid, etas, etan, rho, [1, 1, 1, 1])
(rz, rx), _ = O.x2vp(xr, nx)
for rtol in (1e-5, 1e-7, 1e-9, 1e-11):
    x = S.solve(A, rhs, rtol=rtol)
    (vz, vx), _ = S.x2vp(x, nx)
    err = float(np.sqrt((np.sum((vz - rz) ** 2) + np.sum((vx - rx) ** 2)) / (np.sum(rz ** 2) + np.sum(rx ** 2))))
    st = A.last_stats
    print("rtol %.0e its %3d conv %d res %.2e est %.2e true err %.2e" % (rtol, st["iterations"], st["converged"], st["rel_residual"], st["error_estimate"], err), flush=True)
