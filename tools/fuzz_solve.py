"""Randomised campaign of module-level Stokes solves: random small shapes (5x5 up), uniform and stretched grids,
NOSLIP / FREESLIP z-walls, smooth random viscosity over 0-4 decades -- velocity against the oracle's direct solve
(1e-6) and the converged flag.  Usage: python tools/fuzz_solve.py ncases [seed]
Measured: 378 of 380 cases pass; the two misses are grids only 5 nodes wide (velocity error 1.3e-6 and 2.3e-6)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import pylamp_stokes as S
from oracle import pylamp_oracle as O
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(int(sys.argv[1])):
    nz, nxx = int(rng.integers(5, 80)), int(rng.integers(5, 80))
    nx = [nz, nxx]; h = float(rng.uniform(0.5, 2))
    stretched = bool(rng.integers(0, 2))
    def axis(n):
        if not stretched: return np.linspace(0, h * (n - 1), n)
        d = 1.0 + float(rng.uniform(0.1, 1.0)) * np.sin(np.linspace(0, 3, n - 1) + float(rng.uniform(0, 6))) ** 2
        c = np.concatenate([[0.0], np.cumsum(d)]); return c * (h * (n - 1) / c[-1])
    grid = [axis(nz), axis(nxx)]
    dec = float(rng.uniform(0, 4))
    def fld():
        a = rng.uniform(0, 1, nx)
        for _ in range(6):
            p = np.pad(a, 1, mode='edge'); a = (p[:-2, 1:-1] + p[2:, 1:-1] + p[1:-1, :-2] + p[1:-1, 2:] + 4 * a) / 8
        return 10 ** ((a - a.min()) / max(a.max() - a.min(), 1e-30) * dec)
    es = fld(); en = np.sqrt(es * np.roll(es, -1, 0)); rho = 1 + 0.1 * fld() / 10 ** dec
    bc = [int(rng.integers(0, 2)), 1, int(rng.integers(0, 2)), 1]
    try:
        A, rhs = S.makeStokesMatrix(nx, grid, es, en, rho, bc)
        x = S.solve(A, rhs)
        xr = O.stokes_solve(nx, grid, es, en, rho, bc)
        (vz, vx), p = S.x2vp(x, nx); (rz, rx), rp = O.x2vp(xr, nx)
        ev = float(np.sqrt((np.sum((vz - rz) ** 2) + np.sum((vx - rx) ** 2)) / (np.sum(rz ** 2) + np.sum(rx ** 2))))
        st = A.last_stats
        ok = ev < 1e-6 and st["converged"] == 1
        print(("ok   " if ok else "FAIL ") + "case %d %dx%d str=%d dec=%.1f bc=%s  its %d conv %d vel err %.1e" % (case, nz, nxx, stretched, dec, bc, st["iterations"], st["converged"], ev), flush=True)
        bad += 0 if ok else 1
    except Exception as ex:
        bad += 1; print("EXC  case %d %dx%d  %r" % (case, nz, nxx, ex), flush=True)
print("failures:", bad)
