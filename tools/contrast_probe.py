"""Where the multigrid-preconditioned iteration stops being trustworthy as the viscosity contrast grows:
the reference's stock model 5 (sphere, 201 x 41 nodes; fields of tests/golden/stokes_solve_sphere201x41.npz) with its contrast of
1e10 rescaled to 1e3 ... 1e10, solved (a) by the iteration alone (PYLAMP_NO_DIRECT=1, gate off) and (b) as shipped (contrast gate),
against oracle.stokes_solve_refined (equilibrated LU + extended-precision refinement).
    python tools/contrast_probe.py [k ...]      contrasts 10^k"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from oracle import pylamp_oracle as O

g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'stokes_solve_sphere201x41.npz'))
nx = [int(v) for v in g["nx"]]; grid = [g["gz"], g["gx"]]; bc = list(g["bc"])


def scaled(eta, k):
    lo = np.log10(np.nanmin(eta)); span = np.log10(np.nanmax(eta)) - lo
    return 10 ** (lo + (np.log10(eta) - lo) * (k / span))


def err(S, x, xr):
    (vz, vx), _ = S.x2vp(x, nx); (rz, rx), _ = S.x2vp(xr, nx)
    return np.sqrt((np.sum((vz - rz) ** 2) + np.sum((vx - rx) ** 2)) / (np.sum(rz ** 2) + np.sum(rx ** 2)))


for k in [float(a) for a in sys.argv[1:]] or [3, 4, 5, 6, 7, 8, 10]:
    es, en = scaled(g["etas"], k), scaled(g["etan"], k)
    xr = O.stokes_solve_refined(nx, grid, es, en, g["rho"], bc, refinements=4)
    out = []
    for mode, env in (("iteration only", {"PYLAMP_NO_DIRECT": "1", "PYLAMP_CONTRAST_GATE": "1e30"}), ("as shipped", {})):
        for kk, vv in env.items(): os.environ[kk] = vv
        try:
            from pylamp_amd import pylamp_stokes as S
            A, rhs = S.makeStokesMatrix(nx, grid, es, en, g["rho"], bc)
            x = S.solve(A, rhs); st = A.last_stats
            out.append("%s: error %.1e estimate %.1e converged %d direct %d iterations %d" % (mode, err(S, x, xr), st["error_estimate"], st["converged"], st["used_direct"], st["iterations"]))
        finally:
            for kk in env: os.environ.pop(kk, None)
    print("contrast 1e%g | %s | %s" % (k, out[0], out[1]), flush=True)
