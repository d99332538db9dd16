"""FP32 multigrid levels against FP64 ones: preconditioner output, iteration counts, solution (run on the GPU box)."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import pylamp_stokes as S, driver

def model(n, seed=5):
    nx = [n, n]; L = [660e3, 660e3]
    grid = [np.linspace(0, L[0], n), np.linspace(0, L[1], n)]
    Z, X = np.meshgrid(*grid, indexing='ij')
    gm = [0.5 * (g[1:] + g[:-1]) for g in grid]; gm = [np.append(g, g[-1] + (g[-1] - g[-2])) for g in gm]
    Zc, Xc = np.meshgrid(*gm, indexing='ij')
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * np.pi, 6)
    f = lambda z, x: 1e20 * 10 ** (1.5 + 1.5 * np.sin(3 * np.pi * x / L[1] + ph[0]) * np.cos(2 * np.pi * z / L[0] + ph[1]))
    rho = 3300 + 40 * np.sin(2 * np.pi * X / L[1] + ph[2]) * np.sin(np.pi * Z / L[0]) + 20 * rng.standard_normal(Z.shape)
    return nx, grid, f(Z, X), f(Zc, Xc), rho

for n in [int(a) for a in sys.argv[1:]] or [513, 1025]:
    nx, grid, es, en, rho = model(n)
    A, rhs = S.makeStokesMatrix(nx, grid, es, en, rho, [1, 1, 1, 1])
    out = {}
    for fp32 in (False, True):
        A.set_mg_precision(fp32, 1000)
        t = time.time(); x = S.solve(A, rhs); dt = time.time() - t
        st = dict(A.last_stats)
        r = np.random.default_rng(0).standard_normal(A.shape[0]) * np.abs(rhs).max()
        z = A.precond(r)
        z2 = A.precond(r)
        print("   precond repeatability %.2e" % (np.linalg.norm(z2 - z) / np.linalg.norm(z)), A.mg_info()[1][:3], flush=True)
        out[fp32] = (x, z, st)
        print(n, "fp32" if fp32 else "fp64", "levels", A.mg_precision(), "its", st["iterations"], "conv", st["converged"],
              "res %.2e" % st["rel_residual"], "dev ms %.1f" % st.get("device_ms", -1), "wall %.2f" % dt, flush=True)
    x0, z0, _ = out[False]; x1, z1, _ = out[True]
    v = lambda x: x.reshape(n, n, 3)[:, :, :2]
    print("   precond diff (vel) %.2e   solution diff (vel) %.2e" % (
        np.linalg.norm(v(z1) - v(z0)) / np.linalg.norm(v(z0)), np.linalg.norm(v(x1) - v(x0)) / np.linalg.norm(v(x0))), flush=True)
