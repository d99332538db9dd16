"""A few applications of the Stokes preconditioner on an n x n mantle-like problem (for rocprofv3 --pmc / --kernel-trace):
    python tools/mg_probe.py [n=2049] [applications=3]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pylamp_amd import pylamp_stokes as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
nx = [n, n]; L = [660e3, 660e3]
grid = [np.linspace(0, L[d], nx[d]) for d in range(2)]
Z, X = np.meshgrid(*grid, indexing='ij')
gm = [np.append(0.5 * (g[1:] + g[:-1]), g[-1] + 0.5 * (g[-1] - g[-2])) for g in grid]
Zc, Xc = np.meshgrid(*gm, indexing='ij')
f = lambda z, x: 1e20 * 10 ** (1.5 * np.sin(2 * np.pi * x / L[1]) * np.cos(np.pi * z / L[0]))
rho = 3300 + 40 * np.sin(2 * np.pi * X / L[1]) * np.sin(np.pi * Z / L[0])
A, rhs = S.makeStokesMatrix(nx, grid, f(Z, X), f(Zc, Xc), rho, [1, 1, 1, 1])
r = np.random.default_rng(1).standard_normal(3 * n * n)
for k in range(reps):
    z = A.precond(r)
print("ok", float(np.abs(z).max()))
