import os, sys, numpy as np
sys.path.insert(0, '.')
from pylamp_amd import pylamp_stokes as S, _context
rng = np.random.default_rng(41)
nx = [1025, 1025]; L = [660e3, 660e3]
grid = [np.linspace(0, L[d], nx[d]) for d in range(2)]
Z, X = np.meshgrid(*grid, indexing='ij')
gm = [np.append(0.5 * (g[1:] + g[:-1]), g[-1] + 0.5 * (g[-1] - g[-2])) for g in grid]
Zc, Xc = np.meshgrid(*gm, indexing='ij')
f = lambda z, x: 1e20 * 10 ** (2.0 * np.sin(2 * np.pi * x / L[1]) * np.cos(np.pi * z / L[0]) + 0.3 * np.sin(17 * x / L[1]) * np.sin(23 * z / L[0]))
rho = 3300 + 40 * np.sin(2 * np.pi * X / L[1]) * np.sin(np.pi * Z / L[0])
A, rhs = S.makeStokesMatrix(nx, grid, f(Z, X), f(Zc, Xc), rho, [1, 1, 1, 1])
for rtol in (1e-7, 1e-10):
    x = S.solve(A, rhs, rtol=rtol) if 'rtol' in S.solve.__code__.co_varnames else S.solve(A, rhs)
    print(os.environ.get('PYLAMP_L0_MIXED'), rtol, A.last_stats['iterations'], A.last_stats['rel_residual'], A.last_stats['error_estimate'])
