"""Worker of tests/test_hip_multirank.py: N ranks (gloo) sharing ONE GPU, SPMD module API.
Every rank passes the full arrays like the reference's MPI ranks do; the grid is decomposed
into row slabs inside the library.  Prints PASS lines that the launching test asserts on."""
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
os.environ["PYLAMP_DEVICE"] = "0"

dist.init_process_group(backend="gloo")
rank, size = dist.get_rank(), dist.get_world_size()

from pylamp_amd import pylamp_stokes as S, pylamp_diff as D   # noqa: E402
from oracle import pylamp_oracle as O                          # noqa: E402


def relerr(a, b):
    return np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b))


rng = np.random.default_rng(11)
nx = [129, 97]; L = [660e3, 495e3]
grid = [np.linspace(0, L[0], nx[0]), np.linspace(0, L[1], nx[1])]
etas = 1e19 * 10 ** rng.uniform(0, 3, nx); etan = 1e19 * 10 ** rng.uniform(0, 3, nx)
rho = 3300 + rng.uniform(-50, 50, nx)
for bc in ([1, 1, 1, 1], [0, 1, 0, 1]):
    A, rhs = S.makeStokesMatrix(nx, grid, etas, etan, rho, bc)
    assert A._ctx.nranks == size
    x = rng.standard_normal(A.shape[0])
    e = relerr(A @ x, O.stokes_apply(nx, grid, etas, etan, bc, x))
    assert e < 1e-13, e
    assert np.allclose(rhs, O.stokes_rhs(nx, rho), rtol=1e-14, atol=0)
if rank == 0:
    print("PASS apply", flush=True)

# smooth viscosity + blocky density: solve vs the oracle's direct solve
Z, X = np.meshgrid(*grid, indexing='ij')
T = 273 + 1350 * Z / L[0] + 30 * np.sin(3 * np.pi * X / L[1]) * np.sin(np.pi * Z / L[0])
eta = lambda t: np.clip(1e20 * np.exp(120e3 / (O.GASR * t) - 120e3 / (O.GASR * 1623)), 1e17, 1e23)
gm = O.gridmp_of(grid)
Zc, Xc = np.meshgrid(*gm, indexing='ij')
Tc = np.clip(273 + 1350 * Zc / L[0] + 30 * np.sin(3 * np.pi * Xc / L[1]) * np.sin(np.pi * Zc / L[0]), 273, 1700)
es, en, rh = eta(T), eta(Tc), 3300 / (3.5e-5 * (T - 1623) + 1)
bc = [1, 1, 1, 1]
A, rhs = S.makeStokesMatrix(nx, grid, es, en, rh, bc)
xs = S.solve(A, rhs)
st = A.last_stats
xref = O.stokes_solve(nx, grid, es, en, rh, bc)
(vz, vx), p = S.x2vp(xs, nx); (rz, rx), rp = O.x2vp(xref, nx)
ev = np.sqrt((np.sum((vz - rz) ** 2) + np.sum((vx - rx) ** 2)) / (np.sum(rz ** 2) + np.sum(rx ** 2)))
assert st["converged"] == 1 and ev < 1e-6, (ev, st)
if rank == 0:
    print("PASS solve its=%d vel_err=%.2e" % (st["iterations"], ev), flush=True)

# heat
kz = rng.uniform(2, 5, nx); kx = rng.uniform(2, 5, nx); Cp = rng.uniform(1000, 1250, nx)
H = rng.uniform(0, 1e-9, nx) * 3300; T0 = rng.uniform(273, 1623, nx)
ts = 0.67 * (L[0] / (nx[0] - 1)) ** 2 / np.max(2 * kz / (rh * Cp))
A, rhs = D.makeDiffusionMatrix(nx, grid, gm, T0, [kz, kx], Cp, rh, H, [0, 1, 0, 1], [273.0, 0.0, 1623.0, 0.0], ts)
x = rng.standard_normal(A.shape[0])
assert relerr(A @ x, O.heat_apply(nx, grid, gm, [kz, kx], Cp, rh, [0, 1, 0, 1], ts, x)) < 1e-13
sol = D.solve(A, rhs)
ref = O.heat_solve(nx, grid, gm, T0, [kz, kx], Cp, rh, H, [0, 1, 0, 1], [273.0, 0.0, 1623.0, 0.0], ts)
assert relerr(sol, ref) < 1e-6
if rank == 0:
    print("PASS heat", flush=True)
dist.barrier()
dist.destroy_process_group()
