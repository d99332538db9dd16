"""NumPy prototype: dense spectrum of A M^-1 (row-scaled Stokes operator times the block-triangular / V-cycle preconditioner) on a
33 x 33 grid -- one eigenvalue near 1/(number of cells) (the pressure-anchor mode), a cluster at 0.16-0.3 (boundary cells), the
rest near 1.  python tools/spectrum.py [iso|layered]"""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from oracle import pylamp_oracle as O, proto_stokes_solver as PS
n = 33
nx = [n, n]; L = [660e3, 660e3]
grid = [np.linspace(0, L[0], n), np.linspace(0, L[1], n)]
Z, X = np.meshgrid(*grid, indexing='ij'); Zc, Xc = np.meshgrid(*O.gridmp_of(grid), indexing='ij')
which = sys.argv[1] if len(sys.argv) > 1 else "iso"
if which == "iso":
    fes = np.full(nx, 1e21); fen = np.full(nx, 1e21)
else:
    fes = 10 ** (23 - 3 * Z / L[0]); fen = 10 ** (23 - 3 * np.clip(Zc, 0, L[0]) / L[0])
frho = 3300 + 30 * np.sin(2 * np.pi * X / L[1]) * np.sin(np.pi * Z / L[0])
bc = [1, 1, 1, 1]
M = PS.Precond(nx, grid, fes, fen, frho, bc, nu=(2, 2)); Sc = PS.Scaled(M, nx, grid)
A = Sc.A.toarray(); N = A.shape[0]
t = time.time()
Minv = np.zeros((N, N))
for k in range(N):
    e = np.zeros(N); e[k] = 1.0
    Minv[:, k] = Sc.apply(e)
print("built M^-1 in %.1f s" % (time.time() - t))
T = A @ Minv
w, V = np.linalg.eig(T)
idx = np.argsort(np.abs(w))
print("smallest |lambda|:", " ".join("%.3g" % abs(w[i]) for i in idx[:40]))
print("largest  |lambda|:", " ".join("%.3g" % abs(w[i]) for i in idx[-10:]))
print("count |lambda-1| < 0.3: %d of %d;  |lambda| < 0.3: %d" % (np.sum(np.abs(w - 1) < 0.3), N, np.sum(np.abs(w) < 0.3)))
cls = O.stokes_row_class(nx)
for i in idx[:14]:
    v = np.real(V[:, i]).reshape(n, n, 3)
    e = [np.linalg.norm(v[:, :, q]) for q in range(3)]
    P = np.abs(v[:, :, 2]); tot = np.sum(P ** 2) + 1e-300
    bnd = (np.sum(P[0] ** 2) + np.sum(P[n - 2] ** 2) + np.sum(P[:, 0] ** 2) + np.sum(P[:, n - 2] ** 2)) / tot
    k = np.unravel_index(np.argmax(P), P.shape)
    print("lambda %.3g%+.3gj  |vz| %.2f |vx| %.2f |p| %.2f   p: boundary-cell share %.2f, max at %s, ghost share %.2f" % (
        w[i].real, w[i].imag, e[0], e[1], e[2], bnd, k, (np.sum(P[n - 1] ** 2) + np.sum(P[:, n - 1] ** 2)) / tot))
