"""NumPy prototype: rank-1 deflation of the pressure-anchor mode in the preconditioner (see pl_solver.hip, "deflation of the
pressure-anchor mode"): BiCGStab histories with and without, w exact and w solved to 1e-3.  python tools/defl.py [n]"""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from pylamp_amd import driver
from oracle import pylamp_oracle as O, proto_stokes_solver as PS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 129
nx = [n, n]; L = [660e3, 660e3]
grid = [np.linspace(0, L[0], n), np.linspace(0, L[1], n)]
rng = np.random.default_rng(1)
tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng); O.property_update(tr_f, True, True)
frho, fes = O.trac2grid(tr_x, tr_f[:, [0, 1]], grid, nx, [5, 6])
fen, = O.trac2grid(tr_x, tr_f[:, [1]], O.gridmp_of(grid), nx, [6])
bc = [1, 1, 1, 1]
def hist(A, b, apply, tol=1e-10):
    n_ = b.size; x = np.zeros(n_); r = b.copy()
    rt = np.random.default_rng(1234).standard_normal(n_)
    rho = alpha = omega = 1.0; v = np.zeros(n_); p = np.zeros(n_); bn = np.linalg.norm(b); h = []
    for it in range(120):
        rho_new = rt @ r; beta = (rho_new / rho) * (alpha / omega)
        p = r + beta * (p - omega * v); y = apply(p); v = A @ y
        alpha = rho_new / (rt @ v); s = r - alpha * v; z = apply(s); t = A @ z
        omega = (t @ s) / (t @ t); x = x + alpha * y + omega * z; r = s - omega * t; rho = rho_new
        h.append(np.linalg.norm(r) / bn)
        if h[-1] < tol: break
    return x, h
M = PS.Precond(nx, grid, fes, fen, frho, bc, nu=(2, 2)); Sc = PS.Scaled(M, nx, grid)
A, b = Sc.A.tocsr(), Sc.b
x0, h0 = hist(A, b, Sc.apply)
print("no deflation       its %2d : %s" % (len(h0), " ".join("%.0e" % v for v in h0)))
cls = O.stokes_row_class(nx)
en = np.array(fen, copy=True); en[~np.isfinite(en)] = 1.0
hz = L[0] / (n - 1); hx = L[1] / (n - 1)
cont = cls[2] == 1
# u: slow right eigenvector of A M^-1 in (scaled) residual space: continuity residual that M^-1 turns into a constant pressure
U = np.zeros((nx[0], nx[1], 3)); U[:, :, 2][cont] = 1.0 / (en[cont] * (1 / hz + 1 / hx)); u = U.reshape(-1); u /= np.linalg.norm(u)
# y: left eigenvector: area-weighted sum of the continuity rows (Gauss)
Y = np.zeros((nx[0], nx[1], 3)); Y[:, :, 2][cont] = hz + hx; y = Y.reshape(-1)
lu = spl.splu(A.tocsc())
for name, w in (("exact w = A^-1 u", lu.solve(u)), ("w to 1e-3", None)):
    if w is None:
        w, hw = hist(A, u, Sc.apply, tol=1e-3); print("   (w solve: %d its)" % len(hw))
    Aw = A @ w
    yAw = y @ Aw
    print("   cos(A w, u) = %.4f   y.Aw = %.3e" % ((Aw @ u) / np.linalg.norm(Aw), yAw))
    def apply(r):
        z = Sc.apply(r)
        return z + w * ((y @ (r - A @ z)) / yAw)
    x1, h1 = hist(A, b, apply)
    print("deflated (%s) its %2d : %s" % (name, len(h1), " ".join("%.0e" % v for v in h1)))
    print("   vel diff to undeflated solve %.2e" % (np.linalg.norm((x1 - x0).reshape(nx[0], nx[1], 3)[:, :, :2]) / np.linalg.norm(x0.reshape(nx[0], nx[1], 3)[:, :, :2])))
