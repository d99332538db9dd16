"""NumPy prototype: the plateau of BiCGStab and GMRES on the mantle model and where its residual lives.  python tools/plateau.py [n]"""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, scipy.sparse.linalg as spl
from pylamp_amd import driver
from oracle import pylamp_oracle as O, proto_stokes_solver as PS
n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
nx = [n, n]; L = [660e3, 660e3]
grid = [np.linspace(0, L[0], n), np.linspace(0, L[1], n)]
rng = np.random.default_rng(1)
tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng); O.property_update(tr_f, True, True)
frho, fes = O.trac2grid(tr_x, tr_f[:, [0, 1]], grid, nx, [5, 6])
fen, = O.trac2grid(tr_x, tr_f[:, [1]], O.gridmp_of(grid), nx, [6])
bc = [1, 1, 1, 1]
M = PS.Precond(nx, grid, fes, fen, frho, bc, nu=(2, 2)); Sc = PS.Scaled(M, nx, grid)
A, b = Sc.A, Sc.b
# hydrostatic start like the GPU: subtract the hydrostatic pressure solution?  (prototype: plain b)
def bicg_hist(A, b, M, maxit=60, seed=1234):
    n_ = b.size; x = np.zeros(n_); r = b.copy()
    rt = np.random.default_rng(seed).standard_normal(n_)
    rho = alpha = omega = 1.0; v = np.zeros(n_); p = np.zeros(n_); bn = np.linalg.norm(b); h = []
    for it in range(maxit):
        rho_new = rt @ r; beta = (rho_new / rho) * (alpha / omega)
        p = r + beta * (p - omega * v); y = M.apply(p); v = A @ y
        alpha = rho_new / (rt @ v); s = r - alpha * v; z = M.apply(s); t = A @ z
        omega = (t @ s) / (t @ t); x = x + alpha * y + omega * z; r = s - omega * t; rho = rho_new
        R = r.reshape(nx[0], nx[1], 3)
        h.append((np.linalg.norm(r) / bn, np.linalg.norm(R[:, :, 2]) / np.linalg.norm(r)))
        if h[-1][0] < 1e-11: break
    return x, h
x, h = bicg_hist(A, b, Sc)
print("BiCGStab |r|/|b| (continuity share):")
print(" ".join("%.0e(%.2f)" % v for v in h))
# right-preconditioned GMRES (full, no restart) through scipy: operator A M^-1
res = []
class Op(spl.LinearOperator):
    def __init__(self): super().__init__(dtype=float, shape=A.shape)
    def _matvec(self, v): return A @ Sc.apply(v)
y, info = spl.gmres(Op(), b, rtol=1e-11, restart=80, maxiter=1, callback=lambda rk: res.append(rk), callback_type='pr_norm')
print("GMRES(80) preconditioned residual per iteration:")
print(" ".join("%.0e" % v for v in res))

# where does the plateau residual live?
def bicg_upto(A, b, M, nit, seed=1234):
    n_ = b.size; x = np.zeros(n_); r = b.copy()
    rt = np.random.default_rng(seed).standard_normal(n_)
    rho = alpha = omega = 1.0; v = np.zeros(n_); p = np.zeros(n_)
    for it in range(nit):
        rho_new = rt @ r; beta = (rho_new / rho) * (alpha / omega)
        p = r + beta * (p - omega * v); y = M.apply(p); v = A @ y
        alpha = rho_new / (rt @ v); s = r - alpha * v; z = M.apply(s); t = A @ z
        omega = (t @ s) / (t @ t); x = x + alpha * y + omega * z; r = s - omega * t; rho = rho_new
    return x, r
x20, r20 = bicg_upto(A, b, Sc, 20)
Rp = np.abs(r20.reshape(nx[0], nx[1], 3)[:, :, 2])
tot = np.sum(Rp ** 2)
print("share of |r_p|^2 in: first row %.3f last cell row %.3f first col %.3f last cell col %.3f corners(3x3) %.3f, rows 1..3 %.3f" % (
    np.sum(Rp[0] ** 2) / tot, np.sum(Rp[nx[0] - 2] ** 2) / tot, np.sum(Rp[:, 0] ** 2) / tot, np.sum(Rp[:, nx[1] - 2] ** 2) / tot,
    (np.sum(Rp[:3, :3] ** 2) + np.sum(Rp[:3, -4:] ** 2) + np.sum(Rp[-4:, :3] ** 2) + np.sum(Rp[-4:, -4:] ** 2)) / tot, np.sum(Rp[1:4] ** 2) / tot))
idx = np.argsort(Rp.ravel())[::-1][:12]
print("largest |r_p| at", [(int(i // nx[1]), int(i % nx[1]), float("%.2e" % Rp.ravel()[i])) for i in idx])
prof = np.sqrt(np.mean(Rp ** 2, axis=1)); print("row profile (every 16th):", " ".join("%.1e" % v for v in prof[::16]))
prof = np.sqrt(np.mean(Rp ** 2, axis=0)); print("col profile (every 16th):", " ".join("%.1e" % v for v in prof[::16]))
# correlation with viscosity
en = np.array(fen); en[~np.isfinite(en)] = np.nanmean(fen)
lg = np.log10(en[:nx[0]-1, :nx[1]-1]); rr = Rp[:nx[0]-1, :nx[1]-1]
for lo, hi in ((20, 20.5), (20.5, 21), (21, 21.5), (21.5, 22), (22, 22.5), (22.5, 23.01)):
    m = (lg >= lo) & (lg < hi)
    if m.any(): print("  log10(eta) in [%.1f,%.1f): cells %6d rms |r_p| %.2e" % (lo, hi, m.sum(), np.sqrt(np.mean(rr[m] ** 2))))
