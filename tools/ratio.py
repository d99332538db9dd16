"""Which part of the residual carries the velocity error?  NumPy prototype on the 33 x 41 mantle fixture of the reference
trajectory: BiCGStab to several tolerances against the direct solve, and the error caused by each block of the final residual."""
import sys
import os; R = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from conftest import golden
from oracle import pylamp_oracle as O, proto_stokes_solver as PS
g = golden("traj_mantle33x41")
gz, gx = g["gz"], g["gx"]; nx = [gz.size, gx.size]; grid = [gz, gx]
tr_x = g["init_tr_x"].copy(); tr_f = g["init_tr_f"].copy()
O.property_update(tr_f, True, True)
frho, fes = O.trac2grid(tr_x, tr_f[:, [0, 1]], grid, nx, [5, 6])
fen, = O.trac2grid(tr_x, tr_f[:, [1]], O.gridmp_of(grid), nx, [6])
print("eta range %.2e %.2e" % (np.nanmin(fes), np.nanmax(fes)))
bc = [1, 1, 1, 1]
M = PS.Precond(nx, grid, fes, fen, frho, bc, nu=(2, 2)); Sc = PS.Scaled(M, nx, grid)
A, b = Sc.A, Sc.b
xs = spl.spsolve(sp.csc_matrix(A), b)
cls = O.stokes_row_class(nx)
def parts(v):
    V = v.reshape(nx[0], nx[1], 3); return V[:, :, 0], V[:, :, 1], V[:, :, 2]
vel = lambda v: np.concatenate([parts(v)[0].ravel(), parts(v)[1].ravel()])
# hydrostatic-like reference: use ||b|| here
for rtol in (1e-8, 1e-9, 1e-10, 1e-11):
    x, it, res = PS.bicgstab(A, b, Sc, rtol=rtol, maxit=300)
    r = b - A @ x
    rz, rx, rp = parts(r)
    e = x - xs
    print("rtol %.0e its %d  res %.2e | res parts z %.2e x %.2e p %.2e | vel err %.2e  p err %.2e" % (
        rtol, it, res, np.linalg.norm(rz) / np.linalg.norm(b), np.linalg.norm(rx) / np.linalg.norm(b), np.linalg.norm(rp) / np.linalg.norm(b),
        np.linalg.norm(vel(e)) / np.linalg.norm(vel(xs)), np.linalg.norm(parts(e)[2]) / np.linalg.norm(parts(xs)[2])))
# which residual component drives the error?  solve A e = r_part
x, it, res = PS.bicgstab(A, b, Sc, rtol=1e-10, maxit=300)
r = b - A @ x
lu = spl.splu(sp.csc_matrix(A))
for name, k in (("z-momentum", 0), ("x-momentum", 1), ("continuity", 2)):
    rr = np.zeros_like(r).reshape(nx[0], nx[1], 3); rr[:, :, k] = r.reshape(nx[0], nx[1], 3)[:, :, k]
    e = lu.solve(rr.ravel())
    print("  error from %-11s residual: %.2e (its share of |r|: %.2f)" % (name, np.linalg.norm(vel(e)) / np.linalg.norm(vel(xs)), np.linalg.norm(rr) / np.linalg.norm(r)))
print("|b| parts", [float(np.linalg.norm(p)) for p in parts(b)], " |x_vel| %.3e" % np.linalg.norm(vel(xs)))
