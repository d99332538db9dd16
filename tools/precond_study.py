"""Experiment (not part of the product): which half of the block preconditioner limits BiCGStab?
Compares, in the NumPy prototype, one V-cycle against an exact velocity-block solve."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, scipy.sparse as sp, scipy.sparse.linalg as spl
from pylamp_amd import driver
from oracle import pylamp_oracle as O, proto_stokes_solver as PS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 129
model = sys.argv[2] if len(sys.argv) > 2 else "mantle"
nx = [n, n]; L = [660e3, 660e3]
grid = [np.linspace(0, L[0], n), np.linspace(0, L[1], n)]
rng = np.random.default_rng(5)
if model == "block":
    tr_x, tr_f = driver.falling_block_tracers(nx, L, 16, rng); O.property_update(tr_f, False, False)
else:
    tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng); O.property_update(tr_f, True, True)
frho, fes = O.trac2grid(tr_x, tr_f[:, [0, 1]], grid, nx, [5, 6])
fen, = O.trac2grid(tr_x, tr_f[:, [1]], O.gridmp_of(grid), nx, [6])
bc = [1, 1, 1, 1]
print("viscosity range %.2e .. %.2e" % (np.nanmin(fes), np.nanmax(fes)))

class Exact(PS.Precond):
    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        N = nx[0] * nx[1]
        iv = np.sort(np.concatenate([np.arange(N) * 3, np.arange(N) * 3 + 1]))
        self.lu = spl.splu(self.A[iv][:, iv].tocsc())
    def apply(self, r):
        self.napply += 1
        rz, rx, rp = PS.split(r, nx)
        zp = self.Sinv * rp
        for i0 in (0, nx[0] - 2):
            zp[i0, 0] = zp[i0, 1] - rp[i0, 0] / self.Kb
            zp[i0, nx[1] - 2] = zp[i0, nx[1] - 3] - rp[i0, nx[1] - 2] / self.Kb
        rv = np.stack([rz, rx], axis=2).reshape(-1) - self.Avp @ zp.reshape(-1)
        v = self.lu.solve(rv).reshape(nx[0], nx[1], 2)
        return PS.join(v[:, :, 0], v[:, :, 1], zp)

class Multi(PS.Precond):
    ncyc = 2
    def apply(self, r):
        self.napply += 1
        rz, rx, rp = PS.split(r, nx)
        zp = self.Sinv * rp
        for i0 in (0, nx[0] - 2):
            zp[i0, 0] = zp[i0, 1] - rp[i0, 0] / self.Kb
            zp[i0, nx[1] - 2] = zp[i0, nx[1] - 3] - rp[i0, nx[1] - 2] / self.Kb
        rv = np.stack([rz, rx], axis=2).reshape(-1) - self.Avp @ zp.reshape(-1)
        R = rv.reshape(nx[0], nx[1], 2)
        L0 = self.Ls[0]
        fz = np.pad(R[:, :, 0], 1); fx = np.pad(R[:, :, 1], 1)
        gz = fz / self.Kc; gx = fx / self.Kc
        vz, vx = PS.vcycle(self.Ls, 0, fz * L0.mz, fx * L0.mx, gz, gx, self.nu)
        for _ in range(self.ncyc - 1):
            yz, yx = PS.apply(L0, vz, vx)
            ez, ex = PS.vcycle(self.Ls, 0, (fz - yz) * L0.mz, (fx - yx) * L0.mx, None, None, self.nu)
            vz += ez; vx += ex
        return PS.join(vz[1:-1, 1:-1], vx[1:-1, 1:-1], zp)

for name, cls in (("1 V(2,2)", PS.Precond), ("2 V(2,2)", Multi), ("exact A_vv", Exact)):
    M = cls(nx, grid, fes, fen, frho, bc)
    Sc = PS.Scaled(M, nx, grid)
    t = time.time()
    x, it, res = PS.bicgstab(Sc.A, Sc.b, Sc, rtol=1e-10, maxit=150)
    print("%-12s its %3d  res %.2e  (%.1f s)" % (name, it, res, time.time() - t), flush=True)
