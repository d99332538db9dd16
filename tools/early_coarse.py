"""NumPy-prototype experiment: the coarse levels >= K of the V-cycle take R^K f instead of the restricted residual of level K-1
(additive early coarse branch), BiCGStab iteration counts against the standard V-cycle.  python tools/early_coarse.py [n]"""
import sys, time
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver
from oracle import pylamp_oracle as O, proto_stokes_solver as PS

n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
nx = [n, n]; L = [660e3, 660e3]
grid = [np.linspace(0, L[0], n), np.linspace(0, L[1], n)]
rng = np.random.default_rng(5)
tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng); O.property_update(tr_f, True, True)
frho, fes = O.trac2grid(tr_x, tr_f[:, [0, 1]], grid, nx, [5, 6])
fen, = O.trac2grid(tr_x, tr_f[:, [1]], O.gridmp_of(grid), nx, [6])
bc = [1, 1, 1, 1]

def vcycle_early(Ls, l, fz, fx, gz, gx, nu, K, eK, damp_c=1.0):
    """levels < K: standard, except that level K-1 takes the precomputed coarse solution eK instead of restricting its residual"""
    L = Ls[l]
    vz = np.zeros_like(fz); vx = np.zeros_like(fx)
    PS.smooth(L, vz, vx, fz, fx, nu[0], gz, gx)
    if l == K - 1:
        ez, ex = eK
    else:
        yz, yx = PS.apply(L, vz, vx)
        cz, cx = PS.restrict(L, Ls[l + 1], (fz - yz) * L.mz, (fx - yx) * L.mx)
        ez, ex = vcycle_early(Ls, l + 1, cz, cx, None, None, nu, K, eK, damp_c)
    pz, px = PS.prolong(L, Ls[l + 1], ez, ex)
    d = damp_c if l == K - 1 else 1.0
    vz += d * pz; vx += d * px
    PS.smooth(L, vz, vx, fz, fx, nu[1], gz, gx)
    return vz, vx

class Early(PS.Precond):
    K = 2; damp_c = 1.0
    def apply(self, r):
        self.napply += 1
        nx = self.nx
        rz, rx, rp = PS.split(r, nx)
        zp = self.Sinv * rp
        for i0 in (0, nx[0] - 2):
            zp[i0, 0] = zp[i0, 1] - rp[i0, 0] / self.Kb
            zp[i0, nx[1] - 2] = zp[i0, nx[1] - 3] - rp[i0, nx[1] - 2] / self.Kb
        rv = np.stack([rz, rx], axis=2).reshape(-1) - self.Avp @ zp.reshape(-1)
        R = rv.reshape(nx[0], nx[1], 2)
        L0 = self.Ls[0]
        fz = np.pad(R[:, :, 0], 1) * L0.mz; fx = np.pad(R[:, :, 1], 1) * L0.mx
        gz = np.pad(R[:, :, 0], 1) / self.Kc; gx = np.pad(R[:, :, 1], 1) / self.Kc
        # early branch: restrict the right-hand side itself K times, solve from level K down
        cz, cx = fz, fx
        for l in range(self.K):
            cz, cx = PS.restrict(self.Ls[l], self.Ls[l + 1], cz * self.Ls[l].mz, cx * self.Ls[l].mx)
        eK = PS.vcycle(self.Ls, self.K, cz, cx, None, None, self.nu)
        vz, vx = vcycle_early(self.Ls, 0, fz, fx, gz, gx, self.nu, self.K, eK, self.damp_c)
        return PS.join(vz[1:-1, 1:-1], vx[1:-1, 1:-1], zp)

def run(name, M):
    Sc = PS.Scaled(M, nx, grid)
    t = time.time()
    x, it, res = PS.bicgstab(Sc.A, Sc.b, Sc, rtol=1e-10, maxit=200)
    print("%-28s its %3d  res %.2e  (%.1f s)" % (name, it, res, time.time() - t), flush=True)

nu = (2, 2)
run("standard V(2,2)", PS.Precond(nx, grid, fes, fen, frho, bc, nu=nu))
for K in (2, 3, 4):
    for d in (1.0,):
        M = Early(nx, grid, fes, fen, frho, bc, nu=nu); M.K = K; M.damp_c = d
        run("early coarse K=%d damp %.1f" % (K, d), M)
