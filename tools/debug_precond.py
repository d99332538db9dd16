import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver, _lib
Pz, Px = int(sys.argv[1]), int(sys.argv[2])
nx = [129, 257]; L = [660e3, 1320e3]
rng = np.random.default_rng(5)
tr_x, tr_f = driver.mantle_tracers(nx, L, 12, rng, perturb=60.0)
opt = driver.Options()
r = np.random.default_rng(1).standard_normal(3 * nx[0] * nx[1])
R = r.reshape(nx[0], nx[1], 3)
# zero on wall rows (ignored by the preconditioner anyway)
def prec(sim):
    z = np.empty_like(r)
    sim.ctx.check(sim.ctx.lib.pl_stokes_precond_apply(sim.ctx.handle(), _lib.dptr(r), _lib.dptr(z)))
    return z
sim = driver.Simulation(nx, L, tr_x, tr_f, opt)
sim.step()
z1 = prec(sim).reshape(nx[0], nx[1], 3)
sim.close()
vc = driver.VirtualCluster(nx, L, Pz, Px, tr_x, tr_f, opt)
vc.step()
z2 = vc.all(prec)[0].reshape(nx[0], nx[1], 3)
for q, nm in enumerate(("vz", "vx", "p")):
    d = np.abs(z2[:, :, q] - z1[:, :, q])
    i, j = np.unravel_index(np.argmax(d), d.shape)
    print(nm, "rel", np.linalg.norm(z2[:, :, q] - z1[:, :, q]) / np.linalg.norm(z1[:, :, q]), "max at", (i, j), z1[i, j, q], z2[i, j, q])
    rows = np.where(d.max(axis=1) > 1e-6 * np.abs(z1[:, :, q]).max())[0]; cols = np.where(d.max(axis=0) > 1e-6 * np.abs(z1[:, :, q]).max())[0]
    print("   rows with diff:", rows[:20], len(rows), " cols:", cols[:20], len(cols))
vc.close()
