import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver
Pz, Px = int(sys.argv[1]), int(sys.argv[2])
nx = [129, 257]; L = [660e3, 1320e3]
rng = np.random.default_rng(5)
tr_x, tr_f = driver.mantle_tracers(nx, L, 12, rng, perturb=60.0)
opt = driver.Options()
sim = driver.Simulation(nx, L, tr_x, tr_f, opt)
r1 = sim.step()
names = ("rho", "etas", "etan", "cp", "kz", "kx", "f_T", "H", "velz", "velx", "pres", "temp")
ref = {k: sim.field(k) for k in names}
X1, F1 = sim.tracers()
sim.close()
vc = driver.VirtualCluster(nx, L, Pz, Px, tr_x, tr_f, opt)
reps = vc.step()
print("tstep", r1["tstep"], reps[0]["tstep"], r1["tstep_heat"], [r["tstep_heat"] for r in reps], r1["tstep_stokes"], reps[0]["tstep_stokes"])
print("its", r1["stokes"]["iterations"], [r["stokes"]["iterations"] for r in reps], [r["stokes"]["converged"] for r in reps])
for k in names:
    a = vc.field(k); b = ref[k]
    m = np.isfinite(b) & np.isfinite(a)
    d = np.abs(a - b); d[~m] = 0
    i, j = np.unravel_index(np.argmax(d), d.shape)
    print("%5s nanmask_equal=%s maxdiff=%.3e at (%d,%d) ref=%.6e got=%.6e rel=%.2e" % (k, np.array_equal(np.isfinite(a), np.isfinite(b)), d.max(), i, j, b[i, j], a[i, j], np.linalg.norm((a - b)[m]) / np.linalg.norm(b[m])))
X, F, V = vc.tracers()
print("ntrac", X.shape, X1.shape, "pos rel", np.linalg.norm(X - X1) / np.linalg.norm(X1) if X.shape == X1.shape else None)
vc.close()
