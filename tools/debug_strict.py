import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver
g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests', 'golden', 'traj_surfstab41.npz'))
gz, gx = g["gz"], g["gx"]
nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
rel = lambda a, b: np.linalg.norm(a - b) / np.linalg.norm(b)
opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, surface_stabilization=True, stokes_maxit=int(os.environ.get("MAXIT", "400")))
sim = driver.Simulation(nx, L, g["init_tr_x"], g["init_tr_f"], opt)
for it in range(1, int(g["nsteps"]) + 1):
    rep = sim.step()
    print(it, "resolves", rep["stokes_resolves"], "its", rep["stokes"]["iterations"], "conv", rep["stokes"]["converged"], "res", rep["stokes"]["rel_residual"],
          "velz err", rel(sim.field("velz"), g["s%d_velz" % it]), "tstep", rep["tstep"], "limiter", rep["limiter"])
