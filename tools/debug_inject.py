import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver
nx = [65, 81]; L = [660e3, 820e3]
rng = np.random.default_rng(9)
tr_x, tr_f = driver.falling_block_tracers(nx, L, 10, rng)
for fence in (True, False):
    for mod in (False, True):
        X = tr_x.copy()
        if mod:
            X[7:60, 0] = -2000.0; X[100:140, 1] = -1500.0
        opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, tracdens=10, tracdens_min=7, inject_unique_ids=True, tracs_fence_enabled=fence)
        sim = driver.Simulation(nx, L, X, tr_f, opt)
        cen = sim.census()
        rep = sim.step()
        print("fence", fence, "mod", mod, "deficient cells before", int((cen < 7).sum()), "ninj", rep["ninjected"], "nrem", rep["nremoved"], "ntrac", rep["ntrac"], "conv", rep["stokes"]["converged"])
        sim.close()
