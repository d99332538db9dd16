import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver
for nz, nxx in ((257, 257), (1025, 257), (2049, 257), (4097, 513)):
    nx = [nz, nxx]; L = [660e3 * (nz - 1) / (nxx - 1), 660e3]
    rng = np.random.default_rng(1)
    tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng)
    sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options())
    for k in range(3):
        r = sim.step()
        print(nx, "step", k, "stokes ms %.1f its %d res %.2e conv %d  total ms %.1f" % (r["ms_stokes"], r["stokes"]["iterations"], r["stokes"]["rel_residual"], r["stokes"]["converged"], r["ms_total"]), flush=True)
    sim.close()
