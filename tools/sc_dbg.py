"""timing experiments on k_scatter_cells: PYLAMP_SC_DBG masks, HIP-event timed through the stage call"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, ctypes as C
from pylamp_amd import driver
n = 2049; nx = [n, n]; L = [660e3, 660e3]
tr_x, tr_f = driver.mantle_tracers(nx, L, 16, np.random.default_rng(20260103))
sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options())
del tr_x, tr_f
cfg = sim._config(); lib, hnd = sim.ctx.lib, sim.ctx.handle()
for dbg in [0, 0, 4, 2, 6, 1, 7, 8, 15, 0]:
    os.environ["PYLAMP_SC_DBG"] = str(dbg)
    t0 = time.perf_counter()
    sim.ctx.check(lib.pl_resident_scatter(hnd, C.byref(cfg), 1))
    print("dbg %2d: stage %.2f ms" % (dbg, 1e3 * (time.perf_counter() - t0)), flush=True)
