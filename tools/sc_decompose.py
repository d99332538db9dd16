"""Where does k_scatter_cells spend its time?  The fused scatter with parts switched off (PYLAMP_SC_DBG bits: 1 no tracer loop,
2 no row epilogue, 4 no emission, 8 no staging; results are wrong then, the timing is the point): python tools/sc_decompose.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np, ctypes as C
from pylamp_amd import driver
n = 2049; nx = [n, n]; L = [660e3, 660e3]
tr_x, tr_f = driver.mantle_tracers(nx, L, 16, np.random.default_rng(20260103))
sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options())
del tr_x, tr_f
cfg = sim._config(); lib, hnd = sim.ctx.lib, sim.ctx.handle()
for dbg in (0, 1, 2, 3, 4, 8, 9, 11, 15):
    os.environ["PYLAMP_SC_DBG"] = str(dbg)
    ts = []
    for r in range(4):
        t0 = time.perf_counter(); sim.ctx.check(lib.pl_resident_scatter(hnd, C.byref(cfg), 1)); ts.append(1e3 * (time.perf_counter() - t0))
    print("dbg %2d: stage %.2f ms (properties 0.8 + scatter + finalisation 0.08)" % (dbg, min(ts[1:])), flush=True)
sim.close()
