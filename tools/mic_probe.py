"""Runs the marker stages of the resident step a few times on the bench configuration (2049^2 nodes, 16 markers per node) --
target for kernel traces and PMC passes of the marker kernels alone (no Stokes solve):
    rocprofv3 --kernel-trace --stats -d out -- python3 tools/mic_probe.py [n] [dens] [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
dens = int(sys.argv[2]) if len(sys.argv) > 2 else 16
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nx = [n, n]; L = [660e3, 660e3]
tr_x, tr_f = driver.mantle_tracers(nx, L, dens, np.random.default_rng(20260103))
sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options())
del tr_x, tr_f
h = L[0] / (n - 1)
g = np.concatenate([[-0.5 * h], (np.arange(n - 1) + 0.5) * h, [L[0] + 0.5 * h]])
Zp, Xp = np.meshgrid(g, g, indexing="ij")
Vz = 1e-9 * np.sin(np.pi * Zp / L[0]) * np.cos(7 * np.pi * Xp / L[1]); Vx = -1e-9 / 7 * np.cos(np.pi * Zp / L[0]) * np.sin(7 * np.pi * Xp / L[1])
import ctypes as C
from pylamp_amd import _lib
cfg = sim._config()
lib, hnd = sim.ctx.lib, sim.ctx.handle()
for r in range(reps):
    t0 = time.perf_counter()
    sim.ctx.check(lib.pl_resident_scatter(hnd, C.byref(cfg), 1))
    t1 = time.perf_counter()
    fT = sim.field("f_T")
    newT = fT + 5 * np.sin(np.arange(n) / 50.0)[:, None]
    nt = _lib.f64(newT)
    t2 = time.perf_counter()
    sim.ctx.check(lib.pl_resident_temp_to_tracers(hnd, C.byref(cfg), 0, _lib.dptr(nt), 0.3 * h * h * 3300 * 1250 / 4.0))
    t3 = time.perf_counter()
    Lc = (C.c_double * 2)(*L)
    vz = _lib.f64(Vz); vx = _lib.f64(Vx)
    sim.ctx.check(lib.pl_resident_rk4(hnd, _lib.dptr(vz), _lib.dptr(vx), 0.5 * h / 1e-9, 1, Lc))
    t4 = time.perf_counter()
    print("rep %d: scatter %.2f ms, temp_to_tracers %.2f ms (incl. upload), rk4+sort %.2f ms (incl. upload)" %
          (r, 1e3 * (t1 - t0), 1e3 * (t3 - t2), 1e3 * (t4 - t3)), flush=True)
sim.close()
