"""the CPU oracle (numerically the reference path: NumPy scatter/gather, scipy spsolve) against the
resident GPU step, same seeded mantle model, on this box.  Usage: python tools/cpu_table.py [sizes...]"""
import sys, os, time, platform
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver
from oracle import pylamp_oracle as O

sizes = [int(a) for a in sys.argv[1:]] or [41, 129, 257, 513]
cpu = ""
try:
    cpu = [l.split(":")[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
except Exception:
    cpu = platform.processor()
print("host: %s, %d logical CPUs; oracle runs in one process" % (cpu, os.cpu_count()), flush=True)
print("| nodes | tracers | CPU oracle s/step | GPU ms/step | ratio | velocity rel-L2 (GPU vs oracle) |", flush=True)
print("|---|---|---|---|---|---|", flush=True)
for n in sizes:
    nx = [n, n]; L = [660e3, 660e3]
    rng = np.random.default_rng(20260101 + n)
    tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng)
    sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options())
    st = dict(nx=nx, L=L, grid=[np.linspace(0, L[0], n), np.linspace(0, L[1], n)], tr_x=tr_x.copy(), tr_f=tr_f.copy())
    cfg = O.StepConfig()
    tc = []; tg = []; err = 0.0
    for it in (1, 2):
        t0 = time.perf_counter(); out = O.step(st, cfg, it); tc.append(time.perf_counter() - t0)
        rep = sim.step(); tg.append(rep["ms_total"])
        vz = sim.field("velz"); vx = sim.field("velx")
        err = max(err, np.sqrt((np.sum((vz - out["velz"]) ** 2) + np.sum((vx - out["velx"]) ** 2)) / (np.sum(out["velz"] ** 2) + np.sum(out["velx"] ** 2))))
    sim.close()
    print("| %d² | %d | %.2f | %.2f | %.0f× | %.1e |" % (n, tr_x.shape[0], tc[1], tg[1], 1e3 * tc[1] / tg[1], err), flush=True)
