"""the CPU oracle (numerically the reference path: NumPy scatter/gather, scipy spsolve) against the
resident GPU step, same seeded mantle model, on this box.  Usage: python tools/cpu_table.py [sizes...]"""
import sys, os, time, platform
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver
from oracle import pylamp_oracle as O

import json
args = [a for a in sys.argv[1:] if not a.startswith("--json=")]
jpath = ([a[7:] for a in sys.argv[1:] if a.startswith("--json=")] or [None])[0]
sizes = [int(a) for a in args] or [41, 129, 257, 513]
rows = []
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
    cpu_t = []
    for it in (1, 2):
        t0 = time.perf_counter(); c0 = time.process_time(); out = O.step(st, cfg, it); tc.append(time.perf_counter() - t0)
        cpu_t.append(time.process_time() - c0)
        rep = sim.step(); tg.append(rep["ms_total"])
        vz = sim.field("velz"); vx = sim.field("velx")
        err = max(err, np.sqrt((np.sum((vz - out["velz"]) ** 2) + np.sum((vx - out["velx"]) ** 2)) / (np.sum(out["velz"] ** 2) + np.sum(out["velx"] ** 2))))
    sim.close()
    print("| %d² | %d | %.2f | %.2f | %.0f× | %.1e |" % (n, tr_x.shape[0], tc[1], tg[1], 1e3 * tc[1] / tg[1], err), flush=True)
    rows.append({"nodes": "%dx%d" % (n, n), "tracers": int(tr_x.shape[0]), "cpu_oracle_s_per_step": round(tc[1], 3), "gpu_ms_per_step": round(tg[1], 3),
                 "cpu_cell_updates_per_s": round((n - 1) ** 2 / tc[1], 1), "gpu_cell_updates_per_s": round((n - 1) ** 2 / (tg[1] * 1e-3), 1),
                 "cpu_cores_busy": round(cpu_t[1] / tc[1], 2), "velocity_rel_l2_gpu_vs_oracle": float("%.3g" % err)})
if jpath:
    json.dump({"host": cpu, "logical_cpus": os.cpu_count(), "what": "CPU oracle (NumPy scatter / gather / RK4, scipy spsolve = numerically the reference path), one Python "
               "process, BLAS / SuperLU threading as installed; cpu_cores_busy = process CPU time / wall time of the step; second step of each run; "
               "seeded mantle model, 16 markers per node; the direct solve is infeasible beyond ~1025^2", "rows": rows}, open(jpath, "w"), indent=1)
