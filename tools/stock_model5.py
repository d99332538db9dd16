"""Stock pylamp2.py configuration (model 5, pylamp2.py:37-40,218-242): 201x41 nodes, L=[1,0.2],
45 markers/node, sphere of eta 1e12 in eta 1e2 (contrast 1e10), heat off.  Device step vs oracle."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver
from pylamp_amd.pylamp_const import *
from oracle import pylamp_oracle as O
nx = [201, 41]; L = [1.0, 0.2]
rng = np.random.default_rng(5)
n = int(np.prod(nx)) * 45
tr_x = rng.random((n, 2)) * np.array(L)
tr_f = np.zeros((n, NFTRAC)); tr_f[:, TR__ID] = np.arange(n)
contrast_hi = float(sys.argv[1]) if len(sys.argv) > 1 else 1e12
tr_f[:, TR_RH0] = 1420; tr_f[:, TR_MAT] = 1; tr_f[:, TR_ET0] = 1e2
idx = (tr_x[:, IX] - 0.1) ** 2 + (tr_x[:, IZ] - 0.2) ** 2 < 0.01 ** 2
tr_f[idx, TR_RH0] = 1470; tr_f[idx, TR_MAT] = 2; tr_f[idx, TR_ET0] = contrast_hi
rt = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-10
opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False, tracdens=45, tracdens_min=25, stokes_rtol=rt, stokes_maxit=1500)
sim = driver.Simulation(nx, L, tr_x, tr_f, opt)
st = dict(nx=nx, L=L, grid=[np.linspace(0, L[0], nx[0]), np.linspace(0, L[1], nx[1])], tr_x=tr_x.copy(), tr_f=tr_f.copy())
cfg = O.StepConfig(do_heatdiff=False, tdep_rho=False, tdep_eta=False)
rel = lambda a, b: np.linalg.norm(np.ravel(a) - np.ravel(b)) / np.linalg.norm(np.ravel(b))
for it in (1, 2):
    t0 = time.time(); rep = sim.step(); tg = time.time() - t0
    t0 = time.time(); out = O.step(st, cfg, it); tc = time.time() - t0
    print("rtol %.0e contrast %.0e step %d: GPU %.3fs (stokes its %d conv %d res %.1e) CPU %.2fs | velz err %.2e velx err %.2e tstep ratio %.6f injected %d" % (
        rt, contrast_hi / 1e2, it, tg, rep["stokes"]["iterations"], rep["stokes"]["converged"], rep["stokes"]["rel_residual"], tc,
        rel(sim.field("velz"), out["velz"]), rel(sim.field("velx"), out["velx"]), rep["tstep"] / out["tstep"], rep["ninjected"]), flush=True)
