"""Smoothing-count sweep of the multigrid preconditioner on two models (mantle: T-dependent viscosity 1e20..1e23;
block: 1e3 falling block), resident steps at n x n.  One process per setting (the knobs are read once per process):
    python tools/tune_nu.py <mantle|block> <n> <nu0> <nu>        e.g.  mantle 2049 1,1 2,2
prints the mean Stokes time and iteration count of steps 2..4."""
import sys, os
model, n, nu0, nu = sys.argv[1], int(sys.argv[2]), sys.argv[3], sys.argv[4]
os.environ["PYLAMP_MG_NU0"] = nu0; os.environ["PYLAMP_MG_NU"] = nu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver
nx = [n, n]; L = [660e3, 660e3]
rng = np.random.default_rng(1)
opt = driver.Options()
if model == "block":
    tr_x, tr_f = driver.falling_block_tracers(nx, L, 16, rng)
    opt.do_heatdiff = False; opt.tdep_rho = False; opt.tdep_eta = False
else:
    tr_x, tr_f = driver.mantle_tracers(nx, L, 16, rng)
sim = driver.Simulation(nx, L, tr_x, tr_f, opt)
ms, its, conv = [], [], []
for k in range(4):
    r = sim.step()
    if k >= 1: ms.append(r["ms_stokes"]); its.append(r["stokes"]["iterations"]); conv.append(r["stokes"]["converged"])
print("%-7s %5d nu0 %s nu %s  stokes %.1f ms  its %.1f  conv %d" % (model, n, nu0, nu, np.mean(ms), np.mean(its), min(conv)), flush=True)
