"""Falling-block model (the values of pylamp2.py:177-183) on the GPU-resident driver; writes the reference's snapshot
files (griddata.NNNNNN.npz / tracs.NNNNNN.npz, readable by the reference's pylamp_post.py).

    python examples/falling_block.py [n=257] [steps=20] [outdir=out]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from pylamp_amd import driver                                                    # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 257
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
outdir = sys.argv[3] if len(sys.argv) > 3 else "out"
os.makedirs(outdir, exist_ok=True)

nx = [n, n]; L = [660e3, 660e3]
tr_x, tr_f = driver.falling_block_tracers(nx, L, 16, np.random.default_rng(1))    # 16 markers per node
opt = driver.Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False,          # isothermal, constant properties
                     tracdens=16, tracdens_min=6)                                 # refill depleted cells
sim = driver.Simulation(nx, L, tr_x, tr_f, opt)
for it in range(1, steps + 1):
    rep = sim.step()
    print("step %3d  t = %8.3f Myr  dt = %.3e s (%s)  Stokes %3d its %s  %6.1f ms" %
          (it, sim.totaltime / 3.15576e13, rep["tstep"], rep["limiter"], rep["stokes"]["iterations"],
           "ok" if rep["stokes"]["converged"] else "NOT CONVERGED", rep["ms_total"]), flush=True)
    if it % 10 == 0 or it == steps:
        sim.write_snapshot(outdir)
sim.close()
