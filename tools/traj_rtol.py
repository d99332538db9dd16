"""Velocity error of the resident step against the reference-driver trajectory fixture, per step, for several Stokes tolerances."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..', 'tests'))
import numpy as np
from conftest import golden, relerr
from pylamp_amd import driver
for name, heat in (("traj_mantle33x41", True), ("traj_block41", False)):
    g = golden(name)
    gz, gx = g["gz"], g["gx"]
    nx = [gz.size, gx.size]; L = [gz[-1], gx[-1]]
    for rtol in (1e-6, 1e-7, 1e-8, 1e-10, 1e-12):
        opt = driver.Options(do_heatdiff=heat, tdep_rho=heat, tdep_eta=heat); opt.stokes_rtol = rtol
        if os.environ.get('HEAT_RTOL'): opt.heat_rtol = float(os.environ['HEAT_RTOL'])
        sim = driver.Simulation(nx, L, g["init_tr_x"], g["init_tr_f"], opt)
        out = []
        for it in range(1, int(g["nsteps"]) + 1):
            rep = sim.step()
            out.append("%.2e/%.1e/%.1e/%d" % (relerr(sim.field("velz"), g["s%d_velz" % it]), rep["stokes"]["rel_residual"],
                                              rep["stokes"]["error_estimate"], rep["stokes"]["iterations"]))
        print(name, "rtol %.0e  err/res/est/its per step:" % rtol, " ".join(out), flush=True)
