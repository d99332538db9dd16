"""Step a bench configuration one step at a time and print each report (which step fails, with what solver statistics):
    python tools/step_debug.py <nodes per side> <markers per node> <steps> [tracdens_min, default 9/16 of the density as in bench.py]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
n, dens, steps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
if len(sys.argv) > 4:
    import numpy as np
    from pylamp_amd import driver
    nx = [n, n]; L = [660e3, 660e3]
    sim = driver.Simulation(nx, L, options=driver.Options(tracdens=dens, tracdens_min=int(sys.argv[4])), device=0)
    tr_x, tr_f = driver.mantle_tracers(nx, L, dens, np.random.default_rng(20260103))
    sim.upload(tr_x, tr_f)
    del tr_x, tr_f
else:
    sim = bench.build_sim(n, dens, 20260103, 0, 0, 1, "strong")
for k in range(steps):
    try:
        r = sim.step()
    except Exception as e:
        print("step", k + 1, "FAILED:", str(e)[:200], flush=True)
        break
    s, h = r["stokes"], r["heat"]
    print("step %d dt %.4e limiter %s | stokes it %d conv %d res %.2e est %.2e direct %d | heat it %d conv %d res %.2e | injected %d | %.1f ms" % (
        k + 1, r["tstep"], r["limiter"], s["iterations"], s["converged"], s["rel_residual"], s["error_estimate"], s["used_direct"],
        h["iterations"], h["converged"], h["rel_residual"], r["ninjected"], r["ms_total"]), flush=True)
