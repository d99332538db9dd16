"""PMC / kernel-trace target, ONE problem size per process (grid-stride kernels have the same launch size at every problem size: a
process that ran two sizes would merge their rows):
    python tools/pmc_target.py step [n=2049] [markers/node=16] [steps=3]     the resident time step of bench.py (mantle model): every kernel
                                                                            of the timed loop -- k_scatter_cells, k_rk4, k_gather, k_place_permute,
                                                                            k_property_update, k_heat_cheb, k_xrp_update_dev, the FP32-storage level-0 kernels ...
    python tools/pmc_target.py apply [n=2049]                                the Stokes stencil alone (row-scaled and plain) + the stream triad (calibration)
    python tools/pmc_target.py 3d [n=257]                                    one 3-D Stokes + heat solve pair (BASELINE config 5)
Run under rocprofv3 (--kernel-trace --stats, or --pmc ... in passes of their own): tools/pmc_passes.sh."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np

what = sys.argv[1] if len(sys.argv) > 1 else "step"
if what == "step":
    from pylamp_amd import driver
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2049
    dens = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    steps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
    nx = [n, n]; L = [660e3, 660e3]
    tr_x, tr_f = driver.mantle_tracers(nx, L, dens, np.random.default_rng(20260103))
    sim = driver.Simulation(nx, L, tr_x, tr_f, driver.Options(tracdens=dens, tracdens_min=(dens * 9) // 16))
    del tr_x, tr_f
    for _ in range(steps):
        r = sim.step()
        print("step %d: %.2f ms, %d iterations" % (r["it"], r["ms_total"], r["stokes"]["iterations"]), flush=True)
    sim.close()
elif what == "apply":
    from pylamp_amd import pylamp_stokes as S
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2049
    rng = np.random.default_rng(1)
    grid = [np.linspace(0, 660e3, n), np.linspace(0, 660e3, n)]
    es = 1e19 * 10 ** rng.uniform(0, 3, (n, n)); en = 1e19 * 10 ** rng.uniform(0, 3, (n, n))
    A, rhs = S.makeStokesMatrix([n, n], grid, es, en, 3300 + rng.uniform(-50, 50, (n, n)), [1, 1, 1, 1])
    ctx = A._ctx; ms = C.c_double()
    ctx.check(ctx.lib.pl_stokes_apply_bench(ctx.h, 10, C.byref(ms))); tp = ms.value
    ctx.check(ctx.lib.pl_stokes_apply_scaled_bench(ctx.h, 10, C.byref(ms))); ts = ms.value
    ctx.check(ctx.lib.pl_stream_triad_bench(ctx.h, 1 << 27, 5, C.byref(ms)))
    print("n=%d plain %.4f ms scaled %.4f ms triad %.3f ms" % (n, tp, ts, ms.value))
else:
    sys.argv = [sys.argv[0], "--config", "3d257", "--grid3", sys.argv[2] if len(sys.argv) > 2 else "257", "--steps", "1", "--warmup", "0"]
    import runpy
    runpy.run_path(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "bench.py"), run_name="__main__")
