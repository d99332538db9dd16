"""Randomised campaign: the resident GPU step against the oracle's step() on small random configurations
(grid sizes incl. non-coarsenable ones, wall types, heat on/off, T-dependence, uniform / graded grids, marker
densities).  Usage: python tools/fuzz_step.py [ncases] [seed] [cell aspect ratio dx/dz, default random 0.7..1.4]"""
import sys, os, traceback
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
DIST = "RANK" in os.environ and int(os.environ.get("WORLD_SIZE", "1")) > 1
if DIST:      # torchrun --nproc-per-node N tools/fuzz_step.py ...: N row slabs sharing GPU 0 through the gloo transport
    os.environ.setdefault("PYLAMP_DEVICE", "0")
    import torch.distributed as dist
    dist.init_process_group(backend="gloo")
R = int(os.environ.get("WORLD_SIZE", "1")) if DIST else 1
RANK = int(os.environ.get("RANK", "0")) if DIST else 0
from pylamp_amd import driver
from oracle import pylamp_oracle as O

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 20
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
rel = lambda a, b: float(np.linalg.norm(np.nan_to_num(a - b)) / max(np.linalg.norm(np.nan_to_num(b)), 1e-300))
# velocities of a model without any density contrast are pure round-off: measure them against 1e-13 m/s (3 um/yr)
relv = lambda a, b: float(np.linalg.norm(np.nan_to_num(a - b)) / max(np.linalg.norm(np.nan_to_num(b)), 1e-13 * np.sqrt(b.size)))
bad = 0
for case in range(ncases):
    lo_n, hi_n = (int(os.environ.get("FUZZ_NMIN", "17")), int(os.environ.get("FUZZ_NMAX", "90")))
    nz, nxx = int(rng.integers(lo_n, hi_n)), int(rng.integers(lo_n, hi_n))
    if DIST: nz = 16 * R * int(rng.integers(1, 4)) + 1        # even slabs of >= 8 rows that coarsen at least once
    aspect = float(rng.uniform(0.7, 1.4)) if len(sys.argv) <= 3 else float(sys.argv[3])      # cell aspect ratio dx/dz
    nx = [nz, nxx]; L = [660e3, 660e3 * (nxx - 1) / (nz - 1) * aspect]
    heat = bool(rng.integers(0, 2)); tdep = bool(rng.integers(0, 2)) and heat
    model = "mantle" if heat else ("block" if rng.integers(0, 2) else "mantle")
    bcz = [int(rng.integers(0, 2)), int(rng.integers(0, 2))]
    bc = [bcz[0], 1, bcz[1], 1]
    dens = int(rng.integers(12, 30))       # fewer leave empty nodes (NaN fields) on graded grids
    graded = bool(rng.integers(0, 2))
    trng = np.random.default_rng(1000 + case)
    if model == "block":
        tr_x, tr_f = driver.falling_block_tracers(nx, L, dens, trng)
    else:
        tr_x, tr_f = driver.mantle_tracers(nx, L, dens, trng)
    grid = None
    if graded:
        def g(n, Lx):
            h = 1.0 + float(rng.uniform(0.2, 2.0)) * (0.5 + 0.5 * np.sin(np.linspace(0, 2 * np.pi, n - 1) + float(rng.uniform(0, 6))))
            c = np.concatenate([[0.0], np.cumsum(h)]); c *= Lx / c[-1]; c[-1] = Lx
            return c
        grid = [g(nz, L[0]), g(nxx, L[1])]
    desc = "case %d: %dx%d L=%.2g,%.2g asp=%.2f %s heat=%d tdep=%d bc=%s dens=%d graded=%d" % (case, nz, nxx, L[0], L[1], aspect, model, heat, tdep, bc, dens, graded)
    try:
        # a node (of any staggering) without a single marker in reach is NaN in the reference too: not a test case
        gchk = grid if grid is not None else [np.linspace(0, L[0], nz), np.linspace(0, L[1], nxx)]
        gmp = O.gridmp_of(gchk)
        with O.rect_search(graded):
            # (the ghost row of k_z counts: the reference's heat time step takes np.max over it, pylamp2.py:340-341)
            empty = any(np.isnan(O.trac2grid(tr_x, tr_f[:, [1]] * 0 + 1.0, tg, nx, [5])[0][:nz - 1 + sz, :nxx - 1 + sx]).any()
                        for tg, sz, sx in (([gchk[0], gchk[1]], 1, 1), ([gmp[0], gmp[1]], 0, 0), ([gmp[0], gchk[1]], 1, 1), ([gchk[0], gmp[1]], 1, 0)))
        if empty:
            if RANK == 0: print("skip " + desc + "  (a node has no marker in reach)", flush=True)
            continue
        opt = driver.Options(do_heatdiff=heat, tdep_rho=tdep, tdep_eta=tdep, bcstokes=bc)
        sim = driver.Simulation(nx, L, tr_x, tr_f, opt, grid=grid)
        st = dict(nx=nx, L=L, grid=sim.grid, tr_x=tr_x.copy(), tr_f=tr_f.copy())
        cfg = O.StepConfig(do_heatdiff=heat, tdep_rho=tdep, tdep_eta=tdep, bcstokes=bc)
        worst = {}
        for it in (1, 2):
            rep = sim.step()
            with O.rect_search(graded):
                out = O.step(st, cfg, it)
            X = sim.gather_tracers()[0] if DIST else sim.tracers()[0]
            e = dict(vz=relv(sim.field("velz"), out["velz"]), vx=relv(sim.field("velx"), out["velx"]),
                     dt=abs(rep["tstep"] - out["tstep"]) / out["tstep"], x=rel(X, st["tr_x"]) if X.shape == st["tr_x"].shape else 1.0)
            if heat: e["T"] = rel(sim.field("temp"), out["temp"])
            for k, v in e.items(): worst[k] = max(worst.get(k, 0.0), v)
            if not rep["stokes"]["converged"]: worst["noconv"] = 1.0
        sim.close()
        ok = all(v < 1e-5 for k, v in worst.items())
        if RANK == 0: print(("ok   " if ok else "FAIL ") + desc + "  " + " ".join("%s=%.1e" % kv for kv in worst.items()), flush=True)
        bad += 0 if ok else 1
    except Exception as ex:
        bad += 1
        print("EXC  rank %d " % RANK + desc + "  " + repr(ex)[:200], flush=True)
        if DIST: raise          # the other ranks would wait forever in the next collective
if RANK == 0: print("failures:", bad)
if DIST:
    dist.barrier(); dist.destroy_process_group()
