#!/usr/bin/env python3
"""Benchmark of the PyLamp hot path on MI355X: full time step (MIC scatter, Stokes solve, heat
solve, temperature to tracers, RK4 advection) at BASELINE config 3 — 2049x2049 nodes
(2048x2048 cells), 16 markers per node (67.2 M tracers), T-dependent mantle model with heat.

    python bench.py --gpus N --steps K --warmup W

Prints ONE JSON line (rank 0).  `value` = cell-updates/s = cells * steps / time, whole job.
Extra objects: `roofline` (matrix-free Stokes stencil, algorithmic 64 B/node/apply, timed with
HIP events on the library's own stream) and `cpu_baseline` (the NumPy/SciPy oracle = numerically
the reference path, on a bounded sample, rank 0 at N=1 only).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 GB/s achievable


def build_sim(n, tracdens, seed, device, rank, world, scaling, model="mantle"):
    """strong (default, what BASELINE.json's metric names: the 2048^2-cell problem at 1/2/4/8 GPUs): the SAME n x n
    grid for every N, split into Pz x Px blocks (1x2, 2x2, 2x4).  weak (--scaling weak): every GPU holds one
    (n-1) x (n-1)-cell block of a ((n-1) Pz + 1) x ((n-1) Px + 1) grid over a correspondingly larger domain (square
    cells).  Every rank draws the tracers of its own block; N = 1 is BASELINE's config 3 either way."""
    from pylamp_amd import driver
    from pylamp_amd.parallel import choose_layout
    Pz, Px = choose_layout(world)
    if scaling == "weak":
        nx = [(n - 1) * Pz + 1, (n - 1) * Px + 1]; L = [660e3 * Pz, 660e3 * Px]
    else:
        nx = [n, n]; L = [660e3, 660e3]
    # census + injection inside the timed step, like the reference (pylamp2.py:39-40,588-633; its stock values are
    # 45 / 25): cells below 9 markers are refilled to 16
    opt = driver.Options(tracdens=tracdens, tracdens_min=(tracdens * 9) // 16)
    if model == "block":            # SURVEY 8d C3 "same physics as C1": the 10^3 falling block of pylamp2.py:172-183, heat off
        opt = driver.Options(tracdens=tracdens, tracdens_min=(tracdens * 9) // 16, do_heatdiff=False, tdep_rho=False, tdep_eta=False)
    sim = driver.Simulation(nx, L, options=opt, device=device)
    rng = np.random.default_rng(seed + rank)
    zr, xr = sim.block()
    per_rank = nx[0] * nx[1] * tracdens // max(world, 1)
    if model == "block":
        if world > 1:
            raise Exception("--model block is a single-GPU secondary mode")
        tr_x, tr_f = driver.falling_block_tracers(nx, L, tracdens, rng)
    else:
        tr_x, tr_f = driver.mantle_tracers(nx, L, tracdens, rng, zrange=None if world == 1 else zr, xrange=None if world == 1 else xr,
                                           id0=rank * per_rank)
    sim.upload(tr_x, tr_f)
    del tr_x, tr_f
    return sim


def _apply_point(ctx, nodes, reps, traffic):
    """HIP-event timing of the resident Stokes stencil on this context: the row-scaled variant the Krylov solver
    launches and the plain operator; algorithmic bytes = 64 B/node (x 24 + etas 8 + etan 8 + y 24)."""
    ms = C.c_double()
    out = {}
    for name, fn in (("scaled", ctx.lib.pl_stokes_apply_scaled_bench), ("plain", ctx.lib.pl_stokes_apply_bench)):
        vals = []
        for _ in range(3):
            ctx.check(fn(ctx.h, reps, C.byref(ms)))
            vals.append(ms.value)
        avg = vals[-1]                                  # the LAST average is the reported one, the best is informative
        ach = 64.0 * nodes / (avg * 1e-3) / 1e9
        out[name] = {"achieved": round(ach, 1), "frac": round(ach / HBM_PEAK_GBS, 4), "avg_launch_ms": round(avg, 5),
                     "best_launch_ms": round(min(vals), 5), "traffic": traffic.get(name)}
    return out


def kernel_roofline(sim, reps):
    """`roofline` of the bench line: the dominant north-star kernel is the matrix-free Stokes stencil.  The figure
    reported at the top level is the ROW-SCALED variant (k_stokes_apply_v2<R, true>) -- the one BiCGStab launches twice
    per iteration -- on the bench grid; the plain operator, the 4097^2 point (working set 1.07 GB >> the 256 MiB
    Infinity Cache, so no replay is cache-served), the measured stream triad of this box and the multigrid smoother
    follow as sub-objects.  `traffic` = PMC bytes per launch from profiles/traffic.json (tools/make_traffic.py)."""
    from pylamp_amd import pylamp_stokes as S
    ctx = sim.ctx
    ms = C.c_double()
    _, ni_, _, nj_, _, _ = ctx.local_block()
    n = ni_ * nj_                           # nodes of this rank's block (the whole grid on one GPU)
    tj = {}
    tf = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tf):
        try:
            tj = json.load(open(tf))
        except Exception:
            tj = {}
    whole = sim.ctx.nranks == 1              # the traffic table and the 4097^2 point are single-GPU figures

    def tr(kernel, size):
        return tj.get(kernel, {}).get(str(size)) if whole else None
    here = _apply_point(ctx, n, reps, {"scaled": tr("k_stokes_apply_scaled", sim.nx[1]), "plain": tr("k_stokes_apply", sim.nx[1])})
    # measured stream triad (3 x 1 GiB arrays)
    triad = None
    try:
        nt = 1 << 27
        ctx.check(ctx.lib.pl_stream_triad_bench(ctx.h, nt, 20, C.byref(ms)))
        triad = 24.0 * nt / (ms.value * 1e-3) / 1e9
    except Exception:
        triad = None
    # the smoother: one Chebyshev sweep of the finest multigrid level (largest share of a step)
    sweep = None
    try:
        ctx.check(ctx.lib.pl_stokes_sweep_bench(ctx.h, reps, C.byref(ms)))
        sb = 80.0 * n                       # v 16 + v_prev 16 + f 16 + etas 8 + etan 8 + write 16 B per node
        sweep = {"kernel": "k_vv_sweep2<0> (Chebyshev sweep, finest level)", "bound": "hbm",
                 "achieved": round(sb / (ms.value * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                 "frac": round(sb / (ms.value * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "traffic": tr("k_vv_sweep2_cheb", sim.nx[1]),
                 "algorithmic_bytes_per_launch": sb, "avg_launch_ms": round(ms.value, 5), "launches_timed": reps}
    except Exception:
        sweep = None
    # the same stencil at 4097^2 on a second context (one GPU only: HBM-honest point)
    big = None
    if whole and not os.environ.get("PYLAMP_BENCH_NO_4097"):
        try:
            nb = 4097
            rng = np.random.default_rng(1)
            g = [np.linspace(0, 660e3, nb), np.linspace(0, 660e3, nb)]
            es = 1e19 * 10 ** rng.uniform(0, 3, (nb, nb)); en = 1e19 * 10 ** rng.uniform(0, 3, (nb, nb))
            A, _ = S.makeStokesMatrix([nb, nb], g, es, en, 3300 + rng.uniform(-50, 50, (nb, nb)), [1, 1, 1, 1])
            big = _apply_point(A._ctx, nb * nb, max(reps // 2, 10), {"scaled": tr("k_stokes_apply_scaled", nb), "plain": tr("k_stokes_apply", nb)})
            big["algorithmic_bytes_per_launch"] = 64.0 * nb * nb
            del A
            from pylamp_amd import _context
            _context.clear_contexts()
        except Exception as e:
            big = {"error": repr(e)}
    top = here["scaled"]
    out = {"kernel": "k_stokes_apply_v2<R,true> (row-scaled stencil, the in-solver variant)", "bound": "hbm",
           "achieved": top["achieved"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": top["frac"], "traffic": top["traffic"],
           "algorithmic_bytes_per_launch": 64.0 * n, "avg_launch_ms": top["avg_launch_ms"], "best_launch_ms": top["best_launch_ms"],
           "launches_timed": reps, "plain_operator": here["plain"], "at_4097": big,
           "triad_GBps": None if triad is None else round(triad, 1),
           "frac_of_triad": None if not triad else round(top["achieved"] / triad, 4),
           "traffic_commit": tj.get("_commit"), "smoother": sweep}
    return out


def cpu_baseline(n=257, tracdens=16, steps=1):
    """The oracle (NumPy assembly + scipy spsolve + NumPy MIC) on a bounded sample."""
    from oracle import pylamp_oracle as O
    from pylamp_amd import driver
    nx = [n, n]; L = [660e3, 660e3]
    rng = np.random.default_rng(20260103)
    tr_x, tr_f = driver.mantle_tracers(nx, L, tracdens, rng)
    st = dict(nx=nx, L=L, grid=[np.linspace(0, L[0], n), np.linspace(0, L[1], n)], tr_x=tr_x, tr_f=tr_f)
    cfg = O.StepConfig()
    t0 = time.perf_counter(); c0 = time.process_time()
    for it in range(1, steps + 1):
        O.step(st, cfg, it)
    dt = time.perf_counter() - t0
    busy = (time.process_time() - c0) / dt           # cores the process actually kept busy (NumPy is single-threaded, SuperLU / BLAS may not be)
    cells = (n - 1) * (n - 1)
    return {"value": round(cells * steps / dt, 1), "unit": "cell-updates/s", "cores": max(1, int(round(busy))), "cores_busy": round(busy, 2), "kind": "port",
            "sample": "full step (scatter, scipy spsolve Stokes, heat, gather, RK4), mantle model, %dx%d nodes, "
                      "%d markers/node, %d step(s), %.1f s; the direct solve is infeasible at 2049^2" %
                      (n, n, tracdens, steps, dt),
            "steps_per_s": round(steps / dt, 5)}


def layout3(world):
    """Pz x Px x Py of BASELINE config 5 on N GPUs: 2 -> 1x1x2, 4 -> 1x2x2, 8 -> 2x2x2 (PYLAMP_DECOMP3="PzxPxxPy" overrides)."""
    e = os.environ.get("PYLAMP_DECOMP3")
    if e:
        p = [int(v) for v in e.lower().split("x")]
        if len(p) == 3 and p[0] * p[1] * p[2] == world:
            return tuple(p)
    return {1: (1, 1, 1), 2: (1, 1, 2), 4: (1, 2, 2), 8: (2, 2, 2)}.get(world, (1, 1, world))


def bench_3d(args):
    """--config 3d257 (BASELINE config 5: "3D 257^3 staggered Stokes + heat, 1 -> 8 MI355X weak scaling"): 3-D staggered Stokes + heat
    -- a "step" is one Stokes solve (multigrid-preconditioned BiCGStab, default tolerance) plus one implicit heat solve on the same
    grid.  N > 1: every GPU holds one (n-1)^3-cell block of a Pz x Px x Py decomposition (weak scaling: the global grid is
    ((n-1) Pz + 1) x ((n-1) Px + 1) x ((n-1) Py + 1) nodes over a correspondingly larger domain).  No reference implementation exists in 3-D
    (parity unpinned; validation by extrusion / manufactured solutions / blocks against one rank in tests/test_hip_3d.py)."""
    from pylamp_amd import pylamp3d as P3
    rank, world, local_rank, dist, red_dev, barrier, device = init_dist(args)
    P = layout3(world)
    nb = args.n3
    n = [(nb - 1) * P[d] + 1 for d in range(3)]
    L = [660e3 * P[d] for d in range(3)]
    grid = [np.linspace(0, L[d], n[d]) for d in range(3)]
    mid = [np.append(0.5 * (g[1:] + g[:-1]), g[-1] + 0.5 * (g[-1] - g[-2])) for g in grid]
    def field(c):                         # T-dependent mantle: conductive profile + a 3-D perturbation
        Z, X, Y = np.meshgrid(*c, indexing="ij", sparse=True)
        return 273 + 1350 * np.clip(Z / L[0], 0, 1) + 60 * np.sin(3 * np.pi * X / L[1]) * np.sin(np.pi * Z / L[0]) * np.cos(2 * np.pi * Y / L[2])
    eta = lambda T: np.clip(1e20 * np.exp(120e3 / (8.31446 * T) - 120e3 / (8.31446 * 1623)), 1e17, 1e23)
    Tn = field(grid)
    es = eta(Tn); en = eta(field(mid)); rho = 3300 / (3.5e-5 * (Tn - 1623) + 1)
    ctx = P3.Context3(n, grid, device)
    carrier = None
    if world > 1:                         # the 2-D carrier context brings the transport (pylamp_amd/parallel.py), see pl3_set_comm
        from pylamp_amd._context import Context
        ncol = 16 * world + 1
        carrier = Context([17, ncol], [np.linspace(0, 1, 17), np.linspace(0, 1, ncol)], device=device)
        ctx.attach_comm(carrier, *P)
    A, _ = P3.makeStokesMatrix(n, grid, es, en, rho, ctx=ctx)
    k = np.full(n, 4.0); cp = np.full(n, 1250.0); H = np.full(n, 0.02e-6 / 3300)
    dt = 0.67 * (L[0] / (n[0] - 1)) ** 2 / np.max(2 * 4.0 / (rho * 1250.0))
    Ah, _ = P3.makeDiffusionMatrix(n, grid, mid, Tn, [k, k, k], cp, rho, H, [0, 1, 1, 0, 1, 1], [273.0, 0, 0, 1623.0, 0, 0], dt, ctx=ctx)
    del es, en, rho, Tn, k, cp, H
    its, hits, res = [], [], []
    # device-resident solves: coefficients are on the GPU, the solutions stay there (pl3_get_solution fetches them afterwards);
    # every solve is a cold start from the hydrostatic state / from zero
    for s_ in range(args.warmup):
        P3.solve(A, resident=True); P3.solve_heat(Ah, resident=True)
    ctx.comm_stats(reset=True)
    barrier()
    t0 = time.perf_counter()
    for s_ in range(args.steps):
        P3.solve(A, resident=True); its.append(A.last_stats["iterations"]); res.append(A.last_stats["rel_residual"])
        conv = A.last_stats["converged"]
        P3.solve_heat(Ah, resident=True); hits.append(Ah.last_stats["iterations"])
    barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        import torch
        t = torch.tensor([el], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    cells = (n[0] - 1) * (n[1] - 1) * (n[2] - 1)
    if rank == 0:
        ms_s = A.apply_bench(20, True); ms_p = A.apply_bench(20, False)
        ach = 80.0 * nb ** 3 / (ms_s * 1e-3) / 1e9          # this rank's block
        out = {"metric": "stokes_heat_3d_cell_updates_per_s", "value": round(cells * args.steps / el, 1), "unit": "cell-updates/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
               "config": {"workload": "3D %dx%dx%d nodes staggered Stokes (4 DOF/node, T-dependent viscosity 1e20..1e23) + implicit heat, one solve of each per step; "
                                      "device-resident (coefficients uploaded before, solutions left on the GPU: no PCIe in the timed region)" % tuple(n),
                          "parallelism": "1 GPU" if world == 1 else "%d x %d x %d blocks of %d^3 nodes, one-node halos exchanged axis by axis, host all-reduced dot products" % (P + (nb,)),
                          "parity": "unpinned (the reference is 2-D only)"},
               "stokes_iterations": its, "stokes_converged": conv, "stokes_rel_residual": [float("%.3g" % r) for r in res], "heat_iterations": hits,
               "roofline": {"kernel": "k3_apply<true> (3-D row-scaled Stokes stencil)", "bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS,
                            "unit": "GB/s", "frac": round(ach / HBM_PEAK_GBS, 4), "traffic": None, "algorithmic_bytes_per_launch": 80.0 * nb ** 3,
                            "avg_launch_ms": round(ms_s, 5), "plain_operator_ms": round(ms_p, 5)}}
        if world > 1:
            hc, rc = ctx.comm_stats()
            out["comm_calls_per_step"] = {"halo_exchanges": round(hc / float(args.steps), 1), "host_allreduces": round(rc / float(args.steps), 1)}
            out["scaling_note"] = "no multi-GPU node was available to the build: N > 1 has only been rehearsed with virtual ranks / gloo on one GPU"
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher (no WORLD_SIZE in the environment): start the N ranks as child processes, one
    per GPU, with the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT) and wait for them.
    The parent never touches HIP (torch.cuda.device_count() does not initialise the GPU on this image); nothing is exec'ed from a
    process that has.  Fewer visible GPUs than ranks: the ranks share the GPUs under the gloo backend (a rehearsal, said so on the
    bench line).  A failed rank ends the job: the others are terminated (by pid) and the exit code is non-zero."""
    import socket
    import subprocess
    try:
        import torch
        ndev = torch.cuda.device_count()
    except Exception:
        ndev = 0
    env = dict(os.environ)
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in env:
        s = socket.socket(); s.bind(("127.0.0.1", 0)); env["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
    env["WORLD_SIZE"] = str(n); env["LOCAL_WORLD_SIZE"] = str(n)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    if ndev < n and "PYLAMP_DIST_BACKEND" not in env:
        env["PYLAMP_DIST_BACKEND"] = "gloo"
    procs = []
    for r in range(n):
        e = dict(env); e["RANK"] = str(r); e["LOCAL_RANK"] = str(r)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            try:
                code = p.wait(timeout=0.5)
            except subprocess.TimeoutExpired:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:
                    q.terminate()
    if rc:
        sys.exit(rc if 0 < rc < 256 else 1)


def init_dist(args):
    """torch.distributed set-up of one rank (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment: torchrun, or spawn_ranks).
    Returns rank, world, local_rank, the dist module (None on one rank), the device of the timing reductions, a barrier and the GPU."""
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("PYLAMP_DIST_BACKEND", "nccl")     # "gloo": several ranks sharing one GPU (tests)
        if backend == "nccl":
            # Transport of the halo exchanges / all-reduces inside the library: "native" = direct RCCL calls on the solver's
            # stream (self-tested at start-up against a 60 s deadline watched on an event, collective fall-back), "torch" =
            # the callback table into torch.distributed (RCCL as well, but host-synchronous per call).  The native path has
            # never run on more than one GPU (no multi-GPU node was available to the build), so it stays OPT-IN until it has
            # passed once on a real node: --transport native, or PYLAMP_RCCL=1.
            if args.transport == "native":
                os.environ["PYLAMP_RCCL"] = "1"
            torch.cuda.set_device(local_rank)
            import datetime
            # (a hung collective ends as an error after 3 minutes instead of the default 10: the driver's clock is running)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank), timeout=datetime.timedelta(seconds=180))
        else:
            dist.init_process_group(backend=backend)
    red_dev = "cuda" if (dist is not None and dist.get_backend() == "nccl") else "cpu"

    def barrier():
        if dist is not None:
            import torch
            dist.barrier()
            if red_dev == "cuda":
                torch.cuda.synchronize()

    device = int(os.environ["PYLAMP_DEVICE"]) if os.environ.get("PYLAMP_DEVICE") else local_rank
    if world > 1 and not os.environ.get("PYLAMP_DEVICE") and os.environ.get("PYLAMP_DIST_BACKEND", "nccl") != "nccl":
        import torch                                                # rehearsal: several gloo ranks may share the GPUs of the box
        device = local_rank % max(torch.cuda.device_count(), 1)
    return rank, world, local_rank, dist, red_dev, barrier, device


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="2d", choices=["2d", "3d257"], help="3d257: BASELINE config 5 on one GPU (3-D Stokes + heat)")
    ap.add_argument("--grid3", dest="n3", type=int, default=257, help="nodes per side of the 3-D configuration")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--grid", dest="n", type=int, default=2049, help="nodes per side (default 2049: BASELINE config 3)")
    ap.add_argument("--tracdens", type=int, default=16)
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N > 1: strong = the same grid split over N GPUs (BASELINE's metric); weak = one --grid slab per GPU")
    ap.add_argument("--model", choices=["mantle", "block"], default="mantle",
                    help="mantle (default, the headline): T-dependent convection with heat; block: the 10^3 falling block of config 1 "
                         "(heat off) at the same size -- the slower-converging Stokes problem, a secondary figure")
    ap.add_argument("--transport", choices=["torch", "native"], default="torch",
                    help="N > 1: torch.distributed callbacks (default) or direct RCCL calls on the solver stream (opt-in)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--apply-reps", type=int, default=50)
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args.gpus)
    if args.config == "3d257":
        return bench_3d(args)

    rank, world, local_rank, dist, red_dev, barrier, device = init_dist(args)
    sim = build_sim(args.n, args.tracdens, 20260103, device, rank, world, args.scaling, args.model)
    ctx = sim.ctx
    reports = []
    for _ in range(args.warmup):
        reports.append(sim.step())
    ctx.check(ctx.lib.pl_sync(ctx.h))
    cc = (C.c_int64 * 4)()
    ctx.check(ctx.lib.pl_comm_stats(ctx.h, cc, 1))          # counts of the timed region only
    ct = (C.c_double * 4)()
    ctx.check(ctx.lib.pl_comm_times(ctx.h, ct, 1))
    barrier()
    t0 = time.perf_counter()
    timed = []
    for _ in range(args.steps):
        timed.append(sim.step())
    ctx.check(ctx.lib.pl_sync(ctx.h))
    barrier()
    elapsed = time.perf_counter() - t0
    ctx.check(ctx.lib.pl_comm_stats(ctx.h, cc, 0))
    comm_calls = [int(v) for v in cc]
    ctx.check(ctx.lib.pl_comm_times(ctx.h, ct, 0))
    comm_ms = [float(v) for v in ct]
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    cells = (sim.nx[0] - 1) * (sim.nx[1] - 1)
    ms_per_step = 1e3 * elapsed / args.steps
    # N > 1: ONE global problem of Pz x Px blocks (halo exchange + all-reduce over RCCL); value = global cells * steps / time
    # (strong scaling: the global problem is fixed; weak: it grows with N).
    value = cells * args.steps / elapsed
    ntrac_global = sim.ntrac
    if dist is not None:
        import torch
        t = torch.tensor([float(sim.ntrac)], dtype=torch.float64, device=red_dev)
        dist.all_reduce(t)
        ntrac_global = int(t.item())
    if rank == 0:
        roof = kernel_roofline(sim, args.apply_reps)
        stage_keys = ["ms_sort", "ms_props", "ms_scatter", "ms_stokes", "ms_heat", "ms_gather", "ms_advect", "ms_total"]
        stages = {k: round(float(np.mean([r[k] for r in timed])), 3) for k in stage_keys}
        # marker stages against the HBM roof: algorithmic bytes of one step (this rank's tracers and nodes) over the stage's
        # wall time (host-timed with a stream sync each side, so launch gaps are inside -- these stages are 1.7-6 ms long).
        #   scatter  fused kernel (regular grid): positions 16 B + 7 fields 56 B (+ 4 B slot in the epoch layout) read ONCE per marker,
        #            13 accumulator planes written and read by the finalisation + 9 result planes written per node (280 B / node);
        #            the one-set-per-pass kernels (rectilinear grids, several target sets re-reading the positions): 136 B / marker
        #   rk4      positions read and written, velocities written: 48 B / marker (the 32 velocity gathers per marker hit
        #            a cache-resident window)
        #   sort     epoch layout (one rank): key 4 B + destination 4 B + positions, temperature and slot read and written 56 B
        #            = 64 B / marker, + the re-layout of the 10 constant columns and the index every PYLAMP_EPOCH sorts
        #            (172 B / marker / epoch length); classic (several ranks): key pass 16 B + placement 12 B + 17 columns
        #            and the index moved 2 x 144 B = 316 B / marker
        _, ni_, _, nj_, _, _ = sim.ctx.local_block()
        nt, nn = float(sim.ntrac), float(ni_ * nj_)

        def stage_roof(ms, nbytes, what):
            gbs = nbytes / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "algorithmic_bytes_per_step": nbytes, "ms": ms, "bytes": what}
        epoch_age, _lazy = sim.layout()
        epoch_len = int(os.environ.get("PYLAMP_EPOCH", "64"))
        fused_scatter = os.environ.get("PYLAMP_SCATTER", "1") != "0"
        if fused_scatter:
            per_marker = 72.0 + (4.0 if epoch_age > 0 else 0.0)
            roof["scatter"] = stage_roof(stages["ms_scatter"], per_marker * nt + 280.0 * nn,
                                         "%d B/marker + 280 B/node (fused kernel; the reference's four trac2grid calls read 136 B/marker)" % per_marker)
        else:
            roof["scatter"] = stage_roof(stages["ms_scatter"], 136.0 * nt + 208.0 * nn, "136 B/marker + 208 B/node")
        roof["rk4"] = stage_roof(stages["ms_advect"], 48.0 * nt, "48 B/marker")
        if epoch_age > 0 and epoch_len > 0:
            per_marker = 64.0 + 172.0 / epoch_len
            roof["sort"] = stage_roof(stages["ms_sort"], per_marker * nt, "%.1f B/marker (epoch layout: 64 + 172 / %d)" % (per_marker, epoch_len))
        else:
            roof["sort"] = stage_roof(stages["ms_sort"], 316.0 * nt, "316 B/marker")
        out = {
            "metric": "stokes_heat_mic_cell_updates_per_s", "value": round(value, 1), "unit": "cell-updates/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3),
            "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": ("2D %dx%d nodes (%dx%d cells), %d markers/node (%d tracers), " + ("T-dependent mantle "
                                   "model, heat + subgrid diffusion on" if args.model == "mantle" else "falling block (density +50 kg/m3, viscosity x 1e3; "
                                   "pylamp2.py:172-183), heat off: SECONDARY mode, not the headline") + ", all free-slip, full time step") %
                                   (sim.nx[0], sim.nx[1], sim.nx[0] - 1, sim.nx[1] - 1, args.tracdens, ntrac_global),
                       "parallelism": "1 GPU" if world == 1 else "%d x %d blocks of the node grid (8-neighbour halo exchange with corners + all-reduce), transport: %s" %
                                      (sim.ctx.local_block()[4], sim.ctx.local_block()[5], "direct RCCL on the solver stream" if (sim.ctx.comm is not None and sim.ctx.comm.native)
                                       else "torch.distributed (%s)" % (dist.get_backend() if dist is not None else "-")),
                       "stokes_rtol": sim.opt.stokes_rtol, "heat_rtol": sim.opt.heat_rtol,
                       # what `dtype: f64` does not say: storage of the preconditioner's finest level and the tracer layout
                       "precond_storage": ("level-0 f / v1 / r of the multigrid preconditioner stored in FP32 (FP64 arithmetic; operator, Krylov "
                                           "vectors and every residual FP64)" if os.environ.get("PYLAMP_L0_MIXED", "1") != "0" else "FP64 throughout"),
                       "tracer_layout": "epoch (PYLAMP_EPOCH=%s), age %d" % (os.environ.get("PYLAMP_EPOCH", "64"), epoch_age) if epoch_age > 0 else "classic",
                       "stokes_stop": "true relative residual <= stokes_rtol AND estimated relative velocity error <= %s "
                                      "(pl_solve_stats.error_estimate; the drop-in's promise against the reference's direct solve is 1e-6)"
                                      % os.environ.get("PYLAMP_STOKES_ETOL", "3e-8")},
            "time_steps_per_s": round(args.steps / elapsed, 4),
            "stage_ms": stages,
            "stokes_iterations": [r["stokes"]["iterations"] for r in timed],
            "stokes_precond_applies": [r["stokes"]["precond_applies"] for r in timed],
            "stokes_operator_applies": [r["stokes"]["operator_applies"] for r in timed],
            "stokes_rel_residual": [float("%.3g" % r["stokes"]["rel_residual"]) for r in timed],
            "stokes_converged": [r["stokes"]["converged"] for r in timed],
            "stokes_error_estimate": [float("%.3g" % r["stokes"]["error_estimate"]) for r in timed],
            "heat_iterations": [r["heat"]["iterations"] for r in timed],
            "tracers_injected": [int(r["ninjected"]) for r in timed],
            "roofline": roof,
        }
        if world > 1:       # communication calls of rank 0 per timed step (halo exchanges, all-gathers, device / host all-reduces)
            out["comm_calls_per_step"] = [round(c / float(args.steps), 1) for c in comm_calls]
            # rank 0's time inside them per step: [halo exchanges incl. pack / unpack, all-gathers, device all-reduces] between HIP
            # events on the solver stream, [host all-reduces] host wall time
            out["comm_ms_per_step"] = [round(v / float(args.steps), 3) for v in comm_ms]
            out["transport"] = ("native RCCL (ncclSend/Recv groups, ncclAllReduce on the solver stream)" if (sim.ctx.comm is not None and sim.ctx.comm.native)
                                else "torch.distributed callbacks (%s), host-synchronous" % (dist.get_backend() if dist is not None else "-"))
            out["scaling_note"] = "no multi-GPU node was available to the build: N > 1 has only been rehearsed with virtual ranks / gloo on one GPU"
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
