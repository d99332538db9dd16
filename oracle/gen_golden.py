#!/usr/bin/env python3
"""Generate tests/golden/*.npz by RUNNING the reference (PyLamp) itself.

Run once in the build container, where /root/reference exists:

    python oracle/gen_golden.py

The reference's files are imported from where they lie and are never copied; only the
inputs/outputs (data) are stored.  Two run-time shims are applied *in this process only*:

  * pylamp_trac.np is replaced by a proxy whose add.at turns the reference's list index
    `[i, j]` into a tuple (numpy >= 1.23 no longer accepts a list as a multi-index;
    pylamp_trac.py:257-298).  Everything else goes to numpy unchanged.
  * for the trajectory fixtures the stock driver script text is exec'd in memory with a
    stub single-rank mpi4py, time.clock=perf_counter and its configuration variables
    (grid size, model, BCs, step count) replaced; the file on disk is untouched.

All randomness is seeded, so the fixtures are reproducible.
"""
import os
import sys
import types
import time
import tempfile

import numpy as np
import scipy.sparse as sp
import scipy.sparse.linalg as spla

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")
sys.path.insert(0, REF)

import pylamp_stokes as RS  # noqa: E402
import pylamp_diff as RD    # noqa: E402
import pylamp_trac as RT    # noqa: E402
from pylamp_const import *  # noqa: E402,F401,F403


class _AddProxy:
    def at(self, a, idx, b):
        if isinstance(idx, list):
            idx = tuple(idx)
        return np.add.at(a, idx, b)

    def __call__(self, *a, **k):
        return np.add(*a, **k)


class _NpProxy:
    add = _AddProxy()

    def __getattr__(self, name):
        return getattr(np, name)


RT.np = _NpProxy()


def save(name, **kw):
    os.makedirs(OUT, exist_ok=True)
    p = os.path.join(OUT, name + ".npz")
    np.savez_compressed(p, **kw)
    print("wrote", name, "%.1f KB" % (os.path.getsize(p) / 1024))


def nonuniform(n, L, rng):
    w = rng.uniform(0.6, 1.4, n - 1)
    g = np.concatenate([[0.0], np.cumsum(w)])
    return g * (L / g[-1])


def csr_parts(A):
    A = sp.csr_matrix(A)
    A.sort_indices()
    return dict(data=A.data, indices=A.indices.astype(np.int32), indptr=A.indptr.astype(np.int32))


# ------------------------------------------------------------------------------------ F1/F3
def gen_stokes_operator():
    cases = [
        ("a", [9, 7], True, [1, 1, 1, 1], 3),
        ("b", [12, 10], False, [0, 1, 0, 1], 6),
        ("c", [8, 13], False, [0, 1, 1, 1], 2),
        ("d", [41, 41], True, [1, 1, 1, 1], 4),
        ("e", [21, 37], False, [1, 1, 0, 1], 5),
    ]
    for tag, nx, uniform, bc, decades in cases:
        rng = np.random.default_rng(100 + ord(tag))
        L = [660e3, 500e3]
        grid = [np.linspace(0, L[d], nx[d]) if uniform else nonuniform(nx[d], L[d], rng)
                for d in range(2)]
        etas = 1e19 * 10 ** rng.uniform(0, decades, nx)
        etan = 1e19 * 10 ** rng.uniform(0, decades, nx)
        rho = 3300 + rng.uniform(-50, 50, nx)
        A, rhs = RS.makeStokesMatrix(nx, grid, etas, etan, rho, bc)
        A = sp.csr_matrix(A)
        xs = rng.standard_normal((3, A.shape[0]))
        ys = np.stack([A @ x for x in xs])
        save("stokes_op_" + tag, nx=np.array(nx), gz=grid[0], gx=grid[1], etas=etas, etan=etan,
             rho=rho, bc=np.array(bc), rhs=rhs, xs=xs, ys=ys, **csr_parts(A))
        if tag in ("b", "a"):
            tstep = 3.0e11
            A2, rhs2 = RS.makeStokesMatrix(nx, grid, etas, etan, rho, bc, surfstab=True,
                                           tstep=tstep, surfstab_theta=0.5)
            A2 = sp.csr_matrix(A2)
            ys2 = np.stack([A2 @ x for x in xs])
            save("stokes_surfstab_" + tag, nx=np.array(nx), gz=grid[0], gx=grid[1], etas=etas,
                 etan=etan, rho=rho, bc=np.array(bc), rhs=rhs2, xs=xs, ys=ys2, tstep=tstep,
                 theta=0.5, **csr_parts(A2))


# ------------------------------------------------------------------------------------ F2
def falling_block_tracers(nx, L, dens, rng):
    n = int(np.prod(nx)) * dens
    tr_x = rng.random((n, 2)) * np.array(L)
    tr_f = np.zeros((n, NFTRAC))
    tr_f[:, TR_RH0] = 3300; tr_f[:, TR_MAT] = 1; tr_f[:, TR_ET0] = 1e19
    idxb = (tr_x[:, 0] > 200e3) & (tr_x[:, 0] < 300e3) & (tr_x[:, 1] > 280e3) & (tr_x[:, 1] < 380e3)
    tr_f[idxb, TR_RH0] = 3350; tr_f[idxb, TR_MAT] = 2; tr_f[idxb, TR_ET0] = 1e22
    tr_f[:, TR_RHO] = tr_f[:, TR_RH0]; tr_f[:, TR_ETA] = tr_f[:, TR_ET0]
    return tr_x, tr_f


def ref_grids(nx, L):
    grid = [np.linspace(0, L[i], nx[i]) for i in range(2)]
    mesh = np.meshgrid(*grid, indexing='ij')
    gridmp = [(grid[i][1:nx[i]] + grid[i][0:(nx[i] - 1)]) / 2 for i in range(2)]
    for i in range(2):
        gridmp[i] = np.append(gridmp[i], gridmp[i][-1] + (gridmp[i][-1] - gridmp[i][-2]))
    meshmp = np.meshgrid(*gridmp, indexing='ij')
    return grid, mesh, gridmp, meshmp


def gen_stokes_solve():
    # (1) falling block 41x41, 16 markers/node
    nx = [41, 41]; L = [660e3, 660e3]
    rng = np.random.default_rng(2001)
    grid, mesh, gridmp, meshmp = ref_grids(nx, L)
    tr_x, tr_f = falling_block_tracers(nx, L, 16, rng)
    f_rho = np.zeros(nx); f_etas = np.zeros(nx); f_etan = np.zeros(nx)
    RT.trac2grid(tr_x, tr_f[:, [TR_RHO, TR_ETA]], mesh, grid, [f_rho, f_etas], nx,
                 avgscheme=[RT.INTERP_AVG_ARITHW, RT.INTERP_AVG_GEOMW])
    RT.trac2grid(tr_x, tr_f[:, [TR_ETA]], meshmp, gridmp, [f_etan], nx,
                 avgscheme=[RT.INTERP_AVG_GEOMW])
    bc = [1, 1, 1, 1]
    A, rhs = RS.makeStokesMatrix(nx, grid, f_etas, f_etan, f_rho, bc)
    x = spla.spsolve(sp.csc_matrix(A), rhs)
    save("stokes_solve_block41", nx=np.array(nx), gz=grid[0], gx=grid[1], etas=f_etas,
         etan=f_etan, rho=f_rho, bc=np.array(bc), rhs=rhs, x=x)
    # (2) smooth T-dependent viscosity, non-square, NOSLIP top
    nx = [33, 49]; L = [660e3, 990e3]
    grid, mesh, gridmp, meshmp = ref_grids(nx, L)
    Z, X = mesh
    T = 273 + 1350 * Z / L[0] + 40 * np.sin(3 * np.pi * X / L[1]) * np.sin(np.pi * Z / L[0])
    Zc, Xc = meshmp
    Tc = 273 + 1350 * Zc / L[0] + 40 * np.sin(3 * np.pi * Xc / L[1]) * np.sin(np.pi * Zc / L[0])
    Tc = np.clip(Tc, 273, 1700)

    def eta(T):
        return np.clip(1e20 * np.exp(120e3 / (GASR * T) - 120e3 / (GASR * 1623)), 1e17, 1e23)
    f_etas = eta(T); f_etan = eta(Tc)
    f_rho = 3300 / (3.5e-5 * (T - 1623) + 1)
    bc = [0, 1, 1, 1]
    A, rhs = RS.makeStokesMatrix(nx, grid, f_etas, f_etan, f_rho, bc)
    x = spla.spsolve(sp.csc_matrix(A), rhs)
    save("stokes_solve_tdep33x49", nx=np.array(nx), gz=grid[0], gx=grid[1], etas=f_etas,
         etan=f_etan, rho=f_rho, bc=np.array(bc), rhs=rhs, x=x)


# ------------------------------------------------------------------------------------ F4
def gen_heat():
    combos = [[0, 1, 0, 1], [0, 0, 0, 0], [1, 0, 0, 1], [0, 1, 1, 0]]
    for tag, nx, uniform in (("a", [9, 7], True), ("b", [41, 41], False), ("c", [17, 30], True)):
        rng = np.random.default_rng(300 + ord(tag))
        L = [660e3, 400e3]
        grid = [np.linspace(0, L[d], nx[d]) if uniform else nonuniform(nx[d], L[d], rng)
                for d in range(2)]
        gridmp = [(g[1:] + g[:-1]) / 2 for g in grid]
        gridmp = [np.append(m, m[-1] + (m[-1] - m[-2])) for m in gridmp]
        kz = rng.uniform(2, 5, nx); kx = rng.uniform(2, 5, nx)
        Cp = rng.uniform(1000, 1250, nx); rho = rng.uniform(2900, 3400, nx)
        H = rng.uniform(0, 1e-9, nx) * 3300
        T = rng.uniform(273, 1623, nx)
        hmin = min(np.min(np.diff(grid[0])), np.min(np.diff(grid[1])))
        tstep = 0.67 * hmin ** 2 / np.max(2 * kz / (rho * Cp)) * (5.0 if tag == "b" else 1.0)
        for ci, bc in enumerate(combos):
            bcv = [273.0, 0.0, 1623.0, 0.0]
            for w in range(4):
                if bc[w] == 1 and w in (0, 2):
                    bcv[w] = 0.03 if w == 0 else -0.02
                if bc[w] == 0 and w in (1, 3):
                    bcv[w] = 800.0 + 100 * w
            A, rhs = RD.makeDiffusionMatrix(nx, grid, gridmp, T, [kz, kx], Cp, rho, H, bc, bcv, tstep)
            A = sp.csr_matrix(A)
            sol = spla.spsolve(sp.csc_matrix(A), rhs)
            xs = rng.standard_normal((2, A.shape[0]))
            ys = np.stack([A @ x for x in xs])
            save("heat_%s%d" % (tag, ci), nx=np.array(nx), gz=grid[0], gx=grid[1], gmz=gridmp[0],
                 gmx=gridmp[1], T=T, kz=kz, kx=kx, Cp=Cp, rho=rho, H=H, bc=np.array(bc),
                 bcvalue=np.array(bcv), tstep=tstep, rhs=rhs, sol=sol, xs=xs, ys=ys, **csr_parts(A))


# ------------------------------------------------------------------------------------ F5
def gen_trac2grid():
    nx = [11, 14]; L = [5.0, 9.0]
    grid, mesh, gridmp, meshmp = ref_grids(nx, L)
    targets = {
        "nodes": (mesh, grid),
        "centres": (meshmp, gridmp),
        "zmid": ([meshmp[0], mesh[1]], [gridmp[0], grid[1]]),
        "xmid": ([mesh[0], meshmp[1]], [grid[0], gridmp[1]]),
    }
    rng = np.random.default_rng(500)
    dense = rng.random((int(np.prod(nx)) * 12, 2)) * np.array(L)
    sparse = rng.random((60, 2)) * np.array(L)              # leaves empty nodes -> NaN
    outside = rng.random((900, 2)) * np.array([L[0] * 1.25, L[1] * 1.3]) - np.array([0.4, 1.1])
    schemes = [1, 2, 5, 6]
    res = {}
    for cname, tx in (("dense", dense), ("sparse", sparse), ("outside", outside)):
        f = np.stack([rng.uniform(1, 10, tx.shape[0]), 10 ** rng.uniform(17, 23, tx.shape[0]),
                      rng.uniform(-5, 5, tx.shape[0]) ** 2 + 0.1, rng.uniform(0.5, 2, tx.shape[0])], axis=1)
        res["%s_tr_x" % cname] = tx
        res["%s_tr_f" % cname] = f
        for tname, (m, g) in targets.items():
            gf = [np.zeros(nx) for _ in schemes]
            with np.errstate(all='ignore'):
                RT.trac2grid(tx, f.copy(), m, g, gf, nx, avgscheme=list(schemes))
            res["%s_%s" % (cname, tname)] = np.stack(gf)
    # zero-valued tracer under GEOM (pylamp_trac.py:301): node log-sum -inf reset to 0 -> 1
    tx = dense[:400].copy()
    f = rng.uniform(1, 10, (400, 2)); f[::37, :] = 0.0
    gf = [np.zeros(nx), np.zeros(nx)]
    with np.errstate(all='ignore'):
        RT.trac2grid(tx, f.copy(), mesh, grid, gf, nx, avgscheme=[6, 2])
    res["zero_tr_x"] = tx; res["zero_tr_f"] = f; res["zero_nodes"] = np.stack(gf)
    save("trac2grid", nx=np.array(nx), L=np.array(L), schemes=np.array(schemes), **res)


# ------------------------------------------------------------------------------------ F6/F7
def gen_grid2trac_rk():
    nx = [13, 10]; L = [4.0, 7.0]
    grid, mesh, gridmp, meshmp = ref_grids(nx, L)
    rng = np.random.default_rng(600)
    F = [rng.standard_normal(nx), rng.standard_normal(nx)]
    inside = rng.random((3000, 2)) * np.array(L) * 0.999
    mixed = rng.random((800, 2)) * np.array(L) * 1.4 - 0.2 * np.array(L)
    # a tracer whose cell index equals n-1 (coordinate in [L, L+h)) is accepted by the
    # reference's range test (pylamp_trac.py:52) and then indexes out of bounds: exclude.
    ce = [np.floor((nx[d] - 1) * mixed[:, d] / L[d]) for d in range(2)]
    mixed = mixed[(ce[0] != nx[0] - 1) & (ce[1] != nx[1] - 1)]
    res =dict(nx=np.array(nx), L=np.array(L), F=np.stack(F), inside=inside, mixed=mixed)
    for mname, meth in (("linear", RT.INTERP_METHOD_LINEAR), ("nearest", RT.INTERP_METHOD_NEAREST),
                        ("veldiv", RT.INTERP_METHOD_VELDIV)):
        o = np.zeros((inside.shape[0], 2))
        RT.grid2trac(inside, o, grid, F, nx, method=meth)
        res["inside_" + mname] = o
        for dname, dv in (("nan", np.nan), ("zero", 0.0)):
            o = np.zeros((mixed.shape[0], 2))
            # tracers exactly in the last accepted "cell" index n-1 would index out of bounds
            # in the reference (pylamp_trac.py:52 accepts ielem == nz-1); keep them out.
            RT.grid2trac(mixed, o, grid, F, nx, defval=dv, method=meth)
            res["mixed_%s_%s" % (mname, dname)] = o
    save("grid2trac", **res)

    # RK4 on the padded cell-centre grid, analytic solenoidal field + random field
    nx = [17, 21]; L = [660e3, 660e3]
    grid, mesh, gridmp, meshmp = ref_grids(nx, L)
    gz = np.insert(gridmp[0], 0, gridmp[0][0] - (gridmp[0][1] - gridmp[0][0]))
    gx = np.insert(gridmp[1], 0, gridmp[1][0] - (gridmp[1][1] - gridmp[1][0]))
    ZZ, XX = np.meshgrid(gz, gx, indexing='ij')
    U = 3e-9
    Vz = U * np.sin(np.pi * ZZ / L[0]) * np.cos(np.pi * XX / L[1])
    Vx = -U * np.cos(np.pi * ZZ / L[0]) * np.sin(np.pi * XX / L[1])
    tr = rng.random((5000, 2)) * np.array(L)
    tstep = 0.67 * (L[0] / (nx[0] - 1)) / U
    v1, x1 = RT.RK(tr, [gz, gx], [Vz, Vx], nx, tstep)
    # random, everywhere-negative field and a big step: some stages leave the padded grid on
    # the LOW side (velocity defval 0 there).  Leaving on the high side by less than one cell
    # indexes out of bounds in the reference (pylamp_trac.py:52), so it cannot be a fixture.
    Vz2 = -U * np.abs(rng.standard_normal(Vz.shape)); Vx2 = -U * np.abs(rng.standard_normal(Vx.shape))
    v2, x2 = RT.RK(tr, [gz, gx], [Vz2, Vx2], nx, 4 * tstep)
    save("rk4", nx=np.array(nx), L=np.array(L), gz=gz, gx=gx, Vz=Vz, Vx=Vx, tr=tr, tstep=tstep,
         v1=v1, x1=x1, Vz2=Vz2, Vx2=Vx2, v2=v2, x2=x2)


# ------------------------------------------------------------------------------------ F8
def run_driver(tag, repl, nsteps, seed, capture_post=False, compact=None):
    """exec the stock driver text with in-memory substitutions; collect its snapshots.
    capture_post: also store the tracer state at the END of every loop pass, i.e. after deletion and injection
    (pylamp2.py:574-633) -- the stock snapshot holds the state BEFORE injection (prev_tr_x / prev_tr_f).
    compact: function that reduces the collected arrays before they are stored (the stock configuration has 370 845 tracers)."""
    src = open(os.path.join(REF, "pylamp2.py")).read()
    if capture_post:
        hook = "        if IPROC == 0 and output_numpy and ((output_stride > 0"
        assert hook in src
        src = src.replace(hook, "        np.savez(output_outdir + '/post.{:06d}.npz'.format(it), tr_x=tr_x, tr_f=tr_f)\n" + hook)
    for a, b in repl:
        assert a in src, a
        src = src.replace(a, b)
    src = src.replace("max_it = 1e10", "max_it = %d" % nsteps)
    # capture the initial tracer state right before the time loop
    src = src.replace("    it = 0\n    totaltime = 0\n",
                      "    np.savez(output_outdir + '/init.npz', tr_x=tr_x, tr_f=tr_f)\n"
                      "    it = 0\n    totaltime = 0\n")
    mpi = types.ModuleType("mpi4py")

    class _Comm:
        def Get_rank(self): return 0
        def Get_size(self): return 1
        def Bcast(self, buf, root=0): pass
        def Allreduce(self, s, r, op=None): r[0][...] = s[0]
    MPI = types.ModuleType("mpi4py.MPI")
    MPI.COMM_WORLD = _Comm(); MPI.DOUBLE = None; MPI.SUM = None
    mpi.MPI = MPI
    sys.modules["mpi4py"] = mpi; sys.modules["mpi4py.MPI"] = MPI
    time.clock = time.perf_counter
    cwd = os.getcwd()
    with tempfile.TemporaryDirectory() as td:
        os.makedirs(os.path.join(td, "out"))
        os.chdir(td)
        try:
            np.random.seed(seed)
            g = {"__name__": "__main__"}
            try:
                exec(compile(src, "pylamp2_inmem", "exec"), g)
            except SystemExit:
                pass
        finally:
            os.chdir(cwd)
        init = np.load(os.path.join(td, "out", "init.npz"))
        res = dict(init_tr_x=init["tr_x"], init_tr_f=init["tr_f"], nsteps=nsteps)
        for it in range(1, nsteps + 1):
            gd = np.load(os.path.join(td, "out", "griddata.%06d.npz" % it))
            tc = np.load(os.path.join(td, "out", "tracs.%06d.npz" % it))
            for k in ("velz", "velx", "pres", "rho", "temp", "time"):
                res["s%d_%s" % (it, k)] = gd[k]
            res["s%d_tr_x" % it] = tc["tr_x"]
            res["s%d_tr_T" % it] = tc["tr_f"][:, TR_TMP]
            res["s%d_tr_v" % it] = tc["tr_v"]
            if capture_post:
                res["s%d_tr_id" % it] = tc["tr_f"][:, TR__ID]
                po = np.load(os.path.join(td, "out", "post.%06d.npz" % it))
                res["p%d_tr_x" % it] = po["tr_x"]; res["p%d_tr_f" % it] = po["tr_f"]
            if it == 1:
                res["gz"] = gd["gridz"]; res["gx"] = gd["gridx"]
    if compact:
        res = compact(res)
    save("traj_" + tag, **res)


def gen_trajectories():
    bcset = ("    bcheat = [[]] * 4\n",
             "    bcheat = [[]] * 4\n    bcstokes = [1, 1, 1, 1]\n")
    # config 1: falling block 41x41, 4 markers/node, heat off (model 2), no injection
    run_driver("block41", [
        ("nx    =   [200+1,40+1]", "nx    =   [41,41]"),
        ("L     =   [1, 0.2] ", "L     =   [660e3, 660e3] "),
        ("tracdens = 45 ", "tracdens = 4 "),
        ("tracdens_min = 25 ", "tracdens_min = 0 "),
        ("choose_model = 5", "choose_model = 2"),
        bcset], 5, 11)
    # model 1 (T-dependent rho/eta, heat + subgrid diffusion), non-square grid
    run_driver("mantle33x41", [
        ("nx    =   [200+1,40+1]", "nx    =   [33,41]"),
        ("L     =   [1, 0.2] ", "L     =   [660e3, 820e3] "),
        ("tracdens = 45 ", "tracdens = 6 "),
        ("tracdens_min = 25 ", "tracdens_min = 0 "),
        ("choose_model = 5", "choose_model = 1"),
        bcset], 3, 12)


def gen_trajectories_census():
    """Census + injection (pylamp2.py:588-633) and deletion (pylamp2.py:574-581) captured from the reference."""
    bcset = ("    bcheat = [[]] * 4\n",
             "    bcheat = [[]] * 4\n    bcstokes = [1, 1, 1, 1]\n")
    # falling block 41x41, 12 markers/node on average, cells below 9 are refilled to 12 at the end of every step.
    # (An EMPTY cell would be refilled with NaN-valued tracers -- 0/0 at pylamp2.py:629 -- and the next step's
    # matrix is singular: seen with 6 markers/node.  At 12 no cell of these runs is empty.)
    run_driver("inject41", [
        ("nx    =   [200+1,40+1]", "nx    =   [41,41]"),
        ("L     =   [1, 0.2] ", "L     =   [660e3, 660e3] "),
        ("tracdens = 45 ", "tracdens = 12 "),
        ("tracdens_min = 25 ", "tracdens_min = 9 "),
        ("choose_model = 5", "choose_model = 2"),
        bcset], 3, 14, capture_post=True)
    # mantle model with heat: the injected tracers' temperature / conductivity etc. are cell means too
    run_driver("inject_mantle25x33", [
        ("nx    =   [200+1,40+1]", "nx    =   [25,33]"),
        ("L     =   [1, 0.2] ", "L     =   [660e3, 880e3] "),
        ("tracdens = 45 ", "tracdens = 14 "),
        ("tracdens_min = 25 ", "tracdens_min = 10 "),
        ("choose_model = 5", "choose_model = 1"),
        bcset], 2, 15, capture_post=True)
    # fence off: tracers outside the domain get TR__ID = -1 and are deleted.  70 tracers start beyond the low
    # walls (the reference's marker code handles those through its grid extension; beyond the HIGH walls it
    # indexes out of bounds, pylamp_trac.py:52)
    run_driver("delete41", [
        ("nx    =   [200+1,40+1]", "nx    =   [41,41]"),
        ("L     =   [1, 0.2] ", "L     =   [660e3, 660e3] "),
        ("tracdens = 45 ", "tracdens = 8 "),
        ("tracdens_min = 25 ", "tracdens_min = 0 "),
        ("tracs_fence_enabled = True", "tracs_fence_enabled = False"),
        ("choose_model = 5", "choose_model = 2"),
        ("    tr_x = np.multiply(tr_x, L)\n", "    tr_x = np.multiply(tr_x, L)\n    tr_x[5:45, IZ] = -3000.0\n    tr_x[100:130, IX] = -2500.0\n"),
        bcset], 2, 16, capture_post=True)


def gen_trajectory_surfstab():
    bcset = ("    bcheat = [[]] * 4\n",
             "    bcheat = [[]] * 4\n    bcstokes = [1, 1, 1, 1]\n")
    # model 3 (rising block under a sticky-air layer), dynamic surfstab time step (pylamp2.py:387-405)
    run_driver("surfstab41", [
        ("nx    =   [200+1,40+1]", "nx    =   [41,41]"),
        ("L     =   [1, 0.2] ", "L     =   [660e3, 660e3] "),
        ("tracdens = 45 ", "tracdens = 6 "),
        ("tracdens_min = 25 ", "tracdens_min = 0 "),
        ("choose_model = 5", "choose_model = 3"),
        ("surface_stabilization = False", "surface_stabilization = True"),
        bcset], 3, 13)


# ------------------------------------------------------------------------------------ F2 / F8: the reference's STOCK configuration
def model5_tracers(nx, L, tracdens, seed):
    """Tracers of the stock run exactly as pylamp2.py:116-125,225-242,255-262 makes them after np.random.seed(seed)."""
    ntrac = int(np.prod(nx)) * tracdens
    tr_x = np.random.RandomState(seed).rand(ntrac, 2) * np.array(L)
    tr_f = np.zeros((ntrac, NFTRAC))
    tr_f[:, TR__ID] = np.arange(0, ntrac)
    tr_f[:, TR_RH0] = 1420; tr_f[:, TR_MAT] = 1; tr_f[:, TR_ET0] = 1e2
    idx = (tr_x[:, 1] - 0.1) ** 2 + (tr_x[:, 0] - 0.2) ** 2 < 0.01 ** 2
    tr_f[idx, TR_RH0] = 1470; tr_f[idx, TR_MAT] = 2; tr_f[idx, TR_ET0] = 1e12
    return tr_x, tr_f


def gen_model5():
    """choose_model = 5 (pylamp2.py:225-242), the configuration the reference ships with: a dense sphere of viscosity 1e12 in a
    fluid of viscosity 1e2 (contrast 1e10), 201 x 41 nodes on a 1 x 0.2 domain, 45 markers per node, heat off."""
    nx = [201, 41]; L = [1.0, 0.2]; seed = 17
    grid, mesh, gridmp, meshmp = ref_grids(nx, L)
    tr_x, tr_f = model5_tracers(nx, L, 45, seed)
    tr_f[:, TR_RHO] = tr_f[:, TR_RH0]; tr_f[:, TR_ETA] = tr_f[:, TR_ET0]
    f_rho = np.zeros(nx); f_etas = np.zeros(nx); f_etan = np.zeros(nx)
    # the advect-only branch of the driver (pylamp2.py:316-319): unweighted geometric mean for etan
    RT.trac2grid(tr_x, tr_f[:, [TR_RHO, TR_ETA]], mesh, grid, [f_rho, f_etas], nx,
                 avgscheme=[RT.INTERP_AVG_ARITHW, RT.INTERP_AVG_GEOMW])
    RT.trac2grid(tr_x, tr_f[:, [TR_ETA]], meshmp, gridmp, [f_etan], nx, avgscheme=[RT.INTERP_AVG_GEOMETRIC])
    bc = [1, 1, 1, 1]
    A, rhs = RS.makeStokesMatrix(nx, grid, f_etas, f_etan, f_rho, bc)
    x = spla.spsolve(sp.csc_matrix(A), rhs)
    save("stokes_solve_sphere201x41", nx=np.array(nx), gz=grid[0], gx=grid[1], etas=f_etas, etan=f_etan, rho=f_rho,
         bc=np.array(bc), rhs=rhs, x=x, seed=seed, tracdens=45)

    # three steps of the UNMODIFIED driver (only max_it and the seed are set).  The stock run holds 370 845 tracers (44 MB per
    # state): stored are the seed (the initial tracers follow from it, model5_tracers above -- checked here against the driver's
    # own), the grid fields of every step, every 41st tracer (position, velocity), sums over all of them, and the rows the
    # driver's census injected (pylamp2.py:588-633) so that a test can continue from the reference's post-injection state.
    def compact(res):
        ix, ifl = model5_tracers(nx, L, 45, seed)
        assert np.array_equal(ix, res["init_tr_x"])
        ref_f = res["init_tr_f"].copy()
        mrk = ref_f[:, TR_MRK].copy(); ref_f[:, TR_MRK] = 0          # passive markers (pylamp2.py:255-262): stored separately
        assert np.array_equal(ifl, ref_f)
        out = dict(seed=seed, tracdens=45, tracdens_min=25, nsteps=res["nsteps"], gz=res["gz"], gx=res["gx"], stride=41,
                   init_mrk_sum=mrk.sum(), init_mrk_sub=mrk[::41])
        for it in range(1, int(res["nsteps"]) + 1):
            p, q = "s%d_" % it, "p%d_" % it
            for k in ("velz", "velx", "pres", "rho", "time"):
                out[p + k] = res[p + k]
            n_old = res[p + "tr_x"].shape[0]
            out[p + "n"] = n_old
            out[p + "tr_x_sub"] = res[p + "tr_x"][::41]; out[p + "tr_v_sub"] = res[p + "tr_v"][::41]
            out[p + "tr_x_sum"] = res[p + "tr_x"].sum(axis=0); out[p + "tr_v_sum"] = np.abs(res[p + "tr_v"]).sum(axis=0)
            out[q + "n"] = res[q + "tr_x"].shape[0]
            out[q + "inj_x"] = res[q + "tr_x"][n_old:]; out[q + "inj_f"] = res[q + "tr_f"][n_old:]
            out[q + "rho_col_sum"] = res[q + "tr_f"][:, TR_RHO].sum(); out[q + "mat_col_sum"] = res[q + "tr_f"][:, TR_MAT].sum()
        return out
    run_driver("model5", [], 3, seed, capture_post=True, compact=compact)


if __name__ == "__main__":
    which = sys.argv[1:] or ["op", "solve", "heat", "t2g", "g2t", "traj", "surfstab", "census", "model5"]
    if "model5" in which: gen_model5()
    if "op" in which: gen_stokes_operator()
    if "solve" in which: gen_stokes_solve()
    if "heat" in which: gen_heat()
    if "t2g" in which: gen_trac2grid()
    if "g2t" in which: gen_grid2trac_rk()
    if "traj" in which: gen_trajectories()
    if "surfstab" in which: gen_trajectory_surfstab()
    if "census" in which: gen_trajectories_census()
