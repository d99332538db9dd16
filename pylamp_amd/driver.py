"""Time-loop driver: the counterpart of the script body of the reference's pylamp2.py.

`Simulation` keeps tracers and every grid field resident on the GPU and advances them with
one C-ABI call per time step (pl_step: pylamp2.py:273-633, fence/deletion and census/injection included).  The
model set-ups mirror the parameter blocks the BASELINE configs use (pylamp2.py:146-183), and
`write_snapshot` emits the reference's griddata/tracs .npz files with the same keys
(pylamp2.py:637-650) so pylamp_post.py can read them.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from ._context import Context
from .pylamp_const import (DIM, IZ, IX, NFTRAC, SECINYR, TR_RHO, TR_ETA, TR_MRK, TR_TMP, TR_HCD, TR_HCP, TR_RH0, TR_ALP,
                           TR_MAT, TR_ACE, TR_ET0, TR_IHT, TR__ID)
from . import pylamp_stokes, pylamp_diff


class Options:
    """Configurable options of pylamp2.py:37-77 honoured by the step."""

    def __init__(self, **kw):
        self.do_heatdiff = True
        self.do_subgrid_heatdiff = True
        self.tdep_rho = True
        self.tdep_eta = True
        self.etamin, self.etamax, self.Tref = 1e17, 1e23, 1623.0
        self.tstep_adv_max = 50e9 * SECINYR; self.tstep_adv_min = 50e-9 * SECINYR
        self.tstep_dif_max = 50e9 * SECINYR; self.tstep_dif_min = 50e-9 * SECINYR
        self.tstep_modifier = 0.67
        self.bcstokes = [pylamp_stokes.BC_TYPE_FREESLIP] * 4
        self.bcheat = [pylamp_diff.BC_TYPE_FIXTEMP, pylamp_diff.BC_TYPE_FIXFLOW,
                       pylamp_diff.BC_TYPE_FIXTEMP, pylamp_diff.BC_TYPE_FIXFLOW]
        self.bcheatvals = [273.0, 0.0, 1623.0, 0.0]
        self.stokes_rtol, self.stokes_maxit = 1e-7, 400       # + velocity-error estimate <= 3e-8 (pl_stokes_solve)
        self.heat_rtol, self.heat_maxit = 1e-12, 2000             # 1e-10 is NOT enough: on the 33x41 reference trajectory it puts 1.9e-6 into the next step's velocity (heat solver)
        self.tracdens, self.tracdens_min, self.inject_seed = 0, 0, 12345     # pylamp2.py:39-40; 0 = no injection
        # False: the reference's ID rule for injected tracers (first new ID of every refilled cell repeats the last one
        # handed out, pylamp2.py:621-622); True: unique IDs
        self.inject_unique_ids = False
        self.tracs_fence_enabled = True                                      # pylamp2.py:41; False: leavers are deleted
        self.surface_stabilization, self.surfstab_theta, self.surfstab_tstep = False, 0.5, -1.0   # pylamp2.py:71-73
        # False: corrected (damping) sign of the stabilisation terms, see pylamp_stokes.makeStokesMatrix
        self.surfstab_strict_reference = True
        for k, v in kw.items():
            if not hasattr(self, k):
                raise Exception("unknown option " + k)
            setattr(self, k, v)


def falling_block_tracers(nx, L, tracdens, rng):
    """Model 2 of the reference (pylamp2.py:172-183)."""
    n = int(np.prod(nx)) * tracdens                     # nodes * density, sic (pylamp2.py:116)
    tr_x = rng.random((n, DIM)) * np.array(L)
    tr_f = np.zeros((n, NFTRAC))
    tr_f[:, TR__ID] = np.arange(n)
    tr_f[:, TR_RH0] = 3300; tr_f[:, TR_MAT] = 1; tr_f[:, TR_ET0] = 1e19
    idxb = (tr_x[:, IZ] > 200e3) & (tr_x[:, IZ] < 300e3) & (tr_x[:, IX] > 280e3) & (tr_x[:, IX] < 380e3)
    tr_f[idxb, TR_RH0] = 3350; tr_f[idxb, TR_MAT] = 2; tr_f[idxb, TR_ET0] = 1e22
    return tr_x, tr_f


def passive_markers(tr_x, L):
    """The chequerboard of passive markers every model of the reference starts with (pylamp2.py:254-262): column TR_MRK."""
    zd = np.linspace(0, L[IZ], 10); xd = np.linspace(0, L[IX], 10)
    m = np.zeros(tr_x.shape[0])
    for i in range(0, 9, 2):
        m[(tr_x[:, IZ] >= zd[i]) & (tr_x[:, IZ] < zd[i + 1])] += 1
    for i in range(1, 9, 2):
        m[(tr_x[:, IZ] >= zd[i]) & (tr_x[:, IZ] < zd[i + 1])] += 2
    for i in range(0, 9, 2):
        m[(tr_x[:, IX] >= xd[i]) & (tr_x[:, IX] < xd[i + 1])] *= -1
    return m


def sphere_tracers(nx, L, tracdens, seed):
    """Model 5 of the reference -- the configuration it ships with (pylamp2.py:37-39,225-242): a dense sphere (1470 against
    1420 kg/m3) of viscosity 1e12 in a fluid of viscosity 1e2, centre (z, x) = (0.2, 0.1), radius 0.01, on the 1 x 0.2 domain
    with 201 x 41 nodes and 45 markers per node; heat off, constant properties.  The positions are the reference's own draw after
    np.random.seed(seed) (pylamp2.py:119): np.random.rand(ntrac, DIM) * L."""
    n = int(np.prod(nx)) * tracdens
    tr_x = np.random.RandomState(seed).rand(n, DIM) * np.array(L)
    tr_f = np.zeros((n, NFTRAC))
    tr_f[:, TR__ID] = np.arange(n)
    tr_f[:, TR_RH0] = 1420; tr_f[:, TR_MAT] = 1; tr_f[:, TR_ET0] = 1e2
    idx = (tr_x[:, IX] - 0.1) ** 2 + (tr_x[:, IZ] - 0.2) ** 2 < 0.01 ** 2
    tr_f[idx, TR_RH0] = 1470; tr_f[idx, TR_MAT] = 2; tr_f[idx, TR_ET0] = 1e12
    tr_f[:, TR_MRK] = passive_markers(tr_x, L)
    return tr_x, tr_f


def mantle_tracers(nx, L, tracdens, rng, perturb=20.0, zrange=None, id0=0, xrange=None):
    """Model-1-like T-dependent mantle (values of pylamp2.py:146-154) with a conductive
    initial temperature plus a sinusoidal perturbation.
    zrange=(lo, hi) / xrange=(lo, hi) draw only this block's share of the tracers (multi-GPU benchmark set-up)."""
    n = int(np.prod(nx)) * tracdens
    if zrange is None and xrange is None:
        tr_x = rng.random((n, DIM)) * np.array(L)
    else:
        zlo, zhi = (0.0, L[IZ]) if zrange is None else (max(zrange[0], 0.0), min(zrange[1], L[IZ]))
        xlo, xhi = (0.0, L[IX]) if xrange is None else (max(xrange[0], 0.0), min(xrange[1], L[IX]))
        n = int(round(n * (zhi - zlo) / L[IZ] * (xhi - xlo) / L[IX]))
        tr_x = rng.random((n, DIM)) * np.array([zhi - zlo, xhi - xlo]) + np.array([zlo, xlo])
    tr_f = np.zeros((n, NFTRAC))
    tr_f[:, TR__ID] = np.arange(n) + id0
    tr_f[:, TR_RH0] = 3300; tr_f[:, TR_ALP] = 3.5e-5; tr_f[:, TR_MAT] = 2; tr_f[:, TR_ET0] = 1e20
    tr_f[:, TR_HCD] = 4.0; tr_f[:, TR_HCP] = 1250; tr_f[:, TR_ACE] = 120e3; tr_f[:, TR_IHT] = 0.02e-6 / 3300
    z, x = tr_x[:, IZ], tr_x[:, IX]
    tr_f[:, TR_TMP] = 273 + 1350 * z / L[IZ] + perturb * np.sin(3 * np.pi * x / L[IX]) * np.sin(np.pi * z / L[IZ])
    return tr_x, tr_f


class Simulation:
    def __init__(self, nx, L, tr_x=None, tr_f=None, options=None, device=None, grid=None, local=None):
        """grid: optional [z, x] node coordinates from 0 to L (rectilinear, strictly increasing).  The stock
        driver only builds regular grids (pylamp2.py:90); a non-uniform one makes the marker kernels locate
        cells by per-axis search."""
        self.nx = [int(nx[0]), int(nx[1])]
        self.L = [float(L[0]), float(L[1])]
        if grid is None:
            self.grid = [np.linspace(0, self.L[i], self.nx[i]) for i in range(DIM)]   # pylamp2.py:90
        else:
            self.grid = [np.ascontiguousarray(grid[i], dtype=np.float64) for i in range(DIM)]
            for i in range(DIM):
                if self.grid[i].size != self.nx[i] or np.any(np.diff(self.grid[i]) <= 0):
                    raise Exception("grid: need nx strictly increasing coordinates per axis")
                if abs(self.grid[i][0]) > 0 or abs(self.grid[i][-1] - self.L[i]) > 1e-12 * self.L[i]:
                    raise Exception("grid: coordinates must run from 0 to L")
        self.opt = options or Options()
        # local = (group, rank, Pz, Px): a virtual rank of an in-process group (VirtualCluster) instead of torch.distributed
        self.ctx = Context(self.nx, self.grid, device=device, attach_dist=local is None)
        if local is not None:
            self.ctx.attach_local(*local)
        self.it = 0
        self.totaltime = 0.0
        self.last = None
        self.ntrac = 0
        if tr_x is not None:
            self.upload(tr_x, tr_f)

    # -- tracer state -------------------------------------------------------------------------
    def block(self):
        """[zlo, zhi) x [xlo, xhi) of the node block this rank owns (the whole plane on one rank; blocks at a domain
        wall extend to infinity so that every tracer has exactly one owner)."""
        if self.ctx.nranks == 1:
            return (-np.inf, np.inf), (-np.inf, np.inf)
        i0, ni, j0, nj, Pz, Px = self.ctx.local_block()
        pz, px = self.ctx.rank // Px, self.ctx.rank % Px
        cz, cx = (self.nx[0] - 1) // Pz, (self.nx[1] - 1) // Px
        zr = (-np.inf if pz == 0 else self.grid[IZ][i0], np.inf if pz == Pz - 1 else self.grid[IZ][i0 + cz])
        xr = (-np.inf if px == 0 else self.grid[IX][j0], np.inf if px == Px - 1 else self.grid[IX][j0 + cx])
        return zr, xr

    def slab(self):
        """z-interval of this rank's block (kept for callers that only split rows)."""
        return self.block()[0]

    def upload(self, tr_x, tr_f):
        """Upload tracers; under several ranks every rank passes the full (or any superset of its) tracer set and keeps
        the ones inside its block."""
        tr_x = _lib.f64(tr_x); tr_f = _lib.f64(tr_f)
        if tr_x.shape[1] != DIM or tr_f.shape != (tr_x.shape[0], NFTRAC):
            raise Exception("tracer arrays must be (n,2) and (n,13)")
        if self.ctx.nranks > 1:
            (zlo, zhi), (xlo, xhi) = self.block()
            keep = (tr_x[:, IZ] >= zlo) & (tr_x[:, IZ] < zhi) & (tr_x[:, IX] >= xlo) & (tr_x[:, IX] < xhi)
            tr_x = np.ascontiguousarray(tr_x[keep]); tr_f = np.ascontiguousarray(tr_f[keep])
        self.ntrac = tr_x.shape[0]
        self.ctx.check(self.ctx.lib.pl_tracers_upload(self.ctx.handle(), self.ntrac, _lib.dptr(tr_x), _lib.dptr(tr_f)))

    def _refresh_count(self):
        n = C.c_int64()
        self.ctx.check(self.ctx.lib.pl_tracers_count(self.ctx.handle(), C.byref(n)))
        self.ntrac = n.value

    def tracers(self):
        """Tracers resident on THIS rank (all of them on one rank, in upload order)."""
        self._refresh_count()
        tr_x = np.empty((self.ntrac, DIM)); tr_f = np.empty((self.ntrac, NFTRAC))
        self.ctx.check(self.ctx.lib.pl_tracers_download(self.ctx.handle(), self.ntrac, _lib.dptr(tr_x), _lib.dptr(tr_f)))
        return tr_x, tr_f

    def gather_tracers(self):
        """All tracers of all ranks on every rank, ordered by TR__ID (tests / snapshots)."""
        tr_x, tr_f = self.tracers()
        v = self.tracer_velocity()
        if self.ctx.nranks > 1:
            import torch.distributed as dist
            parts = [None] * self.ctx.nranks
            dist.all_gather_object(parts, (tr_x, tr_f, v))
            tr_x = np.concatenate([p[0] for p in parts]); tr_f = np.concatenate([p[1] for p in parts])
            v = np.concatenate([p[2] for p in parts])
        o = np.argsort(tr_f[:, TR__ID], kind="stable")
        return tr_x[o], tr_f[o], v[o]

    def census(self):
        """Tracers per cell of this rank's block, shape (cell rows, cell columns) (pylamp2.py:588-598)."""
        r0 = C.c_int(); nr = C.c_int()
        self.ctx.check(self.ctx.lib.pl_tracers_census(self.ctx.handle(), 0, None, C.byref(r0), C.byref(nr)))
        _, _, j0, nj, _, _ = self.ctx.local_block()
        ncx = nj - 1 if j0 + nj >= self.nx[1] else nj
        cnt = np.empty((nr.value, ncx), dtype=np.int32)
        self.ctx.check(self.ctx.lib.pl_tracers_census(self.ctx.handle(), cnt.size, cnt.ctypes.data_as(C.POINTER(C.c_int32)),
                                                      C.byref(r0), C.byref(nr)))
        return cnt

    def tracer_velocity(self):
        self._refresh_count()
        v = np.empty((self.ntrac, DIM))
        self.ctx.check(self.ctx.lib.pl_get_tracer_velocity(self.ctx.handle(), self.ntrac, _lib.dptr(v)))
        return v

    def field(self, name):
        out = np.empty(self.nx)
        self.ctx.check(self.ctx.lib.pl_get_field(self.ctx.handle(), name.encode(), _lib.dptr(out)))
        return out

    # -- one time step ------------------------------------------------------------------------------
    def _config(self):
        o = self.opt
        c = _lib.StepConfig()
        c.do_heatdiff = int(o.do_heatdiff); c.do_subgrid_heatdiff = int(o.do_subgrid_heatdiff)
        c.tdep_rho = int(o.tdep_rho); c.tdep_eta = int(o.tdep_eta)
        c.etamin, c.etamax, c.tref = o.etamin, o.etamax, o.Tref
        c.tstep_adv_max, c.tstep_adv_min = o.tstep_adv_max, o.tstep_adv_min
        c.tstep_dif_max, c.tstep_dif_min = o.tstep_dif_max, o.tstep_dif_min
        c.tstep_modifier = o.tstep_modifier
        for w in range(4):
            c.bcstokes[w] = int(o.bcstokes[w]); c.bcheat[w] = int(o.bcheat[w]); c.bcheatvals[w] = float(o.bcheatvals[w])
        c.stokes_rtol, c.heat_rtol = o.stokes_rtol, o.heat_rtol
        c.stokes_maxit, c.heat_maxit = int(o.stokes_maxit), int(o.heat_maxit)
        c.length[0], c.length[1] = self.L
        c.tracdens, c.tracdens_min, c.inject_seed = int(o.tracdens), int(o.tracdens_min), int(o.inject_seed)
        c.surface_stabilization = int(bool(o.surface_stabilization))
        theta = float(o.surfstab_theta)
        c.surfstab_theta, c.surfstab_tstep = (theta if o.surfstab_strict_reference else -theta), float(o.surfstab_tstep)
        c.inject_unique_ids = int(bool(o.inject_unique_ids)); c.tracs_fence_disabled = int(not o.tracs_fence_enabled)
        return c

    def step(self):
        """Advance one time step; returns the report dict (time step, limiter, solver stats,
        per-stage milliseconds)."""
        self.it += 1
        cfg = self._config()
        rep = _lib.StepReport()
        self.ctx.check(self.ctx.lib.pl_step(self.ctx.handle(), C.byref(cfg), self.it, C.byref(rep)))
        self.totaltime += rep.tstep
        self.ntrac = rep.ntrac
        self.n_before_injection = rep.ntrac - rep.ninjected
        out = {k: getattr(rep, k) for k, _ in rep._fields_ if k not in ("stokes", "heat", "limiter")}
        out["limiter"] = "Ss" if chr(rep.limiter) == "s" else chr(rep.limiter)
        out["stokes"] = rep.stokes.as_dict(); out["heat"] = rep.heat.as_dict()
        out["it"] = self.it; out["time"] = self.totaltime
        self.last = out
        return out

    # -- the marker stages of a step one at a time (same kernels as step(); parity tests) ------------
    def scatter_fields(self, it=1):
        """Stages 1 + 2 of the step (properties, tracer -> grid: pylamp2.py:291-319) on the resident tracers;
        returns the grid fields the stage produces."""
        cfg = self._config()
        self.ctx.check(self.ctx.lib.pl_resident_scatter(self.ctx.handle(), C.byref(cfg), int(it)))
        names = ["rho", "etas", "etan"] + (["cp", "f_T", "H", "mat", "kz", "kx"] if self.opt.do_heatdiff else [])
        return {k: self.field(k) for k in names}

    def temp_to_tracers(self, newtemp, tstep, first=False):
        """Stage 5b of the step (pylamp2.py:436-480): new nodal temperature -> tracers (+ subgrid diffusion); the old nodal
        temperature is the f_T of the last scatter_fields().  Returns the tracers' temperature column (upload order)."""
        cfg = self._config()
        nt = _lib.f64(newtemp)
        if nt.shape != tuple(self.nx):
            raise Exception("temp_to_tracers: newtemp must be (nz, nx)")
        self.ctx.check(self.ctx.lib.pl_resident_temp_to_tracers(self.ctx.handle(), C.byref(cfg), int(bool(first)), _lib.dptr(nt), float(tstep)))
        return self.tracers()[1][:, TR_TMP]

    def layout(self):
        """(epoch age, lazy columns pending) of the resident tracer columns -- see pl_tracers_layout."""
        a = C.c_int(); b = C.c_int()
        self.ctx.check(self.ctx.lib.pl_tracers_layout(self.ctx.handle(), C.byref(a), C.byref(b)))
        return a.value, b.value

    def advect(self, vz_pad, vx_pad, tstep, fence=None, download=True):
        """Stage 6b of the step: RK4 through velocities on the padded (nz+1, nx+1) centre grid (pylamp2.py:547-572),
        then the end-of-step sort.  Returns (tracer velocities, new positions) in upload order, like pylamp_trac.RK
        (download=False: nothing -- a download brings every tracer column into the sorted order, which the step itself
        does not do)."""
        vz = _lib.f64(vz_pad); vx = _lib.f64(vx_pad)
        if vz.shape != (self.nx[0] + 1, self.nx[1] + 1) or vx.shape != vz.shape:
            raise Exception("advect: velocities must be (nz+1, nx+1)")
        fence = self.opt.tracs_fence_enabled if fence is None else fence
        Lc = (C.c_double * 2)(*self.L)
        self.ctx.check(self.ctx.lib.pl_resident_rk4(self.ctx.handle(), _lib.dptr(vz), _lib.dptr(vx), float(tstep), int(bool(fence)), Lc))
        if not download:
            return None
        return self.tracer_velocity(), self.tracers()[0]

    # -- snapshot writer (pylamp2.py:637-650) ----------------------------------------------------------
    def write_snapshot(self, outdir="out"):
        """Collective under several ranks: fields and tracers are gathered, rank 0 writes the two files."""
        velz, velx, pres, rho = self.field("velz"), self.field("velx"), self.field("pres"), self.field("rho")
        temp = self.field("temp") if self.opt.do_heatdiff else velx * 0.0
        if self.ctx.nranks > 1:
            tr_x, tr_f, tr_v = self.gather_tracers()
            if self.ctx.rank != 0:
                return
        else:
            tr_x, tr_f = self.tracers(); tr_v = self.tracer_velocity()
            # the stock snapshot holds the state BEFORE this step's injection (prev_tr_x / prev_tr_f, pylamp2.py:599-600);
            # injected tracers are appended behind the resident ones
            k = getattr(self, "n_before_injection", tr_x.shape[0])
            tr_x, tr_f, tr_v = tr_x[:k], tr_f[:k], tr_v[:k]
        os.makedirs(outdir, exist_ok=True)
        np.savez(os.path.join(outdir, "griddata.{:06d}.npz".format(self.it)), gridz=self.grid[IZ], gridx=self.grid[IX],
                 velz=velz, velx=velx, pres=pres, rho=rho, temp=temp, tstep=self.it, time=self.totaltime)
        np.savez(os.path.join(outdir, "tracs.{:06d}.npz".format(self.it)), tr_x=tr_x, tr_f=tr_f,
                 tr_v=tr_v, tstep=self.it, time=self.totaltime)

    def close(self):
        self.ctx.close()


class VirtualCluster:
    """Pz x Px virtual ranks in ONE process, each a Simulation with its own device context and host thread, joined by
    the library's in-process transport (pl_local_group_*).  The grid is decomposed exactly as on Pz*Px GPUs -- the same
    pack / unpack kernels, halo logic, replicated multigrid levels and 8-neighbour tracer migration -- only the
    messages are device-to-device copies.  This is how the 2 x 4 layout is rehearsed on a single GPU."""

    def __init__(self, nx, L, Pz, Px, tr_x, tr_f, options=None, device=None, grid=None):
        import concurrent.futures
        lib = _lib.load()
        self.Pz, self.Px, self.size = int(Pz), int(Px), int(Pz) * int(Px)
        g = C.c_void_p()
        if lib.pl_local_group_create(C.byref(g), self.size) != 0:
            raise Exception("pl_local_group_create failed")
        self.group, self.lib = g, lib
        self.pool = concurrent.futures.ThreadPoolExecutor(max_workers=self.size)
        self.sims = [Simulation(nx, L, options=options, device=device, grid=grid, local=(g, r, Pz, Px)) for r in range(self.size)]
        self.all(lambda s: s.upload(tr_x, tr_f))

    def all(self, fn, timeout=900):
        """fn(sim) on every rank at once (the calls are collective); returns the list of results in rank order."""
        import concurrent.futures as cf

        def run(s):                      # (PYLAMP_LOCAL_SERIAL=1: one rank at a time on the shared GPU, see pl_local_group_enter)
            self.lib.pl_local_group_enter(self.group)
            try:
                return fn(s)
            finally:
                self.lib.pl_local_group_leave(self.group, s.ctx.h)
        futs = [self.pool.submit(run, s) for s in self.sims]
        done, pending = cf.wait(futs, timeout=timeout, return_when=cf.FIRST_EXCEPTION)
        if pending:                      # a rank raised (or the time is up): release the others from their collective calls
            self.lib.pl_local_group_abort(self.group)
            cf.wait(futs, timeout=60)
        return [f.result(timeout=1) for f in futs]

    def step(self):
        return self.all(lambda s: s.step())

    def field(self, name):
        return self.all(lambda s: s.field(name))[0]          # every rank receives the assembled global array

    def tracers(self):
        """All tracers of all ranks, ordered by TR__ID."""
        parts = self.all(lambda s: s.tracers() + (s.tracer_velocity(),))
        tr_x = np.concatenate([p[0] for p in parts]); tr_f = np.concatenate([p[1] for p in parts]); v = np.concatenate([p[2] for p in parts])
        o = np.argsort(tr_f[:, TR__ID], kind="stable")
        return tr_x[o], tr_f[o], v[o]

    def comm_stats(self, reset=False):
        out = []
        for s in self.sims:
            cc = (C.c_int64 * 4)()
            s.ctx.check(self.lib.pl_comm_stats(s.ctx.handle(), cc, 1 if reset else 0))
            out.append([int(v) for v in cc])
        return out

    def close(self):
        for s in self.sims:
            s.close()
        self.pool.shutdown(wait=False)
        self.lib.pl_local_group_destroy(self.group)


def run(nx, L, model="block", tracdens=4, steps=5, seed=0, outdir=None, options=None):
    """Small stand-alone run, e.g. run([41,41],[660e3,660e3]) = BASELINE config 1."""
    rng = np.random.default_rng(seed)
    if model == "block":
        tr_x, tr_f = falling_block_tracers(nx, L, tracdens, rng)
        options = options or Options(do_heatdiff=False, tdep_rho=False, tdep_eta=False)
    else:
        tr_x, tr_f = mantle_tracers(nx, L, tracdens, rng)
        options = options or Options()
    sim = Simulation(nx, L, tr_x, tr_f, options)
    reports = []
    for _ in range(steps):
        reports.append(sim.step())
        if outdir:
            sim.write_snapshot(outdir)
    return sim, reports
