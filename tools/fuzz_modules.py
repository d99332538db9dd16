"""Randomised campaign at the module level: Stokes apply / rhs / explicit matrix, heat apply / rhs / solve and the
three marker functions on random small shapes (down to the 5x5 minimum), uniform and stretched grids, all supported
wall types -- against the oracle.  Usage: python tools/fuzz_modules.py [ncases] [seed]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import pylamp_stokes as S, pylamp_diff as D, pylamp_trac as T
from oracle import pylamp_oracle as O

ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
mr = lambda a, b: float(np.nanmax(np.abs(a - b)) / max(np.nanmax(np.abs(b)), 1e-300)) if a.size else 0.0
bad = 0
for case in range(ncases):
    nz, nxx = int(rng.integers(5, 70)), int(rng.integers(5, 70))
    nx = [nz, nxx]; L = [float(rng.uniform(0.5, 3)), float(rng.uniform(0.5, 3))]
    stretched = bool(rng.integers(0, 2))
    def axis(n, Lx):
        if not stretched: return np.linspace(0, Lx, n)
        h = 1.0 + float(rng.uniform(0.2, 2.5)) * rng.random(n - 1)
        c = np.concatenate([[0.0], np.cumsum(h)]); c *= Lx / c[-1]; c[-1] = Lx
        return c
    grid = [axis(nz, L[0]), axis(nxx, L[1])]
    gridmp = O.gridmp_of(grid)
    desc = "case %d: %dx%d stretched=%d" % (case, nz, nxx, stretched)
    errs = {}
    try:
        # ---- Stokes operator
        etas = 10 ** rng.uniform(0, 4, nx); etan = 10 ** rng.uniform(0, 4, nx); rho = rng.uniform(1, 2, nx)
        bc = [int(rng.integers(0, 2)), 1, int(rng.integers(0, 2)), 1]
        ss = bool(rng.integers(0, 2)); ts = float(rng.uniform(0.1, 2)) if ss else None
        A, rhs = S.makeStokesMatrix(nx, grid, etas, etan, rho, bc, surfstab=ss, tstep=ts)
        Ar, br = O.stokes_csr(nx, grid, etas, etan, rho, bc, surfstab=ss, tstep=ts)
        x = rng.standard_normal(3 * nz * nxx)
        errs["apply"] = mr(A.matvec(x), Ar @ x); errs["rhs"] = mr(rhs, br)
        if nz * nxx <= 900: errs["csc"] = mr(A.tocsc().toarray(), Ar.toarray())
        # ---- heat
        Tf = rng.uniform(300, 1600, nx); kz = rng.uniform(1, 5, nx); kx = rng.uniform(1, 5, nx)
        cp = rng.uniform(800, 1300, nx); rh = rng.uniform(3000, 3400, nx); H = rng.uniform(0, 1e-8, nx)
        hbc = [int(rng.integers(0, 2)) for _ in range(4)]
        if all(b == 1 for b in hbc): hbc[0] = 0          # all-flux walls are singular
        hv = [float(rng.uniform(0, 1500)) if b == 0 else float(rng.uniform(-1e-2, 1e-2)) for b in hbc]
        dt = float(rng.uniform(0.1, 10)) * min(np.diff(grid[0]).min(), np.diff(grid[1]).min()) ** 2 * 3000 * 1000 / 5
        Ah, bh = D.makeDiffusionMatrix(nx, grid, gridmp, Tf, [kz, kx], cp, rh, H, hbc, hv, dt)
        Ahr, bhr = O.heat_csr(nx, grid, gridmp, Tf, [kz, kx], cp, rh, H, hbc, hv, dt)
        xt = rng.standard_normal(nz * nxx)
        errs["happly"] = mr(Ah.matvec(xt), Ahr @ xt); errs["hrhs"] = mr(bh, bhr)
        xs = D.solve(Ah, bh)
        import scipy.sparse.linalg as spl, scipy.sparse as sp
        errs["hsolve"] = mr(xs, spl.spsolve(sp.csc_matrix(Ahr), bhr))
        # ---- markers
        n = int(rng.integers(1, 4000))
        tr_x = (rng.random((n, 2)) * 1.3 - 0.15) * np.array(L) if rng.integers(0, 2) else rng.random((n, 2)) * np.array(L)
        nf = int(rng.integers(1, 7)); sch = [int(rng.choice([1, 2, 5, 6])) for _ in range(nf)]
        tr_f = 10 ** rng.uniform(-2, 3, (n, nf))
        tg = [[grid[0], grid[1]], [gridmp[0], gridmp[1]], [gridmp[0], grid[1]], [grid[0], gridmp[1]]][int(rng.integers(0, 4))]
        gf = [np.zeros(nx) for _ in sch]
        T.trac2grid(tr_x, tr_f, None, tg, gf, nx, avgscheme=sch)
        with O.rect_search(stretched):
            ref = O.trac2grid(tr_x, tr_f, tg, nx, sch)
        errs["t2g"] = max(mr(a, b) for a, b in zip(gf, ref))
        assert all(np.array_equal(np.isnan(a), np.isnan(b)) for a, b in zip(gf, ref)), "NaN masks differ"
        F = [rng.standard_normal(nx), rng.standard_normal(nx)]
        meth = int(rng.choice([8, 16, 32]))
        gx_ = tr_x.copy()
        if not stretched:       # the reference raises IndexError for a marker within one cell beyond the high wall (pylamp_trac.py:52,75)
            gx_ = np.minimum(gx_, np.array(L) * (1 - 1e-9))
        tf = np.zeros((n, 2)); T.grid2trac(gx_, tf, grid, F, nx, defval=-1.5, method=meth)
        with O.rect_search(stretched):
            errs["g2t"] = mr(tf, O.grid2trac(gx_, grid, F, nx, defval=-1.5, method=meth))
        ok = all(v < 1e-9 for v in errs.values())
        print(("ok   " if ok else "FAIL ") + desc + "  " + " ".join("%s=%.1e" % kv for kv in errs.items()), flush=True)
        bad += 0 if ok else 1
    except Exception as ex:
        bad += 1
        print("EXC  " + desc + "  " + repr(ex)[:300], flush=True)
print("failures:", bad)
