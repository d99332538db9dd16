"""Stock model 5 (contrast 1e10, 201 x 41): error of the GPU solve against the refined direct solution for the iterative path at
several tolerances and for the banded-LU path.  python tools/model5_probe.py"""
import os, sys, subprocess, json
import numpy as np
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)

def one(rtol, maxit):
    from pylamp_amd import pylamp_stokes as S
    from oracle import pylamp_oracle as O
    g = np.load(os.path.join(ROOT, "tests", "golden", "stokes_solve_sphere201x41.npz"))
    nx = [int(v) for v in g["nx"]]; grid = [g["gz"], g["gx"]]; bc = list(g["bc"])
    A, rhs = S.makeStokesMatrix(nx, grid, g["etas"], g["etan"], g["rho"], bc)
    x = S.solve(A, rhs, rtol=rtol, maxit=maxit)
    xr = O.stokes_solve_refined(nx, grid, g["etas"], g["etan"], g["rho"], bc, refinements=4)
    (vz, vx), p = S.x2vp(x, nx); (rz, rx), rp = S.x2vp(xr, nx); (fz, fx), _ = S.x2vp(g["x"], nx)
    ev = np.sqrt((np.sum((vz - rz) ** 2) + np.sum((vx - rx) ** 2)) / (np.sum(rz ** 2) + np.sum(rx ** 2)))
    ef = np.sqrt((np.sum((vz - fz) ** 2) + np.sum((vx - fx) ** 2)) / (np.sum(fz ** 2) + np.sum(fx ** 2)))
    r = rhs - A @ x
    st = A.last_stats
    print(json.dumps(dict(rtol=rtol, err_vs_refined=float(ev), err_vs_fixture=float(ef), unscaled_res=float(np.linalg.norm(r) / np.linalg.norm(rhs)),
                          its=st["iterations"], conv=st["converged"], rel=st["rel_residual"], est=st["error_estimate"], direct=st["used_direct"],
                          ms=st["solve_ms"], env={k: v for k, v in os.environ.items() if k.startswith("PYLAMP_")})), flush=True)

if __name__ == "__main__":
    if len(sys.argv) > 1:
        one(float(sys.argv[1]), int(sys.argv[2]))
    else:
        for env, rtol, maxit in (({}, 1e-7, 400), ({}, 1e-10, 400), ({}, 1e-13, 1000), ({"PYLAMP_STOKES_ETOL": "1e-12"}, 1e-13, 1000),
                                 ({"PYLAMP_FORCE_DIRECT": "1"}, 1e-7, 400), ({"PYLAMP_FORCE_DIRECT": "1"}, 1e-12, 400),
                                 ({"PYLAMP_DEFLATE": "0"}, 1e-10, 400)):
            e = dict(os.environ); e.update(env)
            subprocess.run([sys.executable, os.path.abspath(__file__), str(rtol), str(maxit)], env=e)
