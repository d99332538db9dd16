import os, sys, numpy as np
sys.path.insert(0, '.')
from pylamp_amd import driver
res = {}
for ep in ("64", "0"):
    os.environ["PYLAMP_EPOCH"] = ep
    nx = [513, 513]; L = [660e3, 660e3]
    tr_x, tr_f = driver.mantle_tracers(nx, L, 16, np.random.default_rng(11))
    n0 = tr_x.shape[0]
    opt = driver.Options(tracdens=16, tracdens_min=8, inject_unique_ids=True)
    sim = driver.Simulation(nx, L, tr_x, tr_f, opt)
    inj = 0; its = []
    for it in range(200):
        rep = sim.step(); inj += rep["ninjected"]; its.append(rep["stokes"]["iterations"])
        assert rep["stokes"]["converged"] == 1, (it, rep["stokes"])
    X, F = sim.tracers()
    res[ep] = (sim.field("temp"), sim.field("velz"), X.shape[0], inj, np.mean(its), sim.layout(), F[:n0, [4, 5, 6, 12]].copy(), tr_f[:, [4, 5, 6, 12]].copy())
    print("epoch", ep, "tracers", X.shape[0], "injected", inj, "mean its %.2f" % np.mean(its), "layout after download", sim.layout(), flush=True)
    sim.close()
a, b = res["64"], res["0"]
rel = lambda x, y: np.linalg.norm(x - y) / np.linalg.norm(y)
print("temp rel diff", rel(a[0], b[0]), "velz rel diff", rel(a[1], b[1]), "tracers", a[2], b[2])
print("constants intact (epoch 64):", np.array_equal(a[6], a[7]), "(epoch 0):", np.array_equal(b[6], b[7]))
