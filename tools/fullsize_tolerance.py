"""Full-size check of the loose, warm-started time-step loop against a tightly converged, cold-started one:
    python tools/fullsize_tolerance.py [n=2049] [steps=6]
Run A: defaults (rtol 1e-7 + velocity-error estimate <= 3e-8, extrapolated initial guesses).  Run B: PYLAMP_STOKES_ETOL=1e-10,
stokes_rtol 1e-11, no warm starts.  Prints the relative L2 differences of velocity and temperature after the last step."""
import os, sys, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
code = r'''
import sys, json
sys.path.insert(0, %r)
import numpy as np
import bench
from pylamp_amd import driver
n, steps, tight = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
nx = [n, n]; L = [660e3, 660e3]
opt = driver.Options(tracdens=16, tracdens_min=9)
if tight: opt.stokes_rtol = 1e-11
sim = driver.Simulation(nx, L, options=opt, device=0)
tr_x, tr_f = driver.mantle_tracers(nx, L, 16, np.random.default_rng(20260103))
sim.upload(tr_x, tr_f); del tr_x, tr_f
its, ms = [], []
for k in range(steps):
    r = sim.step(); its.append(r["stokes"]["iterations"]); ms.append(r["ms_total"])
    assert r["stokes"]["converged"] == 1, r["stokes"]
np.save(sys.argv[4], np.stack([sim.field("velz"), sim.field("velx"), sim.field("temp")]))
print("RESULT", json.dumps(dict(its=its, ms=ms, time=sim.totaltime, est=r["stokes"]["error_estimate"], res=r["stokes"]["rel_residual"])))
''' % root
out = {}
for name, tight, env in (("loose_warm", 0, {}), ("tight_cold", 1, {"PYLAMP_STOKES_ETOL": "1e-10", "PYLAMP_X0_EXTRAP": "0", "PYLAMP_HEAT_X0": "0", "PYLAMP_SHADOW": "0"})):
    path = os.path.join(root, "gpurun_out", "fs_%s.npy" % name)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    r = subprocess.run([sys.executable, "-c", code, str(n), str(steps), str(tight), path], capture_output=True, text=True, env=dict(os.environ, **env))
    if r.returncode != 0:
        print(name, "FAILED", r.stderr[-2000:]); sys.exit(1)
    line = [l for l in r.stdout.splitlines() if l.startswith("RESULT")][0]
    out[name] = json.loads(line[7:])
    print(name, line[7:], flush=True)
import numpy as np
a = np.load(os.path.join(root, "gpurun_out", "fs_loose_warm.npy")); b = np.load(os.path.join(root, "gpurun_out", "fs_tight_cold.npy"))
for q, nm in enumerate(("velz", "velx", "temp")):
    print("%s rel L2 difference %.3e   max |diff| / max |field| %.3e" % (nm, np.linalg.norm(a[q] - b[q]) / np.linalg.norm(b[q]), np.abs(a[q] - b[q]).max() / np.abs(b[q]).max()))
print("model time: %.10e vs %.10e" % (out["loose_warm"]["time"], out["tight_cold"]["time"]))
for f in ("fs_loose_warm.npy", "fs_tight_cold.npy"): os.remove(os.path.join(root, "gpurun_out", f))
