"""One process for a kernel trace: K time steps at 2049^2 / 16 markers per node on ONE rank (mode "one") or on Pz x Px virtual ranks of the
same GPU (mode "blocks") -- tools/rehearse_trace.sh runs both under rocprofv3 --kernel-trace with two step counts and takes the
difference, i.e. the summed kernel time of a steady-state time step: the figure the block decomposition is judged by while no
multi-GPU node is available (the wall time of virtual ranks means nothing: 8 contexts share one GPU).
    python tools/rehearse_trace.py one|blocks steps [n=2049] [Pz=2] [Px=4] [markers/node=16]"""
import sys, os, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import driver

mode = sys.argv[1]; steps = int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 2049
Pz = int(sys.argv[4]) if len(sys.argv) > 4 else 2
Px = int(sys.argv[5]) if len(sys.argv) > 5 else 4
dens = int(sys.argv[6]) if len(sys.argv) > 6 else 16
nx = [n, n]; L = [660e3, 660e3]
tr_x, tr_f = driver.mantle_tracers(nx, L, dens, np.random.default_rng(20260103))
opt = driver.Options(tracdens=dens, tracdens_min=(dens * 9) // 16)
if mode == "one":
    sim = driver.Simulation(nx, L, tr_x, tr_f, opt)
    reps = [sim.step() for _ in range(steps)]
    print(json.dumps(dict(mode=mode, steps=steps, its=[r["stokes"]["iterations"] for r in reps], ms=[round(r["ms_total"], 2) for r in reps])))
    sim.close()
else:
    vc = driver.VirtualCluster(nx, L, Pz, Px, tr_x, tr_f, opt)
    its = []; comm = []
    for _ in range(steps):
        vc.comm_stats(reset=True)
        reps = vc.step()
        its.append(reps[0]["stokes"]["iterations"])
        comm.append([max(s[k] for s in vc.comm_stats()) for k in range(4)])
    print(json.dumps(dict(mode=mode, layout=[Pz, Px], steps=steps, its=its, comm_calls_per_step=comm)))
    vc.close()
