"""Multi-rank HIP path on the single test GPU: N processes (gloo) share device 0; the grid is
decomposed into row slabs exactly as on N GPUs (only the transport differs: host staging
instead of RCCL)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _run(script, nproc, port):
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist", script)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    return r.returncode, r.stdout + r.stderr


@pytest.mark.parametrize("nproc", [2, 4])
def test_stokes_heat_row_slabs(nproc):
    rc, out = _run("run_stokes_2rank.py", nproc, 29510 + nproc)
    assert rc == 0, out[-3000:]
    for tag in ("PASS apply", "PASS solve", "PASS heat"):
        assert tag in out, out[-3000:]


@pytest.mark.parametrize("nproc", [2, 4])
def test_resident_step_row_slabs(nproc):
    """Full time step (scatter with reverse halo, solves, gathers, RK4, tracer migration); also on a graded grid
    (slabs of unequal thickness, per-axis cell search) and with the stabilisation-aware multigrid levels."""
    rc, out = _run("run_step_nrank.py", nproc, 29520 + nproc)
    assert rc == 0, out[-3000:]
    for tag in ("PASS block trajectory", "PASS mantle steps", "PASS graded grid steps", "PASS stabilisation loop"):
        assert tag in out, out[-3000:]


def test_device_pointer_views_are_zero_copy():
    """The RCCL transport wraps the library's raw device pointers as torch tensors; check on the
    real GPU that such a view aliases the memory (single process)."""
    import ctypes as C
    import numpy as np
    import torch
    from pylamp_amd import _lib
    from pylamp_amd._context import Context
    from pylamp_amd.parallel import _DevView
    ctx = Context([9, 9], [np.linspace(0, 1, 9), np.linspace(0, 1, 9)])
    t = torch.arange(16, dtype=torch.float64, device="cuda")
    v = torch.as_tensor(_DevView(t.data_ptr(), 16), device="cuda")
    v += 1.0
    torch.cuda.synchronize()
    assert torch.equal(t.cpu(), torch.arange(16, dtype=torch.float64) + 1)
    # and the library sees the same bytes through its own HIP runtime
    host = np.empty(16)
    ctx.check(ctx.lib.pl_memcpy_d2h(ctx.h, host.ctypes.data_as(C.c_void_p), C.c_void_p(t.data_ptr()), host.nbytes))
    assert np.array_equal(host, np.arange(16.0) + 1)
    ctx.close()


def test_nccl_staging_fallback_two_ranks():
    """Same worker with PYLAMP_COMM_STAGING unset under gloo already covers staging; here the
    collective 'agree on transport' path is exercised explicitly."""
    rc, out = _run("run_stokes_2rank.py", 2, 29541)
    assert rc == 0 and "PASS solve" in out, out[-2000:]


def test_native_rccl_single_rank_selftest():
    """dlopen(librccl), ncclCommInitRank and ncclAllGather through the library's own entry points,
    on one rank (the neighbour send/recv pairs need >= 2 GPUs and run on the multi-GPU node)."""
    code = r'''
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, %r)
os.environ["PYLAMP_RCCL_SELFTEST"] = "1"
from pylamp_amd import _lib
from pylamp_amd._context import Context
from pylamp_amd.parallel import CommOps
ctx = Context([17, 17], [np.linspace(0, 1, 17), np.linspace(0, 1, 17)])
ops = CommOps()
ctx.check(ctx.lib.pl_set_comm(ctx.h, 0, 1, C.byref(ops)))
nat = C.c_int(-1)
ctx.check(ctx.lib.pl_comm_info(ctx.h, None, None, C.byref(nat)))
print("NATIVE", nat.value)
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert "NATIVE 1" in r.stdout, (r.stdout + r.stderr)[-2000:]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the way the driver runs the N = 1 line): the script starts its two ranks
    itself before anything touches the GPU (they share the test GPU under gloo) and rank 0 prints "n_gpus": 2."""
    import json
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    env["PYLAMP_BENCH_NO_4097"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--grid", "257",
                        "--tracdens", "16", "--apply-reps", "5"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 2 and all(c == 1 for c in out["stokes_converged"]), out
    assert out["comm_calls_per_step"][3] == 0, out["comm_calls_per_step"]      # no host all-reduce inside the timed loop
