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
