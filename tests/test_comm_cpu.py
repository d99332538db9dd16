"""N > 1 host logic on CPU: world_size 2 and 3, gloo (no GPU needed)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT


@pytest.mark.parametrize("nproc", [2, 3])
def test_comm_layer_gloo(nproc):
    port = 29600 + nproc
    env = dict(os.environ); env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist", "run_comm_cpu.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "PASS comm cpu" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_slab_partition():
    from pylamp_amd.parallel import slab_rows
    for nz, size in ((2049, 1), (2049, 2), (2049, 4), (2049, 8), (4097, 8), (129, 4)):
        rows = [slab_rows(nz, r, size) for r in range(size)]
        assert rows[0][0] == 0 and sum(n for _, n in rows) == nz
        for (a, n), (b, _) in zip(rows[:-1], rows[1:]):
            assert a + n == b and n % 2 == 0
    with pytest.raises(Exception):
        slab_rows(2050, 0, 4)
