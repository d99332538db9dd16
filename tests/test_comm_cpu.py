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


def test_block_partition():
    from pylamp_amd.parallel import block_of, block_1d, choose_layout
    for n, P in ((2049, 1), (2049, 2), (2049, 4), (2049, 8), (4097, 8), (129, 4)):
        parts = [block_1d(n, P, p) for p in range(P)]
        assert parts[0][0] == 0 and sum(c for _, c in parts) == n
        for (a, c), (b, _) in zip(parts[:-1], parts[1:]):
            assert a + c == b and c % 2 == 0
    with pytest.raises(Exception):
        block_1d(2050, 4, 0)
    assert choose_layout(1) == (1, 1) and choose_layout(2) == (1, 2) and choose_layout(4) == (2, 2) and choose_layout(8) == (2, 4)
    owned = set()
    for r in range(8):
        i0, ni, j0, nj = block_of([4097, 4097], r, 2, 4)
        owned |= {(i, j) for i in (i0, i0 + ni - 1) for j in (j0, j0 + nj - 1)}
    assert (0, 0) in owned and (4096, 4096) in owned
