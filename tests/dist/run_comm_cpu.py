"""Worker of tests/test_comm_cpu.py: the communication layer's host logic on CPU (gloo,
world_size 2..3).  A fake context stands in for the HIP library: its 'device memory' is host
memory, so Comm's staging transport (point-to-point message groups with several messages per peer,
all-gather, host all-reduce) runs end to end without a GPU."""
import ctypes as C
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
from pylamp_amd import parallel        # noqa: E402


class FakeLib:
    def pl_set_comm(self, h, rank, size, ops):
        return 0

    def pl_comm_info(self, h, rank, size, native):
        return 0

    def pl_memcpy_d2h(self, h, dst, src, nbytes):
        C.memmove(dst, src, nbytes); return 0

    def pl_memcpy_h2d(self, h, dst, src, nbytes):
        C.memmove(dst, src, nbytes); return 0


class FakeCtx:
    lib = FakeLib(); h = None

    def handle(self):
        return None

    def check(self, rc):
        assert rc == 0


dist.init_process_group(backend="gloo")
rank, size = dist.get_rank(), dist.get_world_size()
comm = parallel.Comm(FakeCtx())
assert comm.rank == rank and comm.size == size and not comm.device_mode

# ---- block partition mirrors pl_set_comm / pl_block_1d
assert parallel.choose_layout(8) == (2, 4) and parallel.choose_layout(2) == (1, 2) and parallel.choose_layout(4) == (2, 2)
blocks = [parallel.block_of([2049, 4097], r, 2, 4) for r in range(8)]
assert blocks[0] == (0, 1024, 0, 1024) and blocks[7] == (1024, 1025, 3072, 1025)
assert sum(b[1] * b[3] for b in blocks) == 2049 * 4097


def call_sendrecv(msgs):
    n = len(msgs)
    peer = (C.c_int * n)(*[m[0] for m in msgs])
    sp = (C.c_void_p * n)(*[m[1].ctypes.data if m[1] is not None else None for m in msgs])
    ns = (C.c_int64 * n)(*[m[1].size if m[1] is not None else 0 for m in msgs])
    rp = (C.c_void_p * n)(*[m[2].ctypes.data if m[2] is not None else None for m in msgs])
    nr = (C.c_int64 * n)(*[m[2].size if m[2] is not None else 0 for m in msgs])
    rc = comm._sendrecv(None, n, peer, sp, ns, rp, nr)
    assert rc == 0, comm.errors


# ---- a halo-like exchange on a ring: two messages per neighbour (different sizes), matched in list order
left, right = (rank - 1) % size, (rank + 1) % size
msgs = []
recv = {}
for p in sorted(set([left, right])):
    for k, n in enumerate((5, 3)):
        s = np.full(n, 1000.0 * rank + 100 * p + k)
        r = np.zeros(n)
        recv[(p, k)] = r
        msgs.append((p, s, r))
call_sendrecv(msgs)
for (p, k), r in recv.items():
    assert np.all(r == 1000.0 * p + 100 * rank + k), (rank, p, k, r)

# ---- one-directional messages (tracer migration: counts differ per direction, some zero)
if size >= 2:
    s = np.arange(rank + 1, dtype=float) + 10 * rank          # rank sends rank+1 values to its right neighbour
    r = np.zeros(left + 1)
    msgs = [(right, s, None), (left, None, r)] if right != left else [(right, s, r)]
    call_sendrecv(msgs)
    assert np.all(r == np.arange(left + 1) + 10 * left)

# ---- all-gather
cnt = 5
mine = np.full(cnt, 10.0 * rank)
out = np.zeros(cnt * size)
assert comm._allgather(None, mine.ctypes.data, out.ctypes.data, cnt) == 0, comm.errors
for r_ in range(size):
    assert np.all(out[r_ * cnt:(r_ + 1) * cnt] == 10.0 * r_)

# ---- host all-reduce (sum / min / max)
for op, want in ((0, size * (size + 1) / 2), (1, 1.0), (2, float(size))):
    b2 = (C.c_double * 1)(rank + 1.0)
    assert comm._allreduce_host(None, b2, 1, op) == 0
    assert b2[0] == want

dist.barrier()
if rank == 0:
    print("PASS comm cpu", flush=True)
dist.destroy_process_group()
