"""Worker of tests/test_comm_cpu.py: the communication layer's host logic on CPU (gloo,
world_size 2..3).  A fake context stands in for the HIP library: its 'device memory' is host
memory, so Comm's staging transport (pack, neighbour exchange, reverse-halo add, all-gather,
variable-size tracer migration) runs end to end without a GPU."""
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

    def check(self, rc):
        assert rc == 0


dist.init_process_group(backend="gloo")
rank, size = dist.get_rank(), dist.get_world_size()
comm = parallel.Comm(FakeCtx())
assert comm.rank == rank and comm.size == size and not comm.device_mode


def ptr(a, off=0):
    return a.ctypes.data + 8 * off


# ---- slab partition mirrors pl_set_comm
rows = [parallel.slab_rows(2049, r, 8) for r in range(8)]
assert rows[0] == (0, 256) and rows[7] == (1792, 257) and sum(n for _, n in rows) == 2049

# ---- forward halo of a 2-plane "vector": (lnz+2) rows of `pitch` doubles per plane
pitch, lnz, nplanes = 16, 6, 2
plane = (lnz + 2) * pitch
v = np.zeros(nplanes * plane)
V = v.reshape(nplanes, lnz + 2, pitch)
for q in range(nplanes):
    for li in range(lnz):
        V[q, li + 1, :] = 1000 * rank + 100 * q + li          # owned rows
rc = comm._exchange(None, ptr(v, pitch), ptr(v, 0), ptr(v, lnz * pitch), ptr(v, (lnz + 1) * pitch), pitch, nplanes, plane, 0)
assert rc == 0, comm.errors
for q in range(nplanes):
    if rank > 0:
        assert np.all(V[q, 0, :] == 1000 * (rank - 1) + 100 * q + (lnz - 1))
    else:
        assert np.all(V[q, 0, :] == 0)
    if rank < size - 1:
        assert np.all(V[q, lnz + 1, :] == 1000 * (rank + 1) + 100 * q)
    else:
        assert np.all(V[q, lnz + 1, :] == 0)

# ---- reverse (accumulating) halo: ring rows are added into the neighbour's boundary rows
acc = np.zeros(plane); A = acc.reshape(lnz + 2, pitch)
A[:, :] = 1.0
rc = comm._exchange(None, ptr(acc, 0), ptr(acc, pitch), ptr(acc, (lnz + 1) * pitch), ptr(acc, lnz * pitch), pitch, 1, plane, 1)
assert rc == 0, comm.errors
assert np.all(A[1, :] == (2.0 if rank > 0 else 1.0)) and np.all(A[lnz, :] == (2.0 if rank < size - 1 else 1.0))
assert np.all(A[2:lnz, :] == 1.0)

# ---- in-place all-gather
cnt = 5
g = np.zeros(2 * (cnt * size + 3)); G = g.reshape(2, cnt * size + 3)
for q in range(2):
    G[q, rank * cnt:(rank + 1) * cnt] = 10 * rank + q
rc = comm._allgather(None, ptr(g), cnt, 2, cnt * size + 3)
assert rc == 0, comm.errors
for q in range(2):
    for r in range(size):
        assert np.all(G[q, r * cnt:(r + 1) * cnt] == 10 * r + q)

# ---- host all-reduce (sum / min / max)
b = (C.c_double * 3)(rank + 1.0, rank + 1.0, rank + 1.0)
for op, want in ((0, size * (size + 1) / 2), (1, 1.0), (2, float(size))):
    b2 = (C.c_double * 1)(rank + 1.0)
    assert comm._allreduce_host(None, b2, 1, op) == 0
    assert b2[0] == want

# ---- variable-size tracer migration: rank r sends (r+1) tracers down and (r+2) up, 3 columns
ncol, cap = 3, 64
n_lo, n_hi = (rank + 1 if rank > 0 else 0), (rank + 2 if rank < size - 1 else 0)
cols = [np.zeros(cap) for _ in range(ncol)]
send_lo = [np.full(max(n_lo, 1), 100 * rank + k, dtype=float) for k in range(ncol)]
send_hi = [np.full(max(n_hi, 1), 100 * rank + 50 + k, dtype=float) for k in range(ncol)]
PL = C.c_void_p * ncol
got = (C.c_int64 * 1)(0)
rc = comm._exchange_var(None, PL(*[a.ctypes.data for a in send_lo]), n_lo, PL(*[a.ctypes.data for a in send_hi]), n_hi,
                        PL(*[a.ctypes.data for a in cols]), cap, ncol, got)
assert rc == 0, comm.errors
m_lo = (rank - 1 + 2) if rank > 0 else 0               # what rank-1 sent up
m_hi = (rank + 1 + 1) if rank < size - 1 else 0        # what rank+1 sent down
assert got[0] == m_lo + m_hi
for k in range(ncol):
    assert np.all(cols[k][:m_lo] == 100 * (rank - 1) + 50 + k)
    assert np.all(cols[k][m_lo:m_lo + m_hi] == 100 * (rank + 1) + k)

dist.barrier()
if rank == 0:
    print("PASS comm cpu", flush=True)
dist.destroy_process_group()
