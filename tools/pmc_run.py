"""PMC target: a few Stokes applies + a streaming calibration kernel (k_axpy_out: 8 B/lane, known
bytes) at 2049^2, nothing else heavy.  Run under `rocprofv3 --pmc ...` (one counter set per pass)."""
import sys, ctypes as C
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import pylamp_stokes as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
rng = np.random.default_rng(1)
nx = [n, n]
grid = [np.linspace(0, 660e3, n), np.linspace(0, 660e3, n)]
etas = 1e19 * 10 ** rng.uniform(0, 3, nx); etan = 1e19 * 10 ** rng.uniform(0, 3, nx)
rho = 3300 + rng.uniform(-50, 50, nx)
A, rhs = S.makeStokesMatrix(nx, grid, etas, etan, rho, [1, 1, 1, 1])
ms = C.c_double()
A._ctx.check(A._ctx.lib.pl_stokes_apply_bench(A._ctx.h, 10, C.byref(ms)))
x = S.solve(A, rhs, rtol=1e-2, maxit=3)          # brings k_axpy_out / k_vv_cheb etc. into the trace
print("done", ms.value)
