import sys, time, ctypes as C
import numpy as np
sys.path.insert(0, __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.abspath(__file__)), '..'))
from pylamp_amd import pylamp_stokes as S, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2049
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
rng = np.random.default_rng(1)
nx = [n, n]
grid = [np.linspace(0, 660e3, n), np.linspace(0, 660e3, n)]
etas = 1e19 * 10 ** rng.uniform(0, 3, nx); etan = 1e19 * 10 ** rng.uniform(0, 3, nx)
rho = 3300 + rng.uniform(-50, 50, nx)
A, rhs = S.makeStokesMatrix(nx, grid, etas, etan, rho, [1, 1, 1, 1])
ctx = A._ctx
print(ctx.device_info())
ms = C.c_double()
for k in range(3):
    ctx.check(ctx.lib.pl_stokes_apply_bench(ctx.h, reps, C.byref(ms)))
    gb = 64.0 * n * n / 1e9
    print("apply %dx%d: %.3f ms  -> %.1f GB/s algorithmic (%.1f%% of 8 TB/s)" % (n, n, ms.value, gb / (ms.value * 1e-3), gb / (ms.value * 1e-3) / 80.0))
