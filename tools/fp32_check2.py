"""Isolated comparison of the preconditioner with FP32 / FP64 multigrid levels (same hierarchy, alternating calls)."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import pylamp_stokes as S
from tools.fp32_check import model

smooth_rhs = len(sys.argv) > 2 and sys.argv[2] == "smooth"
n = int(sys.argv[1]) if len(sys.argv) > 1 else 513
nx, grid, es, en, rho = model(n)
A, rhs = S.makeStokesMatrix(nx, grid, es, en, rho, [1, 1, 1, 1])
x = S.solve(A, rhs)
rng = np.random.default_rng(0)
if smooth_rhs:
    r = np.array(rhs, copy=True)           # the physical load: smooth, large-scale response
else:
    r = rng.standard_normal(A.shape[0]) * np.abs(rhs).max()
v = lambda z: z.reshape(n, n, 3)[:, :, :2]
zs = []
for k, (fp32, mn) in enumerate([(False, 0), (True, 1000), (False, 0), (True, 200000), (False, 0)]):
    A.set_mg_precision(fp32, mn)
    z = A.precond(r)
    zs.append(z)
    print(k, "fp32" if fp32 else "fp64", A.mg_precision(), "lmax0 %.5f" % A.mg_info()[1][0],
          "diff to first %.3e" % (np.linalg.norm(v(z) - v(zs[0])) / np.linalg.norm(v(zs[0]))), flush=True)
# high-frequency content of the difference: A applied to it, relative to A applied to z
d = zs[1] - 0.5 * (zs[0] + zs[2])
print("A*diff / A*z  %.3e" % (np.linalg.norm(A @ d) / np.linalg.norm(A @ zs[0])))
