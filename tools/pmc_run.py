"""PMC / kernel-trace target: the Stokes stencil (row-scaled and plain) at 2049^2 and 4097^2, the finest-level multigrid
kernels (through a short solve), and a stream triad with known bytes (2 GiB read, 1 GiB written, 16 B per lane) as the
calibration of FETCH_SIZE / WRITE_SIZE.  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE` in separate
passes (and `--kernel-trace --stats` for durations), then feed the CSVs to tools/make_traffic.py."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd import pylamp_stokes as S, _context

sizes = [int(a) for a in sys.argv[1:]] or [2049, 4097]
for n in sizes:
    rng = np.random.default_rng(1)
    nx = [n, n]
    grid = [np.linspace(0, 660e3, n), np.linspace(0, 660e3, n)]
    etas = 1e19 * 10 ** rng.uniform(0, 3, nx); etan = 1e19 * 10 ** rng.uniform(0, 3, nx)
    rho = 3300 + rng.uniform(-50, 50, nx)
    A, rhs = S.makeStokesMatrix(nx, grid, etas, etan, rho, [1, 1, 1, 1])
    ctx = A._ctx
    ms = C.c_double()
    ctx.check(ctx.lib.pl_stokes_apply_bench(ctx.h, 10, C.byref(ms))); t_plain = ms.value
    ctx.check(ctx.lib.pl_stokes_apply_scaled_bench(ctx.h, 10, C.byref(ms))); t_scaled = ms.value
    x = S.solve(A, rhs, rtol=1e-2, maxit=3)          # brings the multigrid / vector kernels into the trace
    ctx.check(ctx.lib.pl_stokes_sweep_bench(ctx.h, 10, C.byref(ms))); t_sweep = ms.value
    if n == sizes[0]:
        ctx.check(ctx.lib.pl_stream_triad_bench(ctx.h, 1 << 27, 5, C.byref(ms)))
        print("triad %.3f ms -> %.1f GB/s" % (ms.value, 24.0 * (1 << 27) / ms.value / 1e6))
    print("n=%d plain %.4f ms scaled %.4f ms sweep %.4f ms" % (n, t_plain, t_scaled, t_sweep))
    del A
    _context.clear_contexts()
