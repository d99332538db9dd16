"""Stream-triad rate of the box for a few launch shapes (PYLAMP_TRIAD_BLOCKS) and array sizes."""
import os, sys, ctypes as C
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import numpy as np
from pylamp_amd._context import Context
ctx = Context([9, 9], [np.linspace(0, 1, 9), np.linspace(0, 1, 9)])
ms = C.c_double()
for n in (1 << 25, 1 << 27):
    for nb in (1024, 4096, 8192, 32768, 262144):
        os.environ["PYLAMP_TRIAD_BLOCKS"] = str(nb)
        ctx.check(ctx.lib.pl_stream_triad_bench(ctx.h, n, 10, C.byref(ms)))
        print("n=2^%d blocks=%6d: %.3f ms -> %.0f GB/s" % (int(np.log2(n)), nb, ms.value, 24.0 * n / ms.value / 1e6))
