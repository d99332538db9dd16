"""pylamp_amd — MI355X-native drop-in for PyLamp's per-time-step hot path.

Modules with the reference's names and call surface (pylamp2.py:4-7 imports them):
    pylamp_amd.pylamp_const, pylamp_amd.pylamp_stokes, pylamp_amd.pylamp_diff,
    pylamp_amd.pylamp_trac
plus pylamp_amd.driver (the counterpart of pylamp2.py's time loop) — all backed by the
hand-written HIP kernels in libpylamp_hip.so through the C ABI of include/pylamp_hip.h.
There is no CPU fallback: importing works anywhere, calling needs a gfx950 GPU.
"""
__all__ = ["pylamp_const", "pylamp_stokes", "pylamp_diff", "pylamp_trac", "driver"]
