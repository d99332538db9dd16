"""CPU-side checks of the C-ABI boundary: the library loads, exports every symbol that
include/pylamp_hip.h declares, and the product path fails loudly without a GPU."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def _declared():
    txt = open(os.path.join(ROOT, "include", "pylamp_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pl3?_[a-z0-9_]+)\s*\(", txt)))


def test_header_symbols_exported_and_bound():
    from pylamp_amd import _lib
    lib = _lib.load()
    names = _declared()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), "missing export " + n
        assert n in _lib.SIGNATURES, "no ctypes signature for " + n
    assert sorted(_lib.SIGNATURES) == names


def test_struct_layouts_match_header():
    import ctypes as C
    from pylamp_amd import _lib
    # the compiled library reports its own sizeof / offsetof: the ctypes mirror must agree
    lay = (C.c_size_t * 8)()
    assert _lib.load().pl_abi_layout(lay) == 0
    SC, SR = _lib.StepConfig, _lib.StepReport
    assert list(lay) == [C.sizeof(_lib.SolveStats), C.sizeof(SC), C.sizeof(SR), SC.length.offset, SC.inject_seed.offset,
                         SC.tracs_fence_disabled.offset, SR.ntrac.offset, SR.nremoved.offset]
    assert C.sizeof(_lib.SolveStats) == 48


def test_product_never_imports_oracle():
    pat_py = re.compile(r"^\s*(from|import)\s+\.*oracle|[\"']oracle[/\"']|oracle/_ref", re.M)
    pat_c = re.compile(r"#include\s*[<\"][^\n]*oracle|dlopen\([^\n]*oracle", re.M)
    for dirpath, _, files in os.walk(os.path.join(ROOT, "pylamp_amd")):
        for f in files:
            src = open(os.path.join(dirpath, f), errors="ignore").read() if f.endswith((".py", ".hip", ".h")) else ""
            if f.endswith(".py"):
                assert not pat_py.search(src), f
            elif src:
                assert not pat_c.search(src), f


def test_no_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from pylamp_amd import pylamp_stokes as S
    nx = [9, 7]
    g = [np.linspace(0, 1, 9), np.linspace(0, 1, 7)]
    e = np.ones(nx)
    with pytest.raises(Exception, match="no HIP device|no CPU fallback"):
        S.makeStokesMatrix(nx, g, e, e, e, [1, 1, 1, 1])


def test_pure_python_helpers_match_oracle(oracle):
    from pylamp_amd import pylamp_stokes as S, pylamp_diff as D, pylamp_const as K
    nx = [7, 5]
    x = np.arange(3 * 35, dtype=float)
    (vz, vx), p = S.x2vp(x, nx)
    (oz, ox), op = oracle.x2vp(x, nx)
    assert np.array_equal(vz, oz) and np.array_equal(vx, ox) and np.array_equal(p, op)
    assert S.gidx([2, 3], nx, 2) + K.IP == (2 * 5 + 3) * 3 + 2
    assert D.gidx([2, 3], nx) == 13
    assert np.array_equal(D.x2t(np.arange(35.0), nx), np.arange(35.0).reshape(7, 5))
    for name in ("DIM", "IZ", "IX", "IP", "SECINYR", "GASR", "NFTRAC", "TR_RHO", "TR_ETA", "TR_TMP", "TR_HCD",
                 "TR_HCP", "TR_RH0", "TR_ALP", "TR_MAT", "TR_ACE", "TR_ET0", "TR_IHT", "TR__ID", "EPS"):
        assert getattr(K, name) == getattr(oracle, name), name
