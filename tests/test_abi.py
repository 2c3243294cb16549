"""The C-ABI boundary without a GPU: libcomat_hip.so (cross-compiled for gfx950) loads, exports every entry point that
include/comat_hip.h declares, the ctypes binding covers exactly that set, and argument validation reports errors through
the return code + comat_last_error() instead of launching."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "comat_hip.h")


def declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|int64_t|const char\s*\*)\s+(comat_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from comat_amd import _hip
    lib = _hip.load_library()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/comat_hip.h but not exported"
    bound = set(_hip.SIGNATURES) | {"comat_abi_version", "comat_last_error", "comat_build_id"}
    assert bound == set(names), (bound ^ set(names))
    assert lib.comat_abi_version() == 8


def test_struct_layouts_match_header():
    from comat_amd import _hip
    # 6 pointers + 18 int64 + 2 float + 6 int32 + pointer + int64 + 2 scale pointers (ABI 3)
    # ABI 5: + C2, ldc2, epi2 (+ 4 bytes of padding); ABI 7: + B2, n2, sB2_tail, sC2_tail, alpha2 (+ 4 bytes of padding)
    # ABI 8: + s_scale_b, + A2k, B2k, K2, lda2k, ldb2k, sA2k, sB2k, + q8, q_scale, q_amax, ldq8
    assert C.sizeof(_hip.GemmParams) == 6 * 8 + 18 * 8 + 2 * 4 + 6 * 4 + 8 + 8 + 2 * 8 + 8 + 8 + 8 + 4 * 8 + 8 + 8 + 7 * 8 + 4 * 8
    assert C.sizeof(_hip.TTProblem) == 3 * 8 + 6 * 8  # comat_tt_problem (ABI 4)
    assert C.sizeof(_hip.ConvParams) == 6 * 8 + 13 * 4 + 2 * 4 + 4 * 4 + 4 + 8 + 8 + 2 * 8  # incl. 4 bytes of padding


def test_argument_validation_reports_errors_without_launching():
    from comat_amd import _hip
    lib = _hip.load_library()
    p = _hip.GemmParams()  # all null / zero
    assert lib.comat_gemm(C.byref(p), None) == -1
    assert b"null operand" in lib.comat_last_error()
    assert lib.comat_unary(99, None, None, 0, 0.0, 0.0, 0, 0, None) == -1
    assert lib.comat_sumsq(None, 0, None, None, None) == -1


def test_product_fails_loudly_without_gpu_tensors():
    import torch

    from comat_amd import _hip
    k = _hip.HipKernels()
    x = torch.zeros(4, 4)
    with pytest.raises(RuntimeError, match="HBM"):
        k.unary(0, x, x.clone(), 16)
