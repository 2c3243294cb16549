"""ctypes binding of libcomat_hip.so (the C ABI declared in include/comat_hip.h).

This module is the ONLY place where device pointers leave PyTorch.  It takes torch tensors that live in HBM,
passes raw pointers + sizes + the current HIP stream to the C entry points, and raises on any error.  There is no
CPU path here: if the shared library is missing or a tensor is not on the GPU, it fails loudly.
"""
from __future__ import annotations

import ctypes as C
import os

import torch

F32, BF16, FP8 = 0, 1, 2  # COMAT_F32, COMAT_BF16, COMAT_FP8_E4M3
ACT_NONE, ACT_SILU, ACT_GELU = 0, 1, 2
UN_COPY, UN_SILU, UN_GELU, UN_AFFINE = 0, 1, 2, 3

_LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libcomat_hip.so")


class GemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p),
        ("bias", C.c_void_p), ("bias2", C.c_void_p), ("R", C.c_void_p),
        ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64),
        ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64), ("ldr", C.c_int64),
        ("batch1", C.c_int64), ("batch2", C.c_int64),
        ("sA1", C.c_int64), ("sA2", C.c_int64), ("sB1", C.c_int64), ("sB2", C.c_int64),
        ("sC1", C.c_int64), ("sC2", C.c_int64), ("sR1", C.c_int64), ("sR2", C.c_int64),
        ("rows_per_bias2", C.c_int64),
        ("alpha", C.c_float), ("beta", C.c_float),
        ("transA", C.c_int32), ("transB", C.c_int32), ("act", C.c_int32),
        ("in_dtype", C.c_int32), ("out_dtype", C.c_int32), ("r_dtype", C.c_int32),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("scale_a", C.c_void_p), ("scale_b", C.c_void_p),
        ("C2", C.c_void_p), ("ldc2", C.c_int64), ("epi2", C.c_int32),
        ("B2", C.c_void_p), ("n2", C.c_int64), ("sB2_tail", C.c_int64), ("sC2_tail", C.c_int64), ("alpha2", C.c_float),  # epi2 = 4 (ABI 7)
        ("s_scale_b", C.c_int64),  # ABI 8
        ("A2k", C.c_void_p), ("B2k", C.c_void_p), ("K2", C.c_int64), ("lda2k", C.c_int64), ("ldb2k", C.c_int64),
        ("sA2k", C.c_int64), ("sB2k", C.c_int64),  # bf16 k-tail of an fp8 product (ABI 8)
        ("q8", C.c_void_p), ("q_scale", C.c_void_p), ("q_amax", C.c_void_p), ("ldq8", C.c_int64),  # GEGLU epilogue -> e4m3 (ABI 8)
    ]


class GemmSegment(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("K", C.c_int64), ("lda", C.c_int64), ("ldb", C.c_int64),
                ("sA", C.c_int64), ("sB", C.c_int64)]


class ConvParams(C.Structure):
    _fields_ = [
        ("X", C.c_void_p), ("W", C.c_void_p), ("Y", C.c_void_p),
        ("bias", C.c_void_p), ("bias2", C.c_void_p), ("R", C.c_void_p),
        ("B", C.c_int32), ("Hin", C.c_int32), ("Win", C.c_int32), ("Cin", C.c_int32),
        ("Hout", C.c_int32), ("Wout", C.c_int32), ("Cout", C.c_int32),
        ("KH", C.c_int32), ("KW", C.c_int32), ("stride", C.c_int32), ("pad", C.c_int32),
        ("mode", C.c_int32), ("ups", C.c_int32),
        ("alpha", C.c_float), ("beta", C.c_float),
        ("act", C.c_int32), ("in_dtype", C.c_int32), ("out_dtype", C.c_int32), ("r_dtype", C.c_int32),
        ("ws", C.c_void_p), ("ws_bytes", C.c_int64),
        ("scale_a", C.c_void_p), ("scale_b", C.c_void_p),
    ]


class TTProblem(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("M", C.c_int64), ("N", C.c_int64),
                ("K", C.c_int64), ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64)]


_vp, _i32, _i64, _f = C.c_void_p, C.c_int32, C.c_int64, C.c_float

# name -> argtypes (restype is int unless noted); mirrors include/comat_hip.h one to one.
SIGNATURES = {
    "comat_gemm": [C.POINTER(GemmParams), _vp],
    "comat_gemm_segments": [C.POINTER(GemmParams), C.POINTER(GemmSegment), _i32, _vp],
    "comat_gemm_tt_grouped": [C.POINTER(TTProblem), _i32, _i32, _vp, _i64, _vp],
    "comat_conv2d": [C.POINTER(ConvParams), _vp],
    "comat_groupnorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f, _i32, _i32, _vp],
    "comat_groupnorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _i32, _vp, _i32, _vp],
    "comat_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f, _i32, _vp],
    "comat_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _vp, _i32, _vp],
    "comat_softmax_fwd": [_vp, _vp, _i64, _i32, _i32, _i32, _i32, _vp, _i64, _i32, _i32, _vp],
    "comat_softmax_bwd": [_vp, _vp, _vp, _i64, _i32, _f, _i32, _i32, _i32, _vp],
    "comat_flash_attn_fwd": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f, _i32, _vp],
    "comat_flash_attn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64,
                             _i64, _i64, _f, _i32, _vp, _i64, _vp],
    "comat_unary": [_i32, _vp, _vp, _i64, _f, _f, _i32, _i32, _vp],
    "comat_unary_bwd": [_i32, _vp, _vp, _vp, _i64, _i32, _vp],
    "comat_axpby": [_f, _vp, _f, _vp, _vp, _i64, _i32, _i32, _i32, _vp],
    "comat_geglu_fwd": [_vp, _vp, _i64, _i32, _i32, _vp],
    "comat_geglu_bwd": [_vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "comat_geglu_il_fwd": [_vp, _vp, _i64, _i32, _i32, _vp],
    "comat_geglu_il_bwd": [_vp, _vp, _vp, _i64, _i32, _i32, _vp],
    "comat_copy2d": [_vp, _i64, _vp, _i64, _i64, _i64, _i32, _i32, _vp],
    "comat_copy2d_pair": [_vp, _i64, _vp, _i64, _i64, _vp, _i64, _vp, _i64, _i64, _i64, _i32, _vp],
    "comat_add_rowvec": [_vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "comat_sumpool2x2": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "comat_permute_nchw_nhwc": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "comat_transpose_cast_tiles": [_vp, _vp, _vp, _i64, _i32, _vp],
    "comat_lora_merge": [_vp, _vp, _i64, _f, _vp],
    "comat_cfg_ddpm_fwd": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i32, _vp],
    "comat_cfg_ddpm_bwd": [_vp, _vp, _vp, _i64, _f, _f, _f, _i32, _vp],
    "comat_resample2d": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _vp, _vp, _vp, _vp, _i32, _vp, _vp, _i32,
                         _i32, _vp],
    "comat_patchify": [_vp, _vp, _i32, _i32, _i32, _i32, _i32, _i32, _i32, _vp],
    "comat_embedding": [_vp, _vp, _vp, _i64, _i32, _i64, _i32, _vp],
    "comat_cross_entropy_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _f, _i32, _vp],
    "comat_cross_entropy_bwd": [_vp, _vp, _vp, _vp, _i64, _i32, _i64, _i32, _f, _vp, _vp, _i32, _vp],
    "comat_disc_head_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "comat_disc_head_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64, _i32, _vp],
    "comat_attnmap_gather_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "comat_attnmap_gather_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _vp],
    "comat_sumsq": [_vp, _i64, _vp, _vp, _vp],
    "comat_adamw": [_vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _f, _i32, _vp, _vp, _f, _f, _vp],
    "comat_adamw_tick": [_vp, _vp, _vp],
    "comat_gemm_workspace_bytes": [_i64, _i64, _i64, _i64, _i32],
    "comat_set_option": [C.c_char_p, _i32],
    "comat_last_gemm_kernel": [],
    "comat_flash_attn_fwd_q": [_vp, _vp, _vp, _vp, _vp, _i32, _i32, _i32, _i32, _i32, _i64, _i64, _i64, _i64, _f, _i32, _vp, _i64, _vp, _vp,
                               _vp],
    "comat_fp8_scale": [_vp, _i64, _i32, _vp, _vp, _vp, _vp],
    "comat_fp8_quantize": [_vp, _i64, _i32, _vp, _vp, _vp],
    "comat_fp8_quantize_scaled": [_vp, _i64, _i32, _vp, _vp, _vp, _vp],
    "comat_fp8_scales_update": [_vp, _vp, _i32, _vp],
    "comat_layernorm_fwd_q_ok": [_i32, _i32],
    "comat_layernorm_fwd_q": [_vp, _vp, _vp, _vp, _vp, _i64, _i32, _f, _i32, _vp, _vp, _vp, _vp],
    "comat_groupnorm_fwd_q_ok": [_i32, _i64, _i32, _i32, _i32],
    "comat_groupnorm_fwd_q": [_vp, _vp, _vp, _vp, _vp, _vp, _i32, _i64, _i32, _i32, _f, _i32, _i32, _vp, _vp, _vp, _vp],
}
RESTYPES = {"comat_gemm_workspace_bytes": C.c_int64}
ABI_VERSION = 8
WS_COUNTER_BYTES = 256 * 1024  # COMAT_WS_COUNTER_BYTES: ticket counters at the head of a split-K workspace

_lib = None


def load_library(path: str | None = None):
    """Load libcomat_hip.so (idempotent).  Raises if the library was not built: there is no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    p = path or os.environ.get("COMAT_LIB_PATH") or _LIB_PATH  # COMAT_LIB_PATH: A/B runs of two builds on one box (tools/mb_flash_ab.py)
    if not os.path.exists(p):
        raise RuntimeError(
            f"libcomat_hip.so not found at {p}: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(comat_amd has no CPU fallback)")
    lib = C.CDLL(p)
    lib.comat_abi_version.restype = C.c_int
    if lib.comat_abi_version() != ABI_VERSION:  # checked before anything newer than ABI 1 is bound: a stale build says so
        raise RuntimeError(f"libcomat_hip.so ABI version mismatch: the library at {p} is ABI {lib.comat_abi_version()}, "
                           f"this package binds ABI {ABI_VERSION} - rebuild it (__graft_entry__.build())")
    lib.comat_last_error.restype = C.c_char_p
    lib.comat_build_id.restype = C.c_char_p
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = RESTYPES.get(name, C.c_int)
    _lib = lib
    return lib


GEMM_KERNEL_NAMES = {0: "gemm_kernel / conv_kernel (general 64x64)", 1: "gemm2_kernel (LDS-DMA pipelined)",
                     2: "gemm2_tt_kernel (pipelined, k-major operands)", 3: "gemm2_kernel fp8 (32x32x64 e4m3 MFMA)",
                     4: "gemm2_tt_group_kernel (grouped k-major products)",
                     5: "gemm3_kernel (lean: k-parallel waves, register-direct fragments)"}


def build_id() -> str:
    """hash of the sources the loaded library was built from (include/comat_hip.h: comat_build_id)"""
    load_library()
    return _lib.comat_build_id().decode()


def last_gemm_kernel() -> int:
    """which kernel served this thread's last gemm / gemm_segments / conv2d call (include/comat_hip.h)"""
    load_library()
    return int(_lib.comat_last_gemm_kernel())


def set_option(name: str, value: int):
    """kernel-selection option of the library (include/comat_hip.h: comat_set_option); tests and microbenchmarks only"""
    load_library()
    _check(_lib.comat_set_option(name.encode(), int(value)), "comat_set_option")


def _check(rc: int, name: str):
    if rc != 0:
        msg = _lib.comat_last_error().decode("utf-8", "replace")
        raise RuntimeError(f"{name} failed (rc={rc}): {msg}")


def dt(t: torch.Tensor) -> int:
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.uint8:  # fp8 e4m3 bytes (comat_fp8_quantize): operand dtype of gemm / conv2d only
        return FP8
    raise TypeError(f"comat_amd kernels take float32 or bfloat16 tensors, got {t.dtype}")


def _ptr(t: torch.Tensor | None):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError("comat_amd HIP kernels need tensors resident in HBM (cuda device); got a CPU tensor")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """raw handle of torch's current HIP stream on the current device (every kernel is enqueued on it).  The C-level
    accessors avoid building a torch.cuda.Stream object per launch (~2 us of the ~8 us host cost of a launch)."""
    if _raw_stream is not None and _get_device is not None:
        return _raw_stream(_get_device())
    return torch.cuda.current_stream().cuda_stream


class HipKernels:
    """Thin, 1:1 wrappers over the C ABI.  Every method enqueues on the current torch HIP stream."""

    name = "hip"

    # split-K workspace per (device, stream): [ticket counters (zeroed once, re-armed by the kernels) | fp32 slabs].
    # comat_gemm_workspace_bytes() of the largest split problem of this workload (SDXL 16x16-level convs) is < 200 MB.
    WS_BYTES = 256 << 20

    def __init__(self):
        load_library()
        self._ws = {}

    @staticmethod
    def _no_capture(what):
        """Workspaces carry ticket counters that must be zero before their FIRST use and are re-armed by the kernels.  A
        workspace created while its stream is capturing would be zeroed by a node of THAT graph only: a second graph
        captured later on the same stream, replayed first, would run on uninitialised counters.  So creation during a
        capture is an error: `prepare_stream()` creates them eagerly (ops.prepare_capture_stream does it for the package's
        capture stream and its side stream)."""
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError(f"{what} of a capturing stream must exist before the capture begins: call "
                               "HipKernels.prepare_stream() on that stream first (comat_amd.ops.prepare_capture_stream)")

    def _workspace(self, dev):
        key = (dev, _stream())  # one slab workspace per stream: kernels on different streams may overlap
        ws = self._ws.get(key)
        if ws is None:
            self._no_capture("the split-K workspace")
            ws = self._ws[key] = torch.empty(self.WS_BYTES // 4, dtype=torch.float32, device=dev)
            ws[: WS_COUNTER_BYTES // 4].zero_()  # the ABI asks for zeroed ticket counters before the first use
        return ws

    GN_WS_MIN = 1 << 22

    def prepare_stream(self, dev, gn_groups=64):
        """create (and zero) every per-stream workspace of the CURRENT stream now, outside any capture; idempotent.
        -> True if something had to be created (the caller then synchronises before it starts a capture)"""
        dev = torch.device(dev)
        if dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        key = ("prepared", dev, _stream())
        if key in self._ws:
            return False
        self._no_capture("workspaces")
        self._workspace(dev)
        self._gn_workspace(dev, gn_groups, 1)
        self._fp8_workspace(dev)
        self._scratch(dev, 1 << 20)
        self._ws[key] = True
        return True

    def gemm_workspace_bytes(self, M, N, K, batch=1, dtype=torch.bfloat16):
        return int(_lib.comat_gemm_workspace_bytes(M, N, K, batch, BF16 if dtype == torch.bfloat16 else F32))

    # ---- contraction ---------------------------------------------------------------------------------------
    def gemm(self, A, B, Cout, M, N, K, lda, ldb, ldc, transA=False, transB=False, batch=(1, 1),
             sA=(0, 0), sB=(0, 0), sC=(0, 0), bias=None, bias2=None, rows_per_bias2=0, R=None, ldr=0,
             sR=(0, 0), alpha=1.0, beta=0.0, act=ACT_NONE, scales=None, geglu=None, tail=None, ktail=None, q8=None):
        """q8 = (bytes [M, N / 2] uint8, scale [1], amax [1] int32): the GEGLU epilogue also emits the e4m3 bytes of its product under
        `scale` and tracks its abs-max (delayed fp8 scaling); geglu[0] may then be None (no bf16 copy);
        ktail = (A2 [M, K2], B2 [N, K2], K2, lda2, ldb2, sA2, sB2): bf16 k-tail added to the scaled fp8 product in the same launch;
        scales = (scale_a, scale_b[, s_scale_b]): fp32 device scalars of fp8 (uint8) operands A and B (include/comat_hip.h; batch z of a
        batched product uses scale_b[z * s_scale_b]);
        geglu = (C2 [M, N / 2], keep_pre): the GEGLU epilogue over interleaved value / gate columns (comat_gemm_params::epi2):
        C2 receives value * gelu(gate); Cout receives the pre-activations only when keep_pre (it may be None otherwise);
        geglu = (pre [M, 2 N], "bwd"): the GEGLU backward epilogue (epi2 = 3): Cout [M, 2 N] = gradient of the pre-activations;
        tail = (B2 [n2, K], C2, n2, ldc2, sB2, sC2, alpha2): tail columns (epi2 = 4): N counts the n2 extra columns, whose rows of
        B come from B2 (batch stride sB2) and which are written to C2 (leading dimension ldc2, batch stride sC2) as alpha2 * A B2^T"""
        p = GemmParams()
        if tail is not None:
            assert geglu is None and not transA and not transB
            B2, C2t, n2, ldc2, sB2t, sC2t, alpha2 = tail
            assert B2.dtype == B.dtype and C2t.dtype == Cout.dtype
            p.B2, p.C2, p.n2, p.ldc2, p.sB2_tail, p.sC2_tail, p.alpha2, p.epi2 = _ptr(B2), _ptr(C2t), n2, ldc2, sB2t, sC2t, alpha2, 4
        if geglu is not None and geglu[1] == "bwd":  # epi2 = 3: geglu[0] = the saved pre-activations [M, 2 N], Cout [M, 2 N] their gradient
            p.C2, p.ldc2, p.epi2 = _ptr(geglu[0]), geglu[0].shape[1], 3
            assert geglu[0].dtype == torch.bfloat16 and geglu[0].is_contiguous() and Cout.dtype == torch.bfloat16
        elif geglu is not None:
            p.epi2 = 1 if geglu[1] else 2
            if geglu[0] is not None:
                p.C2, p.ldc2 = _ptr(geglu[0]), geglu[0].shape[1]
                assert geglu[0].dtype == torch.bfloat16 and geglu[0].is_contiguous()
            else:
                assert q8 is not None
            if q8 is not None:
                assert q8[0].dtype == torch.uint8 and q8[0].is_contiguous()
                p.q8, p.q_scale, p.q_amax, p.ldq8 = _ptr(q8[0]), _ptr(q8[1]), _ptr(q8[2]), q8[0].shape[1]
        p.A, p.B, p.C = _ptr(A), _ptr(B), _ptr(Cout)
        p.bias, p.bias2, p.R = _ptr(bias), _ptr(bias2), _ptr(R)
        if bias is not None:
            assert bias.dtype == torch.float32
        if bias2 is not None:
            assert bias2.dtype == torch.float32
        p.M, p.N, p.K = M, N, K
        p.lda, p.ldb, p.ldc, p.ldr = lda, ldb, ldc, ldr
        p.batch1, p.batch2 = batch
        p.sA1, p.sA2 = sA
        p.sB1, p.sB2 = sB
        p.sC1, p.sC2 = sC
        p.sR1, p.sR2 = sR
        p.rows_per_bias2 = rows_per_bias2
        p.alpha, p.beta = alpha, beta
        p.transA, p.transB, p.act = int(transA), int(transB), act
        assert A.dtype == B.dtype
        p.in_dtype, p.out_dtype = dt(A), (dt(Cout) if Cout is not None else BF16)
        assert (scales is not None) == (p.in_dtype == FP8), "fp8 operands come with their scales"
        if scales is not None:  # (scale_a, scale_b[, batch stride of scale_b])
            p.scale_a, p.scale_b = _ptr(scales[0]), _ptr(scales[1])
            p.s_scale_b = scales[2] if len(scales) > 2 else 0
        if ktail is not None:
            assert p.in_dtype == FP8 and ktail[0].dtype == torch.bfloat16 and ktail[1].dtype == torch.bfloat16
            p.A2k, p.B2k = _ptr(ktail[0]), _ptr(ktail[1])
            p.K2, p.lda2k, p.ldb2k, p.sA2k, p.sB2k = ktail[2:]
        p.r_dtype = dt(R) if R is not None else 0
        ws = self._workspace(A.device)
        p.ws, p.ws_bytes = ws.data_ptr(), self.WS_BYTES
        _check(_lib.comat_gemm(C.byref(p), _stream()), "comat_gemm")

    @staticmethod
    def geglu_gemm_ok(x, w, M, N, K):
        """problems the GEGLU epilogue of comat_gemm takes (the pipelined kernel's bf16 / fp8 shapes, whole 32-column tiles)"""
        return (x.dtype in (torch.bfloat16, torch.uint8) and M >= 16 and N % 32 == 0 and K % (32 if x.dtype == torch.bfloat16 else 64) == 0
                and x.data_ptr() % 16 == 0 and w.data_ptr() % 16 == 0)

    def _segments_params(self, segs, Cout, M, N, ldc, bias=None, R=None, ldr=0, alpha=1.0, beta=0.0, batch=1, sC=0, sR=0):
        n = len(segs)
        arr = (GemmSegment * n)()
        for i, sg in enumerate(segs):
            A, B, K, lda, ldb = sg[:5]
            assert A.dtype == B.dtype == segs[0][0].dtype
            arr[i].A, arr[i].B, arr[i].K, arr[i].lda, arr[i].ldb = _ptr(A), _ptr(B), K, lda, ldb
            arr[i].sA, arr[i].sB = (sg[5], sg[6]) if len(sg) > 5 else (0, 0)
        p = GemmParams()
        p.C, p.bias, p.R = _ptr(Cout), _ptr(bias), _ptr(R)
        if bias is not None:
            assert bias.dtype == torch.float32
        p.M, p.N, p.ldc, p.ldr = M, N, ldc, ldr
        p.batch1, p.batch2 = batch, 1
        p.sC1, p.sR1 = sC, sR
        p.alpha, p.beta = alpha, beta
        p.in_dtype, p.out_dtype = dt(segs[0][0]), dt(Cout)
        p.r_dtype = dt(R) if R is not None else 0
        ws = self._workspace(Cout.device)
        p.ws, p.ws_bytes = ws.data_ptr(), self.WS_BYTES
        return p, arr, n

    def gemm_segments(self, segs, Cout, M, N, ldc, bias=None, R=None, ldr=0, alpha=1.0, beta=0.0, batch=1, sC=0, sR=0):
        """Cout[M, N] = alpha * sum_s A_s[M, K_s] B_s[N, K_s]^T + bias + beta * R;  segs: [(A, B, K, lda, ldb[, sA, sB])].
        batch > 1: that many problems in one launch; operands advance by sA / sB, Cout / R by sC / sR elements."""
        p, arr, n = self._segments_params(segs, Cout, M, N, ldc, bias, R, ldr, alpha, beta, batch, sC, sR)
        _check(_lib.comat_gemm_segments(C.byref(p), arr, n, _stream()), "comat_gemm_segments")

    @staticmethod
    def tt_group_ok(A, B, Cacc, M, N, K, lda, ldb, ldc):
        """what comat_gemm_tt_grouped takes (include/comat_hip.h); anything else goes through gemm()"""
        return (A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and Cacc.dtype == torch.float32
                and M >= 8 and N >= 8 and K >= 1 and M % 8 == 0 and N % 8 == 0 and lda % 8 == 0 and ldb % 8 == 0
                and ldc % 4 == 0 and A.data_ptr() % 16 == 0 and B.data_ptr() % 16 == 0 and Cacc.data_ptr() % 16 == 0)

    def gemm_tt_grouped(self, problems):
        """C_p[M, N] (fp32) += A_p^T B_p for independent problems [(A [K, M] ld lda, B [K, N] ld ldb, C, M, N, K, lda, ldb,
        ldc)] in as few launches as possible (<= 48 problems each); the C_p must not overlap."""
        n = len(problems)
        arr = (TTProblem * n)()
        for i, (A, B, Cacc, M, N, K, lda, ldb, ldc) in enumerate(problems):
            assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and Cacc.dtype == torch.float32
            e = arr[i]
            e.A, e.B, e.C, e.M, e.N, e.K, e.lda, e.ldb, e.ldc = _ptr(A), _ptr(B), _ptr(Cacc), M, N, K, lda, ldb, ldc
        ws = self._workspace(problems[0][0].device)
        _check(_lib.comat_gemm_tt_grouped(arr, n, BF16, ws.data_ptr(), self.WS_BYTES, _stream()), "comat_gemm_tt_grouped")

    @staticmethod
    def lora_merge_ok(W, U, Dt, r, ldu, lddt):
        """what comat_lora_merge takes (include/comat_hip.h); anything else is merged by one comat_gemm per group"""
        N, K = W.shape
        return (W.dtype == torch.bfloat16 and U.dtype == torch.bfloat16 and Dt.dtype == torch.bfloat16 and W.is_contiguous()
                and N % 8 == 0 and K % 8 == 0 and r % 16 == 0 and ldu % 8 == 0 and lddt % 8 == 0
                and all(t.data_ptr() % 16 == 0 for t in (W, U, Dt)))

    def lora_merge(self, problems, tiles, scale):
        """problems: device int64 [n, 10], tiles: device int32 [n_tiles, 3] (include/comat_hip.h: comat_lora_merge)"""
        assert problems.dtype == torch.int64 and problems.shape[1] == 10 and tiles.dtype == torch.int32 and tiles.shape[1] == 3
        _check(_lib.comat_lora_merge(_ptr(problems), _ptr(tiles), tiles.shape[0], float(scale), _stream()), "comat_lora_merge")

    def transpose_cast_tiles(self, src, dst, tiles):
        assert src.dtype == torch.float32 and tiles.dtype == torch.int64 and tiles.shape[1] == 6
        _check(_lib.comat_transpose_cast_tiles(_ptr(src), _ptr(dst), _ptr(tiles), tiles.shape[0], dt(dst), _stream()),
               "comat_transpose_cast_tiles")

    def conv2d(self, X, W, Y, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, mode=0, ups=1, bias=None,
               bias2=None, R=None, alpha=1.0, beta=0.0, act=ACT_NONE, scales=None):
        p = ConvParams()
        p.X, p.W, p.Y = _ptr(X), _ptr(W), _ptr(Y)
        p.bias, p.bias2, p.R = _ptr(bias), _ptr(bias2), _ptr(R)
        if bias is not None:
            assert bias.dtype == torch.float32
        if bias2 is not None:
            assert bias2.dtype == torch.float32
        p.B, p.Hin, p.Win, p.Cin, p.Hout, p.Wout, p.Cout = B, Hin, Win, Cin, Hout, Wout, Cout
        p.KH, p.KW, p.stride, p.pad, p.mode, p.ups = KH, KW, stride, pad, mode, ups
        p.alpha, p.beta, p.act = alpha, beta, act
        assert X.dtype == W.dtype
        p.in_dtype, p.out_dtype = dt(X), dt(Y)
        assert (scales is not None) == (p.in_dtype == FP8), "fp8 operands come with their scales"
        if scales is not None:
            p.scale_a, p.scale_b = _ptr(scales[0]), _ptr(scales[1])
        p.r_dtype = dt(R) if R is not None else 0
        ws = self._workspace(X.device)
        p.ws, p.ws_bytes = ws.data_ptr(), self.WS_BYTES
        _check(_lib.comat_conv2d(C.byref(p), _stream()), "comat_conv2d")

    # ---- fp8 operands ---------------------------------------------------------------------------------------
    def fp8_quantize(self, x, out=None, scale=None, amax=None):
        """per-tensor e4m3 quantisation of a contiguous fp32 / bf16 tensor: -> (bytes [same shape] uint8, scale [1] fp32);
        value = scale * fp8.  Two launches (abs-max -> scale, then the bytes).  amax: [1] int32 = a delayed-scaling site's running
        maximum (float bits), which receives this tensor's abs-max too (calibration pass)."""
        assert x.is_contiguous()
        n = x.numel()
        ws = self._fp8_workspace(x.device)
        if scale is None:
            scale = torch.empty(1, dtype=torch.float32, device=x.device)
        if out is None:
            out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        _check(_lib.comat_fp8_scale(_ptr(x), n, dt(x), _ptr(scale), _ptr(ws), _ptr(amax), _stream()), "comat_fp8_scale")
        _check(_lib.comat_fp8_quantize(_ptr(x), n, dt(x), _ptr(scale), _ptr(out), _stream()), "comat_fp8_quantize")
        return out, scale

    def fp8_quantize_scaled(self, x, scale, amax, out=None):
        """delayed scaling: the bytes of x under the scale already in `scale` [1]; max |x| folded into `amax` [1] int32 (float bits).
        One launch."""
        assert x.is_contiguous()
        if out is None:
            out = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
        _check(_lib.comat_fp8_quantize_scaled(_ptr(x), x.numel(), dt(x), _ptr(scale), _ptr(out), _ptr(amax), _stream()),
               "comat_fp8_quantize_scaled")
        return out

    def fp8_scales_update(self, amax, scale, n):
        """once per optimizer step: scale[i] = max(amax[i], 2^-100) / 448 and amax[i] = 0 for every site i < n that saw a tensor"""
        _check(_lib.comat_fp8_scales_update(_ptr(amax), _ptr(scale), int(n), _stream()), "comat_fp8_scales_update")

    def layernorm_fwd_q_ok(self, x):
        return bool(_lib.comat_layernorm_fwd_q_ok(int(x.shape[1]), dt(x))) and x.data_ptr() % 16 == 0

    def layernorm_fwd_q(self, x, gamma, beta, y, stats, M, Cc, eps, q8, scale, amax):
        _check(_lib.comat_layernorm_fwd_q(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), M, Cc, eps, dt(x), _ptr(q8),
                                          _ptr(scale), _ptr(amax), _stream()), "comat_layernorm_fwd_q")

    def groupnorm_fwd_q_ok(self, x, B, HW, Cc, G):
        return bool(_lib.comat_groupnorm_fwd_q_ok(B, HW, Cc, G, dt(x))) and x.data_ptr() % 16 == 0

    def groupnorm_fwd_q(self, x, gamma, beta, y, stats, B, HW, Cc, G, eps, silu, q8, scale, amax):
        ws = self._gn_workspace(x.device, B, G)
        _check(_lib.comat_groupnorm_fwd_q(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), _ptr(ws), B, HW, Cc, G, eps,
                                          int(silu), dt(x), _ptr(q8), _ptr(scale), _ptr(amax), _stream()), "comat_groupnorm_fwd_q")

    def _fp8_workspace(self, dev):
        key = ("fp8", dev, _stream())
        ws = self._ws.get(key)
        if ws is None:
            self._no_capture("the fp8 scale workspace")
            ws = self._ws[key] = torch.zeros(2, dtype=torch.int32, device=dev)  # ticket + running max, re-armed
        return ws

    # ---- normalisation -------------------------------------------------------------------------------------
    def _gn_workspace(self, dev, B, G):
        """GroupNorm workspace of the current stream: ticket counters (zeroed here once, re-armed by the kernels) + the
        per-block partial sums (include/comat_hip.h: COMAT_GN_WS_DOUBLES)"""
        need = 512 + B * G * 2 * 1025
        key = ("gn", dev, _stream())
        ws = self._ws.get(key)
        if ws is None or ws.numel() < need:
            # GN_WS_MIN doubles (32 MiB) hold B * G <= 2045 (sample, group) pairs - a per-GPU batch of 31 under CFG doubling
            # with 32 groups - so that a stream prepared for capture (prepare_stream) never has to grow inside one: only
            # the default stream's workspace is ever warmed up by eager calls, the capture stream's is created before any
            # launch ran on it (the former 2 MiB floor made every capture with B >= 4 raise)
            self._no_capture("the GroupNorm workspace")
            ws = self._ws[key] = torch.zeros(max(need, self.GN_WS_MIN), dtype=torch.float64, device=dev)
        return ws

    def groupnorm_fwd(self, x, gamma, beta, y, stats, B, HW, Cc, G, eps, silu):
        ws = self._gn_workspace(x.device, B, G)
        _check(_lib.comat_groupnorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), _ptr(ws), B, HW, Cc,
                                        G, eps, int(silu), dt(x), _stream()), "comat_groupnorm_fwd")

    def groupnorm_bwd(self, dy, x, gamma, beta, stats, dx, B, HW, Cc, G, silu, add=None):
        ws = self._gn_workspace(x.device, B, G)
        _check(_lib.comat_groupnorm_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(beta), _ptr(stats), _ptr(dx), _ptr(ws),
                                        B, HW, Cc, G, int(silu), _ptr(add), dt(x), _stream()), "comat_groupnorm_bwd")

    def layernorm_fwd(self, x, gamma, beta, y, stats, M, Cc, eps):
        _check(_lib.comat_layernorm_fwd(_ptr(x), _ptr(gamma), _ptr(beta), _ptr(y), _ptr(stats), M, Cc, eps, dt(x),
                                        _stream()), "comat_layernorm_fwd")

    def layernorm_bwd(self, dy, x, gamma, stats, dx, M, Cc, add=None):
        _check(_lib.comat_layernorm_bwd(_ptr(dy), _ptr(x), _ptr(gamma), _ptr(stats), _ptr(dx), M, Cc, _ptr(add), dt(x),
                                        _stream()), "comat_layernorm_bwd")

    # ---- softmax -------------------------------------------------------------------------------------------
    def softmax_fwd(self, S, P, rows, cols, q_len=0, causal=False, causal_offset=0, key_mask=None, rows_per_mask=0):
        _check(_lib.comat_softmax_fwd(_ptr(S), _ptr(P), rows, cols, q_len, int(causal), causal_offset,
                                      _ptr(key_mask), rows_per_mask, dt(S), dt(P), _stream()), "comat_softmax_fwd")

    def softmax_bwd(self, P, dP, dS, rows, cols, scale):
        _check(_lib.comat_softmax_bwd(_ptr(P), _ptr(dP), _ptr(dS), rows, cols, scale, dt(P), dt(dP), dt(dS),
                                      _stream()), "comat_softmax_bwd")

    def flash_attn_fwd(self, q, k, v, o, lse, B, H, Nq, Nk, d, ldq, ldk, ldv, ldo, scale, q8=None):
        """q8 = (bytes [B*Nq, H*d] uint8, scale [1], amax [1] int32): the forward also emits the e4m3 bytes of its output (delayed fp8 scaling)"""
        if q8 is not None:
            _check(_lib.comat_flash_attn_fwd_q(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), B, H, Nq, Nk, d, ldq, ldk, ldv, ldo, scale,
                                               dt(q), _ptr(q8[0]), q8[0].shape[1], _ptr(q8[1]), _ptr(q8[2]), _stream()),
                   "comat_flash_attn_fwd_q")
            return
        _check(_lib.comat_flash_attn_fwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(lse), B, H, Nq, Nk, d, ldq, ldk, ldv,
                                         ldo, scale, dt(q), _stream()), "comat_flash_attn_fwd")

    def flash_attn_bwd(self, q, k, v, o, do, lse, dbuf, dq, dk, dv, B, H, Nq, Nk, d, ldq, ldk, ldv, ldo, scale):
        # the fp32 partials of the split dK/dV pass live BEHIND the GEMMs' ticket counters (the head of the workspace
        # must stay zero: include/comat_hip.h, COMAT_WS_COUNTER_BYTES)
        ws = self._workspace(q.device)
        _check(_lib.comat_flash_attn_bwd(_ptr(q), _ptr(k), _ptr(v), _ptr(o), _ptr(do), _ptr(lse), _ptr(dbuf), _ptr(dq),
                                         _ptr(dk), _ptr(dv), B, H, Nq, Nk, d, ldq, ldk, ldv, ldo, scale, dt(q),
                                         ws.data_ptr() + WS_COUNTER_BYTES, self.WS_BYTES - WS_COUNTER_BYTES, _stream()),
               "comat_flash_attn_bwd")

    # ---- elementwise ---------------------------------------------------------------------------------------
    def unary(self, op, x, y, n, p0=0.0, p1=0.0):
        _check(_lib.comat_unary(op, _ptr(x), _ptr(y), n, p0, p1, dt(x), dt(y), _stream()), "comat_unary")

    def unary_bwd(self, op, dy, x, dx, n):
        _check(_lib.comat_unary_bwd(op, _ptr(dy), _ptr(x), _ptr(dx), n, dt(x), _stream()), "comat_unary_bwd")

    def axpby(self, a, x, b, y, out, n):
        _check(_lib.comat_axpby(a, _ptr(x), b, _ptr(y), _ptr(out), n, dt(x), dt(y) if y is not None else 0, dt(out),
                                _stream()), "comat_axpby")

    def geglu_fwd(self, x, y, M, D):
        _check(_lib.comat_geglu_fwd(_ptr(x), _ptr(y), M, D, dt(x), _stream()), "comat_geglu_fwd")

    def geglu_bwd(self, dy, x, dx, M, D):
        _check(_lib.comat_geglu_bwd(_ptr(dy), _ptr(x), _ptr(dx), M, D, dt(x), _stream()), "comat_geglu_bwd")

    def geglu_il_fwd(self, x, y, M, D):
        _check(_lib.comat_geglu_il_fwd(_ptr(x), _ptr(y), M, D, dt(x), _stream()), "comat_geglu_il_fwd")

    def geglu_il_bwd(self, dy, x, dx, M, D):
        _check(_lib.comat_geglu_il_bwd(_ptr(dy), _ptr(x), _ptr(dx), M, D, dt(x), _stream()), "comat_geglu_il_bwd")

    def copy2d(self, src, ld_src, dst, ld_dst, rows, cols):
        _check(_lib.comat_copy2d(_ptr(src), ld_src, _ptr(dst), ld_dst, rows, cols, dt(src), dt(dst), _stream()),
               "comat_copy2d")

    @staticmethod
    def copy2d_pair_ok(items):
        """what comat_copy2d_pair takes: one dtype, whole 16-byte vectors, aligned pointers; items = [(src, ld_src, dst, ld_dst, cols)] x 2"""
        d0 = items[0][0].dtype
        epv = 16 // items[0][0].element_size()
        return (d0 in (torch.float32, torch.bfloat16)
                and all(s.dtype == d0 and d.dtype == d0 and c % epv == 0 and ls % epv == 0 and ld % epv == 0
                        and s.data_ptr() % 16 == 0 and d.data_ptr() % 16 == 0 for s, ls, d, ld, c in items))

    def copy2d_pair(self, items, rows):
        (s0, ls0, d0, ld0, c0), (s1, ls1, d1, ld1, c1) = items
        _check(_lib.comat_copy2d_pair(_ptr(s0), ls0, _ptr(d0), ld0, c0, _ptr(s1), ls1, _ptr(d1), ld1, c1, rows, dt(s0), _stream()),
               "comat_copy2d_pair")

    def add_rowvec(self, x, v, out, rows, cols):
        _check(_lib.comat_add_rowvec(_ptr(x), _ptr(v), _ptr(out), rows, cols, dt(x), _stream()), "comat_add_rowvec")

    def sumpool2x2(self, x, y, B, H, W, Cc):
        _check(_lib.comat_sumpool2x2(_ptr(x), _ptr(y), B, H, W, Cc, dt(x), _stream()), "comat_sumpool2x2")

    def permute_nchw_nhwc(self, x, y, B, Cc, H, W, to_nhwc):
        _check(_lib.comat_permute_nchw_nhwc(_ptr(x), _ptr(y), B, Cc, H, W, int(to_nhwc), dt(x), dt(y), _stream()),
               "comat_permute_nchw_nhwc")

    def cfg_ddpm_fwd(self, x, eps2, z, x_prev, n, s, cx, ce, sigma):
        _check(_lib.comat_cfg_ddpm_fwd(_ptr(x), _ptr(eps2), _ptr(z), _ptr(x_prev), n, s, cx, ce, sigma, dt(eps2),
                                       _stream()), "comat_cfg_ddpm_fwd")

    def cfg_ddpm_bwd(self, g, dx, deps2, n, s, cx, ce):
        _check(_lib.comat_cfg_ddpm_bwd(_ptr(g), _ptr(dx), _ptr(deps2), n, s, cx, ce, dt(deps2), _stream()),
               "comat_cfg_ddpm_bwd")

    # ---- image path ----------------------------------------------------------------------------------------
    def resample2d(self, src, out, B, Hin, Win, Hout, Wout, Cc, ystart, ywt, xstart, xwt, KT, scale, shift):
        _check(_lib.comat_resample2d(_ptr(src), _ptr(out), B, Hin, Win, Hout, Wout, Cc, _ptr(ystart), _ptr(ywt),
                                     _ptr(xstart), _ptr(xwt), KT, _ptr(scale), _ptr(shift), dt(src), dt(out),
                                     _stream()), "comat_resample2d")

    def patchify(self, img, patches, B, H, W, Cc, P, inverse):
        _check(_lib.comat_patchify(_ptr(img), _ptr(patches), B, H, W, Cc, P, int(inverse), dt(img), _stream()),
               "comat_patchify")

    def embedding(self, ids, table, out, n, dim, vocab):
        assert ids.dtype == torch.int64
        _check(_lib.comat_embedding(_ptr(ids), _ptr(table), _ptr(out), n, dim, vocab, dt(table), _stream()),
               "comat_embedding")

    # ---- losses --------------------------------------------------------------------------------------------
    def _scratch(self, dev, n):
        """small fp32 scratch for the two-stage (fixed-order) reductions; one per (device, stream)"""
        key = ("scratch", dev, _stream())
        t = self._ws.get(key)
        if t is None or t.numel() < n:
            # (no counters in here: a scratch buffer allocated during a capture is merely graph-private memory, but one
            # that OUTLIVES the graph's pool bookkeeping - keep it out of captures as well)
            self._no_capture("the reduction scratch buffer")
            t = self._ws[key] = torch.empty(max(n, 16384), dtype=torch.float32, device=dev)
        return t

    def cross_entropy_fwd(self, logits, labels, logp, row_lse, loss_sum_cnt, T, V, ld, ignore_index, ls):
        assert labels.dtype == torch.int64
        ws = self._scratch(logits.device, T)
        _check(_lib.comat_cross_entropy_fwd(_ptr(logits), _ptr(labels), _ptr(logp), _ptr(row_lse), _ptr(ws),
                                            _ptr(loss_sum_cnt), T, V, ld, ignore_index, ls, dt(logits), _stream()),
               "comat_cross_entropy_fwd")

    def cross_entropy_bwd(self, logits, labels, row_lse, dlogits, T, V, ld, ignore_index, ls, g_up, loss_sum_cnt):
        _check(_lib.comat_cross_entropy_bwd(_ptr(logits), _ptr(labels), _ptr(row_lse), _ptr(dlogits), T, V, ld,
                                            ignore_index, ls, _ptr(g_up), _ptr(loss_sum_cnt), dt(logits), _stream()),
               "comat_cross_entropy_bwd")

    def disc_head_fwd(self, x, w, b, target, loss, P, pix_per_sample):
        ws = self._scratch(x.device, 512)
        _check(_lib.comat_disc_head_fwd(_ptr(x), _ptr(w), _ptr(b), _ptr(target), _ptr(loss), _ptr(ws), P,
                                        pix_per_sample, dt(x), _stream()), "comat_disc_head_fwd")

    def disc_head_bwd(self, x, w, b, target, g_up, dx, dwb, P, pix_per_sample):
        """dwb: fp32 [5] = (dw[0..3], db), accumulated; or None"""
        ws = self._scratch(x.device, 512 * 5)
        _check(_lib.comat_disc_head_bwd(_ptr(x), _ptr(w), _ptr(b), _ptr(target), _ptr(g_up), _ptr(dx), _ptr(dwb),
                                        _ptr(ws), P, pix_per_sample, dt(x), _stream()), "comat_disc_head_bwd")

    def attnmap_gather_fwd(self, amap, mask, tok_idx, tok_obj, num, den, avg, heads, npix, L, n_tok):
        assert tok_idx.dtype == torch.int32 and tok_obj.dtype == torch.int32
        ws = self._scratch(amap.device, (((npix + 127) // 128) * 2 + npix) * heads * n_tok)
        _check(_lib.comat_attnmap_gather_fwd(_ptr(amap), _ptr(mask), _ptr(tok_idx), _ptr(tok_obj), _ptr(num),
                                             _ptr(den), _ptr(avg), _ptr(ws), heads, npix, L, n_tok, dt(amap),
                                             _stream()), "comat_attnmap_gather_fwd")

    def attnmap_gather_bwd(self, g_num, g_den, g_avg, mask, tok_idx, tok_obj, damap, heads, npix, L, n_tok):
        _check(_lib.comat_attnmap_gather_bwd(_ptr(g_num), _ptr(g_den), _ptr(g_avg), _ptr(mask), _ptr(tok_idx),
                                             _ptr(tok_obj), _ptr(damap), heads, npix, L, n_tok, dt(damap),
                                             _stream()), "comat_attnmap_gather_bwd")

    # ---- optimizer -----------------------------------------------------------------------------------------
    def sumsq(self, x, n, out):
        ws = self._scratch(x.device, 1024)
        _check(_lib.comat_sumsq(_ptr(x), n, _ptr(out), _ptr(ws), _stream()), "comat_sumsq")

    def adamw(self, p, g, m, v, n, lr, beta1, beta2, eps, wd, step, gnorm_sq, max_norm, step_dev=None, grad_scale=1.0):
        """step_dev: int32 [2] device counters (applied, skipped) or None; with it the bias correction uses
        step_dev[0] + 1 and `step` is ignored (see adamw_tick); grad_scale: 1 / world when g holds the ranks' SUM"""
        _check(_lib.comat_adamw(_ptr(p), _ptr(g), _ptr(m), _ptr(v), n, lr, beta1, beta2, eps, wd, step,
                                _ptr(step_dev), _ptr(gnorm_sq), max_norm, grad_scale, _stream()), "comat_adamw")

    def adamw_tick(self, counters, gnorm_sq):
        assert counters.dtype == torch.int32 and counters.numel() >= 2
        _check(_lib.comat_adamw_tick(_ptr(counters), _ptr(gnorm_sq), _stream()), "comat_adamw_tick")
