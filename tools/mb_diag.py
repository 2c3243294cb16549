"""Which part of a k-tile bounds the pipelined GEMM / conv kernel?  Times a few large problems under forced block shapes with the
library named by COMAT_LIB_PATH - the product build or one of the `make diag` builds (gemm2.hip: G2_DIAG = 1 no MFMAs, 2 no fragment
reads, 3 no LDS-DMA, 4 no barrier; their outputs are garbage by construction).  tools/calls/r6_k.sh runs it once per build on one box.
    COMAT_LIB_PATH=comat_amd/lib/libcomat_hip_d1.so python tools/mb_diag.py"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from comat_amd import _hip, ops  # noqa: E402
from tools.mb_big import timeit  # noqa: E402

CONVS = [(1, 256, 256, 512, 256, 1), (1, 256, 256, 256, 256, 1), (1, 128, 128, 512, 512, 1), (2, 64, 64, 320, 320, 1)]
GEMMS = [(8192, 5120, 640), (8192, 1280, 1280), (4096, 4096, 4096)]
CFGS = [int(c) for c in os.environ.get("MB_CFGS", "1,12").split(",")]
if os.environ.get("MB_ONLY") == "kscan":  # time against K at fixed M, N: slope = a k-tile, intercept = prologue + epilogue
    CONVS, GEMMS = [], [(4096, 4096, k) for k in (64, 512, 1024, 2048, 4096, 8192)]
if os.environ.get("MB_ONLY") == "unet":  # the UNet's row counts: intercept (launch + prologue + epilogue) and slope per block shape
    CONVS = []
    GEMMS = [(m, n, k) for (m, n) in ((8192, 320), (2048, 640), (512, 1280), (8192, 2560)) for k in (64, 320, 1280)]
if os.environ.get("MB_ONLY") == "big":  # counter passes: one conv, one GEMM
    CONVS, GEMMS = CONVS[:1], GEMMS[2:]


def main():
    dev = torch.device("cuda:0")
    k = _hip.HipKernels()
    ops.set_kernel_backend(k)
    T = torch.bfloat16
    if os.environ.get("MB_ORDER"):
        _hip.set_option("g2_order", int(os.environ["MB_ORDER"]))
    tag = os.environ.get("MB_ORDER", "") + " " + os.path.basename(os.environ.get("COMAT_LIB_PATH", "libcomat_hip.so"))
    r = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).to(T)
    for (B, H, W, Cin, Cout, ups) in CONVS:
        x, w = r(B * H * W, Cin), r(Cout, 3, 3, Cin) * (9 * Cin) ** -0.5
        y = torch.empty((B * H * W, Cout), dtype=T, device=dev)
        fn = lambda: k.conv2d(x, w, y, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, mode=0, ups=ups)
        for c in CFGS:
            _hip.set_option("g2_cfg", c)
            _hip.set_option("g2_splits", 1)
            us = timeit(fn)
            print(f"{tag:24s} conv {B}x{H}x{W} {Cin}->{Cout}  cfg {c:2d}  {us:8.1f} us  {2.0 * B * H * W * Cout * 9 * Cin / us / 1e6:7.1f} TF/s", flush=True)
    for (M, N, K) in GEMMS:
        a, b = r(M, K), r(N, K) * K ** -0.5
        c_ = torch.empty((M, N), dtype=T, device=dev)
        fn = lambda: k.gemm(a, b, c_, M, N, K, K, K, N)
        for c in CFGS:
            _hip.set_option("g2_cfg", c)
            _hip.set_option("g2_splits", 1)
            us = timeit(fn)
            print(f"{tag:24s} gemm {M}x{N}x{K}  cfg {c:2d}  {us:8.1f} us  {2.0 * M * N * K / us / 1e6:7.1f} TF/s", flush=True)


if __name__ == "__main__":
    main()
