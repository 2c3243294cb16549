"""Per-shape microbenchmark of comat_gemm / comat_conv2d on the GPU (HIP events around N back-to-back launches).
Tuning knobs (read once per process by the library): COMAT_FORCE_TILE=64|128|12864|64128 (block tile 64x64, 128x128,
128x64, 64x128), COMAT_FORCE_SPLITS=n.  Sweep:  for t in 64 12864 64128 128; do COMAT_FORCE_TILE=$t python
tools/microbench_gemm.py; done"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import _hip  # noqa: E402

K = _hip.HipKernels()
dev = torch.device("cuda:0")
T = torch.bfloat16


def time_it(fn, n=int(os.environ.get("MB_N", "50"))):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def gemm(M, N, Kd, tA=False, tB=False, batch=(1, 1), out=T):
    b = batch[0] * batch[1]
    A = torch.randn((b, Kd, M) if tA else (b, M, Kd), device=dev).to(T)
    B = torch.randn((b, Kd, N) if tB else (b, N, Kd), device=dev).to(T)
    C = torch.empty((b, M, N), device=dev, dtype=out)
    f = lambda: K.gemm(A, B, C, M, N, Kd, M if tA else Kd, N if tB else Kd, N, transA=tA, transB=tB, batch=batch,
                       sA=(batch[1] * M * Kd, M * Kd), sB=(batch[1] * N * Kd, N * Kd), sC=(batch[1] * M * N, M * N))
    us = time_it(f)
    fl = 2.0 * M * N * Kd * b
    print(f"gemm M={M} N={N} K={Kd} tA={int(tA)} tB={int(tB)} b={batch}: {us:8.1f} us  {fl / us / 1e6:8.1f} TF/s", flush=True)


def conv(B, H, W, Cin, Cout, k=3, stride=1):
    X = torch.randn(B * H * W, Cin, device=dev).to(T)
    Wt = torch.randn(Cout, k, k, Cin, device=dev).to(T)
    Ho, Wo = H // stride, W // stride
    Y = torch.empty(B * Ho * Wo, Cout, device=dev, dtype=T)
    f = lambda: K.conv2d(X, Wt, Y, B, H, W, Cin, Ho, Wo, Cout, k, k, stride, k // 2)
    us = time_it(f)
    fl = 2.0 * B * Ho * Wo * Cout * k * k * Cin
    print(f"conv B={B} {H}x{W} Cin={Cin} Cout={Cout} k={k} s={stride}: {us:8.1f} us  {fl / us / 1e6:8.1f} TF/s", flush=True)


if __name__ == "__main__":
    print("env:", {k: v for k, v in os.environ.items() if k.startswith("COMAT_")})
    conv(2, 64, 64, 320, 320)
    conv(2, 32, 32, 640, 640)
    conv(2, 16, 16, 1280, 1280)
    conv(2, 8, 8, 1280, 1280)
    conv(1, 128, 128, 512, 512)
    conv(1, 256, 256, 256, 256)
    conv(1, 512, 512, 128, 128)
    gemm(8192, 320, 320)
    gemm(2048, 640, 640)
    gemm(512, 1280, 1280)
    gemm(512, 1280, 128)
    gemm(512, 1280, 128, tB=True)
    gemm(8192, 2560, 320)
    gemm(8192, 320, 1280)
    gemm(320, 128, 8192, tA=True, tB=True, out=torch.float32)
    gemm(4096, 4096, 40, batch=(2, 8), out=torch.float32)
    gemm(4096, 40, 4096, tB=True, batch=(2, 8))
    gemm(4096, 40, 4096, tA=True, tB=True, batch=(2, 8))
    gemm(577, 1024, 1024)
    gemm(4096, 4096, 4096)
