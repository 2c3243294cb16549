"""fp8 (e4m3, 32x32x64 MFMA) against bf16 on the pipelined kernel at SDXL-1024^2 shapes, plus the cost of the per-tensor
activation quantisation (two HBM streams).  Prints one line per problem: microseconds and TFLOP/s per variant.

    python tools/mb_fp8.py > gpurun_out/mb_fp8.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import _hip  # noqa: E402

k = _hip.HipKernels()
dev = torch.device("cuda:0")
T = torch.bfloat16


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        fn()
    side.synchronize()
    with torch.cuda.graph(g, stream=side):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def report(name, flops, t16, t8, tq):
    print(f"{name:44s} bf16 {t16:8.1f} us {flops / t16 / 1e6:7.1f} TF/s | fp8 {t8:8.1f} us {flops / t8 / 1e6:7.1f} TF/s | "
          f"quantise x {tq:7.1f} us | fp8 + quantise vs bf16: {t16 / (t8 + tq):.2f}x", flush=True)


for M, N, K in [(32768, 2560, 320), (32768, 320, 1280), (8192, 5120, 640), (8192, 640, 2560), (8192, 640, 640),
                (2048, 10240, 1280), (2048, 1280, 5120), (2048, 1280, 1280)]:
    x, w = torch.randn(M, K, device=dev).to(T), (torch.randn(N, K, device=dev) * 0.05).to(T)
    y = torch.empty(M, N, device=dev, dtype=T)
    x8, sx = k.fp8_quantize(x)
    w8, sw = k.fp8_quantize(w)
    t16 = timeit(lambda: k.gemm(x, w, y, M, N, K, K, K, N))
    t8 = timeit(lambda: k.gemm(x8, w8, y, M, N, K, K, K, N, scales=(sx, sw)))
    tq = timeit(lambda: k.fp8_quantize(x, out=x8, scale=sx))
    report(f"gemm {M}x{N}x{K}", 2.0 * M * N * K, t16, t8, tq)

for B, H, C in [(2, 128, 320), (2, 64, 640), (2, 32, 1280)]:
    x, w = torch.randn(B * H * H, C, device=dev).to(T), (torch.randn(C, 3, 3, C, device=dev) * 0.05).to(T)
    y = torch.empty(B * H * H, C, device=dev, dtype=T)
    x8, sx = k.fp8_quantize(x)
    w8, sw = k.fp8_quantize(w)
    t16 = timeit(lambda: k.conv2d(x, w, y, B, H, H, C, H, H, C, 3, 3, 1, 1))
    t8 = timeit(lambda: k.conv2d(x8, w8, y, B, H, H, C, H, H, C, 3, 3, 1, 1, scales=(sx, sw)))
    tq = timeit(lambda: k.fp8_quantize(x, out=x8, scale=sx))
    report(f"conv3x3 B={B} {H}x{H} {C}->{C}", 2.0 * B * H * H * C * C * 9, t16, t8, tq)
