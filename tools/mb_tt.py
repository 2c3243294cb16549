"""Split-count sweep of the k-major (LoRA weight-gradient) kernel and of the split convs at the small UNet levels.

    python tools/mb_tt.py > gpurun_out/mb_tt.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import _hip  # noqa: E402

k = _hip.HipKernels()
dev = torch.device("cuda:0")
T = torch.bfloat16

def timeit(fn, n=20):
    """back-to-back launches replayed from a hipGraph, HIP events around the replay: microseconds per launch"""
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


print("# k-major weight gradients dU [M, N] += A^T B over K tokens: microseconds per forced split count (0 = heuristic)")
for M, N, K in [(320, 128, 8192), (128, 320, 8192), (640, 128, 2048), (128, 640, 2048), (1280, 128, 512), (128, 1280, 512),
                (384, 320, 8192), (1280, 128, 128), (640, 128, 8192), (1280, 128, 2048)]:
    A = torch.randn(K, M, device=dev).to(T)
    B = torch.randn(K, N, device=dev).to(T)
    C = torch.zeros(M, N, device=dev)
    row = []
    for s in (0, 1, 2, 4, 6, 8, 12, 16, 24, 32):
        _hip.set_option("g2_splits", s)
        t = timeit(lambda: k.gemm(A, B, C, M, N, K, M, N, N, transA=True, transB=True, R=C, ldr=N, beta=1.0))
        row.append(f"{s}:{t:6.1f}")
    _hip.set_option("g2_splits", 0)
    print(f"tt {M}x{N} K={K}  " + "  ".join(row), flush=True)

print("# 3x3 convs with few output tiles: forced split counts on the table's block shape")
for B_, H, C in [(2, 8, 1280), (2, 16, 1280), (1, 8, 1280), (1, 16, 1280), (2, 16, 640)]:
    x = torch.randn(B_ * H * H, C, device=dev).to(T)
    w = (torch.randn(C, 3, 3, C, device=dev) * 0.05).to(T)
    y = torch.empty(B_ * H * H, C, device=dev, dtype=T)
    row = []
    for s in (0, 1, 2, 3, 4, 6, 8, 12, 16, 24):
        _hip.set_option("g2_splits", s)
        t = timeit(lambda: k.conv2d(x, w, y, B_, H, H, C, H, H, C, 3, 3, 1, 1))
        row.append(f"{s}:{t:6.1f}")
    _hip.set_option("g2_splits", 0)
    print(f"conv B={B_} {H}x{H} {C}->{C}  " + "  ".join(row), flush=True)
