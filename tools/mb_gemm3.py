"""Microbenchmark of the lean kernel (gemm3.hip) against the pipelined one (gemm2.hip) on the launch-latency-bound problems of
the step.  (The chained-launch part of round 4 lives with its kernels on branch exp/gemm-chain.)

    python tools/mb_gemm3.py > gpurun_out/mb_gemm3.txt

us per launch: 20 back-to-back launches replayed from a hipGraph, best of 3."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import _hip  # noqa: E402

K = _hip.HipKernels()
dev = torch.device("cuda:0")
T = torch.bfloat16


def R(*s):
    return (torch.randn(*s, device=dev) * 0.3).to(T)


def timeit(fn, n=20):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        fn()
        fn()
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) * 1e3 / n)
    return best


def opts(**kw):
    base = dict(gemm3=0, g3_cfg=0, g2_cfg=0, g2_splits=0)
    base.update(kw)
    for k_, v_ in base.items():
        _hip.set_option(k_, v_)


CFGS = [1, 2, 3, 4, 5, 6, 7, 8, 9]
NAMES = ["32x32/8", "64x32/8", "32x64/8", "64x64/4", "64x64/8", "32x32/4", "64x32/4", "32x32/16", "64x32/16"]

print("# single problems: us per launch.  pipelined = gemm2 with its plan table; lean = gemm3, tile / k-parallel waves")
print(f"# {'problem':40s} pipelined  auto | " + " ".join(f"{n:>8s}" for n in NAMES))
PLAIN = [(512, 128, 1280), (2048, 128, 640), (8192, 128, 320), (512, 384, 1280), (2048, 384, 640), (8192, 384, 320), (154, 256, 768),
         (512, 1280, 1280), (2048, 640, 640), (8192, 320, 320), (512, 1280, 5120), (2048, 640, 2560), (577, 1024, 1024),
         (577, 4096, 1024), (577, 1024, 4096), (577, 3072, 1024), (16, 768, 768), (16, 3072, 768), (16, 768, 3072), (128, 1280, 1280),
         (128, 128, 1280), (512, 10240, 1280), (2048, 5120, 640), (8192, 2560, 320), (32, 30524, 768)]
for M, N, K_ in PLAIN:
    A, B, C = R(M, K_), R(N, K_), torch.empty(M, N, device=dev, dtype=T)
    fn = lambda: K.gemm(A, B, C, M, N, K_, K_, K_, N)
    opts()
    t2 = timeit(fn)
    opts(gemm3=2)
    ta = timeit(fn)
    row = []
    for c in CFGS:
        opts(gemm3=2, g3_cfg=c)
        row.append(timeit(fn))
    print(f"gemm {M}x{N}x{K_:<28d} {t2:9.1f} {ta:5.1f} | " + " ".join(f"{t:8.1f}" for t in row), flush=True)

print("# K-segmented (frozen + low-rank): us per launch")
for M, N, K_, r, G in [(512, 1280, 1280, 128, 1), (512, 1280, 1280, 128, 3), (2048, 640, 640, 128, 1), (2048, 640, 640, 128, 3),
                       (8192, 320, 320, 128, 1), (8192, 320, 320, 128, 3), (154, 1280, 768, 128, 2), (128, 1280, 1280, 128, 3)]:
    x, W, H_, U = R(M, K_), R(G, N, K_), R(M, G * r), R(G, N, r)
    C = torch.empty(G, M, N, device=dev, dtype=T)
    fn = lambda: K.gemm_segments([(x, W[0], K_, K_, K_, 0, N * K_), (H_, U[0], r, G * r, r, r, N * r)], C, M, N, N, batch=G, sC=M * N)
    opts()
    t2 = timeit(fn)
    opts(gemm3=2)
    ta = timeit(fn)
    row = []
    for c in CFGS:
        opts(gemm3=2, g3_cfg=c)
        row.append(timeit(fn))
    print(f"seg {M}x{N}x({K_}+{r}) b={G:<18d} {t2:9.1f} {ta:5.1f} | " + " ".join(f"{t:8.1f}" for t in row), flush=True)

opts()
