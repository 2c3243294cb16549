"""Microbenchmark of the two GEMM / conv kernels on the shapes that carry a C2 step (run on the GPU box):
    python tools/mb_gemm2.py [> gpurun_out/mb_gemm2.txt]
For every shape: the general 64x64 kernel (gemm2 = 0, its own split plan) and the LDS-DMA pipelined kernel under every
block tile (g2_cfg 1..4) with the automatic split plan, plus forced split counts for the short-on-tiles shapes.
Prints microseconds per launch (HIP events around 20 back-to-back launches) and algorithmic TFLOP/s."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from comat_amd import _hip, ops  # noqa: E402

CONVS = [  # B, H, W, Cin, Cout, ups
    (2, 64, 64, 320, 320, 1), (2, 32, 32, 640, 640, 1), (2, 16, 16, 1280, 1280, 1), (2, 8, 8, 1280, 1280, 1),
    (2, 16, 16, 2560, 1280, 1), (2, 64, 64, 640, 320, 1), (2, 32, 32, 1280, 640, 1), (2, 64, 64, 960, 320, 1),
    (2, 32, 32, 640, 640, 2), (1, 64, 64, 512, 512, 1), (1, 128, 128, 512, 512, 1), (1, 256, 256, 256, 256, 1),
    (1, 512, 512, 128, 128, 1), (1, 256, 256, 256, 256, 2)]
GEMMS = [  # M, N, K
    (8192, 320, 320), (8192, 2560, 320), (8192, 320, 1280), (8192, 320, 2560), (8192, 128, 320), (8192, 384, 320),
    (2048, 640, 640), (2048, 5120, 640), (2048, 640, 2560), (512, 1280, 1280), (512, 10240, 1280), (512, 1280, 5120),
    (577, 1024, 1024), (577, 3072, 1024), (577, 4096, 1024), (577, 1024, 4096), (128, 1280, 1280)]
SEGS = [  # M, N, [K...]
    (8192, 320, [320, 128]), (2048, 640, [640, 128]), (512, 1280, [1280, 128]), (8192, 320, [320, 320, 320, 384]),
    (512, 1280, [1280, 1280, 1280, 384])]


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3  # us


def variants(tiles_hint):
    v = [("general", dict(gemm2=0, g2_cfg=0, g2_splits=0))]
    for c, name in ((1, "128x128"), (2, "128x64"), (3, "256x128"), (4, "64x128")):
        v.append((f"g2 {name}", dict(gemm2=1, g2_cfg=c, g2_splits=0)))
    if tiles_hint < 256:
        for s in (1, 2, 4, 8):
            v.append((f"g2 128x128 s{s}", dict(gemm2=1, g2_cfg=1, g2_splits=s)))
            v.append((f"g2 64x128 s{s}", dict(gemm2=1, g2_cfg=4, g2_splits=s)))
    v.append(("g2 auto", dict(gemm2=1, g2_cfg=0, g2_splits=0)))
    return v


def run(name, flops, fn, tiles_hint):
    best = None
    for vn, opts in variants(tiles_hint):
        for k_, v_ in opts.items():
            _hip.set_option(k_, v_)
        us = timeit(fn)
        tf = flops / us / 1e6
        print(f"{name:58s} {vn:18s} {us:9.1f} us {tf:8.1f} TF/s", flush=True)
        if best is None or us < best[1]:
            best = (vn, us)
    print(f"{name:58s} BEST {best[0]} {best[1]:.1f} us", flush=True)


def main():
    dev = torch.device("cuda:0")
    k = _hip.HipKernels()
    ops.set_kernel_backend(k)
    T = torch.bfloat16
    g = torch.Generator(device="cpu").manual_seed(0)
    r = lambda *s: (torch.randn(*s, generator=g) * 0.5).to(dev, T)
    for (B, H, W, Cin, Cout, ups) in CONVS:
        x, w = r(B * H * W, Cin), r(Cout, 3, 3, Cin)
        Ho, Wo = H * ups, W * ups
        y = torch.empty((B * Ho * Wo, Cout), dtype=T, device=dev)
        bias = torch.zeros(Cout, device=dev)
        fn = lambda: k.conv2d(x, w, y, B, H, W, Cin, Ho, Wo, Cout, 3, 3, 1, 1, mode=0, ups=ups, bias=bias)
        run(f"conv B={B} {H}x{W} {Cin}->{Cout} ups={ups}", 2.0 * B * Ho * Wo * Cout * 9 * Cin, fn,
            (B * Ho * Wo // 128) * max(Cout // 128, 1))
    for (M, N, K) in GEMMS:
        a, b = r(M, K), r(N, K)
        c = torch.empty((M, N), dtype=T, device=dev)
        bias = torch.zeros(N, device=dev)
        fn = lambda: k.gemm(a, b, c, M, N, K, K, K, N, bias=bias)
        run(f"gemm M={M} N={N} K={K}", 2.0 * M * N * K, fn, max(M // 128, 1) * max(N // 128, 1))
    for (M, N, Ks) in SEGS:
        segs = [(r(M, K_), r(N, K_), K_, K_, K_) for K_ in Ks]
        c = torch.empty((M, N), dtype=T, device=dev)
        fn = lambda: k.gemm_segments(segs, c, M, N, N)
        run(f"gemm_segments M={M} N={N} K={'+'.join(map(str, Ks))}", 2.0 * M * N * sum(Ks), fn,
            max(M // 128, 1) * max(N // 128, 1))
    for k_, v_ in dict(gemm2=1, g2_cfg=0, g2_splits=0).items():
        _hip.set_option(k_, v_)


if __name__ == "__main__":
    main()
