"""Microbenchmark of the block shapes of the pipelined GEMM / conv kernel (gemm2.hip) on the problems VERDICT r5 names and the
mid-size problems of the C2 / bs-4 / C5 steps (run on the GPU box):
    python tools/mb_big.py [quick] > gpurun_out/mb_big.txt
Per problem: microseconds per launch (16 launches replayed from a hipGraph, best of 3 replays) under the plan table ("auto") and
under forced block shapes / split counts, the algorithmic TFLOP/s, and the largest deviation of each forced variant's output from
the auto variant's (same bf16 operands, fp32 accumulation in a different order: ~1e-2 relative to the output's scale at most)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from comat_amd import _hip, ops  # noqa: E402

CONVS = [  # B, H, W, Cin, Cout, ups
    (1, 128, 128, 512, 512, 1), (1, 256, 256, 256, 256, 1), (2, 64, 64, 320, 320, 1), (2, 16, 16, 1280, 1280, 1),
    (1, 64, 64, 512, 512, 1), (1, 512, 512, 128, 128, 1), (1, 256, 256, 512, 256, 1), (1, 128, 128, 512, 512, 2),
    (2, 32, 32, 640, 640, 1), (2, 64, 64, 640, 320, 1), (2, 32, 32, 1280, 640, 1), (2, 64, 64, 960, 320, 1),
    (2, 16, 16, 2560, 1280, 1), (2, 8, 8, 1280, 1280, 1), (8, 64, 64, 320, 320, 1), (8, 32, 32, 640, 640, 1),
    (2, 128, 128, 320, 320, 1), (2, 64, 64, 640, 640, 1)]
GEMMS = [  # M, N, K
    (2048, 5120, 640), (8192, 2560, 320), (8192, 320, 1280), (2048, 640, 2560), (512, 10240, 1280), (512, 1280, 5120),
    (8192, 320, 320), (2048, 640, 640), (512, 1280, 1280), (32768, 320, 320), (32768, 2560, 320), (8192, 5120, 640),
    (32768, 640, 640), (8192, 1280, 1280)]
QUICK_CONVS, QUICK_GEMMS = CONVS[:4], GEMMS[:1]

_side = None


def timeit(fn, n=16):
    global _side
    if _side is None:
        _side = torch.cuda.Stream()
    with torch.cuda.stream(_side):
        fn()
        fn()
    _side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=_side):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(3):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


def variants(M, N):
    v = [("auto", 0, 0), ("128x128 w4", 1, 0), ("128x128 k4", 11, 0), ("256x128", 3, 0)]
    t256 = -(-M // 256) * -(-N // 256)
    if t256 >= 32:
        v.append(("256x256", 12, 1))
    for extra in os.environ.get("MB_EXTRA_CFGS", "").split(","):
        if extra:
            v.append((f"cfg {extra}", int(extra), 0))
    return v


def run(name, flops, fn, out, M, N):
    ref = None
    best = None
    for vn, c, s in variants(M, N):
        _hip.set_option("g2_cfg", c)
        _hip.set_option("g2_splits", s)
        out.fill_(float("nan"))
        fn()
        torch.cuda.synchronize()
        o = out.float()
        if ref is None:
            ref = o.clone()
            dev_ = 0.0
        else:
            dev_ = float((o - ref).abs().max() / ref.abs().max().clamp_min(1e-6))
            if not torch.isfinite(o).all():
                dev_ = float("nan")
        us = timeit(fn)
        print(f"{name:44s} {vn:16s} {us:9.1f} us {flops / us / 1e6:8.1f} TF/s   dev {dev_:.1e}", flush=True)
        if best is None or us < best[1]:
            best = (vn, us)
    print(f"{name:44s} BEST {best[0]} {best[1]:.1f} us", flush=True)


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    dev = torch.device("cuda:0")
    k = _hip.HipKernels()
    ops.set_kernel_backend(k)
    T = torch.bfloat16
    r = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).to(T)  # uniform [-1, 1): the guide's fill for quoted rates (rule 25)
    for (B, H, W, Cin, Cout, ups) in (QUICK_CONVS if quick else CONVS):
        x, w = r(B * H * W, Cin), r(Cout, 3, 3, Cin) * (9 * Cin) ** -0.5
        Ho, Wo = H * ups, W * ups
        y = torch.empty((B * Ho * Wo, Cout), dtype=T, device=dev)
        bias = torch.zeros(Cout, device=dev)
        fn = lambda: k.conv2d(x, w, y, B, H, W, Cin, Ho, Wo, Cout, 3, 3, 1, 1, mode=0, ups=ups, bias=bias)
        run(f"conv {B}x{H}x{W} {Cin}->{Cout} ups={ups}", 2.0 * B * Ho * Wo * Cout * 9 * Cin, fn, y, B * Ho * Wo, Cout)
        del x, w, y
    for (M, N, K) in (QUICK_GEMMS if quick else GEMMS):
        a, b = r(M, K), r(N, K) * K ** -0.5
        c = torch.empty((M, N), dtype=T, device=dev)
        bias = torch.zeros(N, device=dev)
        fn = lambda: k.gemm(a, b, c, M, N, K, K, K, N, bias=bias)
        run(f"gemm {M}x{N}x{K}", 2.0 * M * N * K, fn, c, M, N)
        del a, b, c
    _hip.set_option("g2_cfg", 0)
    _hip.set_option("g2_splits", 0)


if __name__ == "__main__":
    main()
