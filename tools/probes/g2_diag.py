"""Diagnostic launches of the pipelined GEMM / conv kernel for counter passes (rocprofv3 --pmc ... -- python tools/probes/g2_diag.py)
and, without a profiler, a timing table of the same problems under kernel options (hipGraph-replayed, 20 launches each).

    python tools/probes/g2_diag.py            -> timing table: problem x (g2_order 0/1/2, forced tile shapes)
    G2_DIAG_PMC=1 python tools/probes/g2_diag.py   -> N launches per problem and order, in a fixed sequence (printed), for --pmc passes

Problems = the shapes that carry the C2 step's pipelined-kernel time (profiles/r03_z_bench_shapes.txt)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from comat_amd import _hip  # noqa: E402

K = _hip.HipKernels()
dev = torch.device("cuda:0")
T = torch.bfloat16
PMC = os.environ.get("G2_DIAG_PMC") == "1"
N = int(os.environ.get("G2_DIAG_N", "4"))


def R(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(T)


def conv(B, H, Cin, Cout):
    X, W, Y = R(B * H * H, Cin), R(Cout, 3, 3, Cin), torch.empty(B * H * H, Cout, device=dev, dtype=T)
    return (f"conv B={B} {H}x{H} {Cin}->{Cout}", 2.0 * B * H * H * Cout * 9 * Cin,
            lambda: K.conv2d(X, W, Y, B, H, H, Cin, H, H, Cout, 3, 3, 1, 1))


def gemm(M, N_, K_):
    A, B_, C = R(M, K_), R(N_, K_), torch.empty(M, N_, device=dev, dtype=T)
    return (f"gemm {M}x{N_}x{K_}", 2.0 * M * N_ * K_, lambda: K.gemm(A, B_, C, M, N_, K_, K_, K_, N_))


def seg(M, N_, K_, r, batch=1):
    A, W = R(M, K_), R(batch, N_, K_)
    Hh, U = R(M, batch * r), R(batch, N_, r)
    C = torch.empty(batch, M, N_, device=dev, dtype=T)
    if batch == 1:
        fn = lambda: K.gemm_segments([(A, W[0], K_, K_, K_), (Hh, U[0], r, r, r)], C[0], M, N_, N_)
    else:
        fn = lambda: K.gemm_segments([(A, W[0], K_, K_, K_, 0, N_ * K_), (Hh, U[0], r, batch * r, r, r, N_ * r)], C, M, N_, N_,
                                     batch=batch, sC=M * N_)
    return (f"seg {M}x{N_}x({K_}+{r}) b={batch}", 2.0 * M * N_ * (K_ + r) * batch, fn)


PROBLEMS = [conv(2, 64, 320, 320), conv(2, 32, 640, 640), conv(2, 16, 1280, 1280), conv(2, 8, 1280, 1280),
            conv(2, 16, 2560, 1280), gemm(8192, 2560, 320), gemm(8192, 320, 1280), gemm(2048, 5120, 640),
            gemm(512, 10240, 1280), gemm(512, 1280, 5120), gemm(512, 128, 1280), gemm(2048, 128, 640), gemm(8192, 128, 320),
            seg(512, 1280, 1280, 128), seg(2048, 640, 640, 128), seg(8192, 320, 320, 128), seg(512, 1280, 1280, 128, 3),
            seg(8192, 320, 320, 128, 3)]


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


if PMC:
    seq = 0
    for order in (0, 1):
        _hip.set_option("g2_order", order)
        for name, fl, fn in PROBLEMS:
            for _ in range(N):
                fn()
            print(f"seq {seq}: g2_order={order} {name} x{N}")
            seq += 1
    torch.cuda.synchronize()
else:
    VARIANTS = [("default", {}), ("order1", dict(g2_order=1)), ("order2", dict(g2_order=2)),
                ("128x128", dict(g2_cfg=1)), ("128x64", dict(g2_cfg=2)), ("256x128", dict(g2_cfg=3)), ("64x128", dict(g2_cfg=4)),
                ("64x64", dict(g2_cfg=6)), ("128x128w8", dict(g2_cfg=7)), ("64x64k4", dict(g2_cfg=8)), ("128x64k4", dict(g2_cfg=9)),
                ("64x128k4", dict(g2_cfg=10)), ("128x128k4", dict(g2_cfg=11))]
    print("# us per launch (20 back-to-back launches replayed from a hipGraph, best of 3); TF/s of the default in brackets")
    print("# problem | " + " | ".join(v for v, _ in VARIANTS))
    for name, fl, fn in PROBLEMS:
        row = []
        for vname, opts in VARIANTS:
            for k_, v_ in dict(g2_order=0, g2_cfg=0, g2_splits=0).items():
                _hip.set_option(k_, v_)
            for k_, v_ in opts.items():
                _hip.set_option(k_, v_)
            try:
                row.append(timeit(fn))
            except Exception as e:  # noqa: BLE001
                row.append(float("nan"))
        print(f"{name:34s} [{fl / row[0] / 1e6:6.0f} TF/s] " + " ".join(f"{t:7.1f}" for t in row), flush=True)
