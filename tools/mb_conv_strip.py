"""3x3 convs of the step: the im2col form of the pipelined kernel against the strip form (gemm2_strip_kernel), us per launch
(20 launches replayed from a hipGraph, best of 3), with the planned block shape and with forced ones.

    python tools/mb_conv_strip.py > gpurun_out/mb_conv_strip.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import _hip  # noqa: E402

K = _hip.HipKernels()
dev = torch.device("cuda:0")
T = torch.bfloat16


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
    base = dict(g2_strip=0, g2_cfg=0, g2_splits=0)
    base.update(kw)
    for k_, v_ in base.items():
        _hip.set_option(k_, v_)


CFGS = [(0, "plan"), (1, "128x128"), (7, "128x128w8"), (5, "128x128d6"), (2, "128x64"), (4, "64x128"), (6, "64x64")]
print("# us per launch: im2col / strip, per block shape (plan = the plan table's shape and split count)")
print("# problem".ljust(36) + " | ".join(f"{n:>13s}" for _, n in CFGS))
for B, H, Cin, Cout in [(2, 64, 320, 320), (2, 64, 640, 320), (2, 32, 640, 640), (2, 32, 1280, 640), (2, 16, 1280, 1280), (2, 16, 2560, 1280),
                        (1, 64, 320, 320), (1, 64, 512, 512), (1, 128, 512, 512), (1, 256, 256, 256), (1, 512, 128, 128), (1, 256, 512, 256)]:
    X = (torch.randn(B * H * H, Cin, device=dev) * 0.5).to(T)
    W = (torch.randn(Cout, 3, 3, Cin, device=dev) * 0.05).to(T)
    Ws = W.reshape(Cout, 3, 3, Cin // 32, 32).permute(0, 3, 1, 2, 4).contiguous()
    Y = torch.empty(B * H * H, Cout, device=dev, dtype=T)
    fl = 2.0 * B * H * H * Cout * 9 * Cin
    row = []
    for c, _ in CFGS:
        opts(g2_cfg=c)
        t0 = timeit(lambda: K.conv2d(X, W, Y, B, H, H, Cin, H, H, Cout, 3, 3, 1, 1))
        opts(g2_cfg=c, g2_strip=1)
        t1 = timeit(lambda: K.conv2d(X, W, Y, B, H, H, Cin, H, H, Cout, 3, 3, 1, 1, W_strip=Ws))
        used = _hip.last_gemm_kernel()
        row.append(f"{t0:6.1f}/{t1:6.1f}{'' if used == 7 else '*'}")
    print(f"conv B={B} {H}x{H} {Cin}->{Cout}".ljust(36) + " | ".join(row) + f"   [{fl / 1e9:.1f} GFLOP]", flush=True)
print("# (* = the strip form did not take the problem with that block shape)")
opts()
