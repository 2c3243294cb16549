"""GroupNorm(+SiLU) forward / backward: the one-launch form (a workgroup per (sample, group), group held in registers)
against the three-launch form, on the UNet's shapes (CFG batch 2).  microseconds per call, replayed from a hipGraph.

    python tools/mb_gn.py > gpurun_out/mb_gn.txt"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from comat_amd import _hip  # noqa: E402

k = _hip.HipKernels()
dev = torch.device("cuda:0")
T = torch.bfloat16


def graph_time(fn, reps=20):
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        k.prepare_stream(dev)
        fn()
    side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps * 1e3


print("# B HW C G : fwd us (norm_fused 0 -> 3 -> 4 -> 5), bwd us (0 -> 3 -> 4 -> 5), bytes moved fwd; 0 = three launches, 3 = one launch where it pays else three, 4 = one launch where it pays else two (ticket), 5 = ... else two (finalize in the apply kernel's prologue)")
for B, HW, C, G in [(2, 4096, 320, 32), (2, 4096, 640, 32), (2, 4096, 960, 32), (2, 1024, 640, 32), (2, 1024, 1280, 32),
                    (2, 1024, 1920, 32), (2, 256, 1280, 32), (2, 256, 2560, 32), (2, 64, 1280, 32), (2, 64, 2560, 32),
                    (1, 4096, 320, 32), (1, 4096, 512, 32), (1, 16384, 512, 32), (1, 65536, 256, 32), (1, 262144, 128, 32)]:
    x = torch.randn(B * HW, C, device=dev).to(T)
    dy = torch.randn(B * HW, C, device=dev).to(T)
    add = torch.randn(B * HW, C, device=dev).to(T)
    y, dx = torch.empty_like(x), torch.empty_like(x)
    gamma, beta = torch.ones(C, device=dev), torch.zeros(C, device=dev)
    stats = torch.empty(B, G, 2, device=dev)
    row = []
    for mode in (0, 3, 4, 5):
        _hip.set_option("norm_fused", mode)
        tf = graph_time(lambda: k.groupnorm_fwd(x, gamma, beta, y, stats, B, HW, C, G, 1e-5, True))
        tb = graph_time(lambda: k.groupnorm_bwd(dy, x, gamma, beta, stats, dx, B, HW, C, G, True, add=add))
        row.append((tf, tb))
    print(f"{B} {HW:5d} {C:5d} {G}: fwd {row[0][0]:6.1f} -> {row[1][0]:6.1f} -> {row[2][0]:6.1f} -> {row[3][0]:6.1f}   bwd {row[0][1]:6.1f} -> {row[1][1]:6.1f} -> {row[2][1]:6.1f} -> {row[3][1]:6.1f}   "
          f"({2 * x.numel() * 2 / 1e6:.1f} MB)", flush=True)
