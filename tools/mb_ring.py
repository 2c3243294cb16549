"""Ring depth of the 128-byte-k-tile block shapes of the pipelined kernel on the UNet's latency-bound problems (round 6): the plan
table's choice ("auto") against forced shapes x split counts, 16 launches replayed from a hipGraph.
    python tools/mb_ring.py > gpurun_out/mb_ring.txt"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from comat_amd import _hip, ops  # noqa: E402
from tools.mb_big import timeit  # noqa: E402

GEMMS = [(512, 1408, 1280), (512, 1280, 1280), (2048, 768, 640), (2048, 640, 640), (8192, 448, 320), (8192, 320, 320),
         (8192, 320, 1280), (2048, 640, 2560), (512, 1280, 5120), (2048, 5120, 640), (8192, 2560, 320)]
CONVS = [(2, 16, 16, 1280, 1280), (2, 8, 8, 1280, 1280), (2, 32, 32, 640, 640), (2, 64, 64, 320, 320)]
NAMES = {0: "auto", 6: "64x64 k2 d8", 8: "64x64 k4 d4", 13: "64x64 k4 d8", 14: "64x64 k4 d6", 2: "128x64 k2 d6", 9: "128x64 k4 d4",
         15: "128x64 k4 d6", 4: "64x128 k2 d6", 10: "64x128 k4 d4", 16: "64x128 k4 d6", 1: "128x128 k2 d4", 11: "128x128 k4 d4"}
CFGS = [int(c) for c in os.environ.get("MB_CFGS", "0,8,14,13,9,15,10,16,11").split(",")]
SPLITS = [int(c) for c in os.environ.get("MB_SPLITS", "1,2,4").split(",")]


def main():
    dev = torch.device("cuda:0")
    k = _hip.HipKernels()
    ops.set_kernel_backend(k)
    T = torch.bfloat16
    r = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).to(T)

    def sweep(tag, fn, flop, ref_out):
        base = None
        for c in CFGS:
            for s in ([0] if c == 0 else SPLITS):
                _hip.set_option("g2_cfg", c)
                _hip.set_option("g2_splits", s)
                us = timeit(fn)
                out = ref_out().float()
                if base is None:
                    base = out.clone()
                dev_ = ((out - base).abs().max() / base.abs().max().clamp_min(1e-9)).item()
                print(f"{tag:34s} {NAMES.get(c, str(c)):14s} s={s:2d} {us:8.1f} us {flop / us / 1e6:7.1f} TF/s  dev {dev_:.1e}", flush=True)
        _hip.set_option("g2_cfg", 0)
        _hip.set_option("g2_splits", 0)

    for (M, N, K) in GEMMS:
        a, b = r(M, K), r(N, K) * K ** -0.5
        c_ = torch.empty((M, N), dtype=T, device=dev)
        sweep(f"gemm {M}x{N}x{K}", lambda: k.gemm(a, b, c_, M, N, K, K, K, N), 2.0 * M * N * K, lambda: c_)
    for (B, H, W, Cin, Cout) in CONVS:
        x, w = r(B * H * W, Cin), r(Cout, 3, 3, Cin) * (9 * Cin) ** -0.5
        y = torch.empty((B * H * W, Cout), dtype=T, device=dev)
        sweep(f"conv {B}x{H}x{W} {Cin}->{Cout}", lambda: k.conv2d(x, w, y, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, mode=0, ups=1),
              2.0 * B * H * W * Cout * 9 * Cin, lambda: y)


if __name__ == "__main__":
    main()
