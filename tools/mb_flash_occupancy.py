"""Forward / backward attention time per (batch, head) as the grid grows from one to four 4-wave blocks per CU: how much do more
waves per SIMD buy on the kernels as they are?  (The 2-tile forward holds 128 registers and 36.9 KB of LDS per block: four
blocks fit a CU; the backward kernels hold 196 / 242 registers: two.)  python tools/mb_flash_occupancy.py  (GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
from comat_amd import _hip, ops  # noqa: E402
from mb_flash_ab import replay_us  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    kb = _hip.HipKernels()
    ops.set_kernel_backend(kb)
    T = torch.bfloat16
    print(f"# build {_hip.build_id()}; H=8 Nq=Nk=4096 d=40 (2-tile kernels); blocks per CU = B * 8 * 32 / 256 = B")
    for d, H in ((40, 8), (64, 10)):
        for B in (1, 2, 3, 4, 6, 8):
            HD = H * d
            Nq = Nk = 4096
            q, k, v, g = (torch.randn(B * n, HD, device=dev).to(T) for n in (Nq, Nk, Nk, Nq))
            o = torch.empty_like(q)
            lse = torch.empty(B, H, Nq, device=dev)
            dbuf = torch.empty(B, H, Nq, device=dev)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
            sc = d ** -0.5
            tf = replay_us(lambda: kb.flash_attn_fwd(q, k, v, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, sc), n=10)
            tb = replay_us(lambda: kb.flash_attn_bwd(q, k, v, o, g, lse, dbuf, dq, dk, dv, B, H, Nq, Nk, d, HD, HD, HD, HD, sc), n=10)
            print(f"d={d} H={H:2d} B={B}: fwd {tf:7.1f} us = {tf / B:6.1f} per batch entry   bwd {tb:7.1f} us = {tb / B:6.1f} per batch entry", flush=True)


if __name__ == "__main__":
    main()
