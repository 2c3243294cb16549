"""Forward attention of the two heaviest problems on one library build (COMAT_LIB_PATH): tools/calls/r4_call13.sh runs it on
builds of attention.hip with parts of the 2-tile forward loop deleted (COMAT_FLASH_DIAG) - what each part costs in place."""
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
    row = []
    for (B, H, Nq, Nk, d) in ((2, 8, 4096, 4096, 40), (1, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80)):
        HD = H * d
        q, k, v = (torch.randn(B * n, HD, device=dev).to(T) for n in (Nq, Nk, Nk))
        o = torch.empty_like(q)
        lse = torch.empty(B, H, Nq, device=dev)
        row.append(replay_us(lambda: kb.flash_attn_fwd(q, k, v, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)))
    print(f"{os.path.basename(os.environ.get('COMAT_LIB_PATH', 'default')):24s} fwd 2x8x4096^2 d40 {row[0]:7.1f} us   1x8x4096^2 d40 {row[1]:7.1f} us   "
          f"2x8x1024^2 d80 {row[2]:7.1f} us", flush=True)


if __name__ == "__main__":
    main()
