"""Target for `rocprofv3 --pmc ...`: the step's two heaviest attention problems, forward + backward, three times each.
    (cd /tmp && rocprofv3 --pmc SQ_WAVE_CYCLES ... -d /tmp/fd -o f -- python tools/pmc_flash_diag.py); python tools/pmc_dump.py <db>"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from comat_amd import _hip, ops  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    kb = _hip.HipKernels()
    ops.set_kernel_backend(kb)
    T = torch.bfloat16
    for (B, H, Nq, Nk, d) in ((2, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80)):
        HD = H * d
        q, k, v, g = (torch.randn(B * n, HD, device=dev).to(T) for n in (Nq, Nk, Nk, Nq))
        o = torch.empty_like(q)
        lse = torch.empty(B, H, Nq, device=dev)
        dbuf = torch.empty(B, H, Nq, device=dev)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        for _ in range(3):
            kb.flash_attn_fwd(q, k, v, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
            kb.flash_attn_bwd(q, k, v, o, g, lse, dbuf, dq, dk, dv, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
        torch.cuda.synchronize()


if __name__ == "__main__":
    main()
