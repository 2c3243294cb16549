"""Are two builds of the fused attention kernels the same function, bit for bit?

    python tools/flash_bits.py dump /tmp/a.pt          (with build A in comat_amd/lib/)
    python tools/flash_bits.py dump /tmp/b.pt          (with build B swapped in)
    python tools/flash_bits.py compare /tmp/a.pt /tmp/b.pt

Forward and backward of every shape class of the step (self / cross attention at the four UNet levels, BLIP, ragged tiles,
the query-split dK/dV path, head dims 32 .. 160) on seeded inputs; O, the log-sum-exp, dQ, dK, dV and D are stored raw.  Used
when a change is meant to alter only the instruction stream (register classes, packed arithmetic, hoisted masks): equal
files carry the GPU suite's verdict on build A over to build B."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [  # B, H, Nq, Nk, d
    (2, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80), (2, 8, 256, 256, 160), (2, 8, 64, 64, 160), (2, 8, 4096, 77, 40),
    (2, 8, 1024, 77, 80), (2, 8, 256, 77, 160), (1, 16, 577, 577, 64), (1, 12, 16, 577, 64), (1, 3, 300, 200, 40),
    (2, 2, 70, 130, 32), (1, 1, 130, 65, 48), (1, 2, 96, 4096, 40), (1, 2, 4096, 100, 40), (2, 20, 1024, 1024, 64),
    (1, 10, 4096, 4096, 64), (1, 1, 33, 31, 8), (1, 2, 129, 257, 96)]


def dump(path):
    from comat_amd import _hip, ops
    k = _hip.HipKernels()
    ops.set_kernel_backend(k)
    dev = torch.device("cuda:0")
    out = {}
    for dtype in (torch.bfloat16, torch.float32):
        for (B, H, Nq, Nk, d) in SHAPES:
            if dtype == torch.float32 and Nq * Nk > 1024 * 1024:
                continue
            g = torch.Generator().manual_seed(B * 1000 + Nq + Nk + d)
            HD = H * d
            mk = lambda n: torch.randn(B * n, HD, generator=g).to(dtype).to(dev)
            q, kk, v, go = mk(Nq), mk(Nk), mk(Nk), mk(Nq)
            o = torch.empty_like(q)
            lse, dbuf = torch.empty(B, H, Nq, device=dev), torch.empty(B, H, Nq, device=dev)
            dq, dk, dv = torch.empty_like(q), torch.empty_like(kk), torch.empty_like(v)
            k.flash_attn_fwd(q, kk, v, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
            k.flash_attn_bwd(q, kk, v, o, go, lse, dbuf, dq, dk, dv, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
            torch.cuda.synchronize()
            out[(str(dtype), B, H, Nq, Nk, d)] = [t.cpu() for t in (o, lse, dq, dk, dv, dbuf)]
    torch.save(out, path)
    print(f"{len(out)} cases -> {path}")


def compare(pa, pb):
    a, b = torch.load(pa), torch.load(pb)
    bad = 0
    for key in a:
        for name, x, y in zip(("O", "lse", "dQ", "dK", "dV", "D"), a[key], b[key]):
            if not torch.equal(x.view(torch.uint8), y.view(torch.uint8)):
                bad += 1
                diff = (x.float() - y.float()).abs().max().item()
                print(f"DIFFERENT {key} {name}: max |a - b| = {diff:.3e}")
    print(f"{len(a)} cases x 6 tensors compared: " + ("all bit-identical" if not bad else f"{bad} tensors differ"))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(dump(sys.argv[2]) if sys.argv[1] == "dump" else compare(sys.argv[2], sys.argv[3]))
