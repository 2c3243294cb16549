"""Fused attention on the step's own problems, one library build per process, for same-box A/B of kernel variants.

    COMAT_LIB_PATH=comat_amd/lib/ab/libcomat_fix.so python tools/mb_flash_ab.py [--check]

Every (forward, backward) problem of a C2 step with its call count (profiles/r04_z_bench_shapes.txt), replayed back to back
from a hipGraph of 20 launches (best of 3), and the sum weighted by the call counts = the step's attention time on this build.
--check also compares O / dQ / dK / dV with an fp32 torch reference at the first three shapes (max abs error)."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from comat_amd import _hip, ops  # noqa: E402

STEP = [  # B, H, Nq, Nk, d, forward calls, backward calls per C2 step
    (2, 8, 4096, 4096, 40, 30, 30), (2, 8, 1024, 1024, 80, 30, 30), (2, 8, 256, 256, 160, 30, 30), (2, 8, 4096, 77, 40, 30, 30),
    (2, 8, 1024, 77, 80, 30, 30), (2, 8, 256, 77, 160, 30, 30), (2, 8, 64, 64, 160, 6, 6), (2, 8, 64, 77, 160, 6, 6),
    (1, 8, 4096, 4096, 40, 5, 5), (1, 8, 1024, 1024, 80, 5, 5), (1, 8, 256, 256, 160, 5, 5), (1, 8, 4096, 77, 40, 5, 5),
    (1, 8, 1024, 77, 80, 5, 5), (1, 8, 256, 77, 160, 5, 5), (1, 16, 577, 577, 64, 24, 24), (1, 12, 16, 577, 64, 12, 12),
    (2, 10, 4096, 4096, 64, 0, 0), (2, 20, 1024, 1024, 64, 0, 0)]  # last two: SDXL levels (C4), not in the C2 sum


def replay_us(fn, n=20, reps=3):
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
    best = 1e30
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


def reference(q, k, v, g, B, H, Nq, Nk, d):
    def heads(x, n):
        return x.float().view(B, n, H, d).permute(0, 2, 1, 3).detach().requires_grad_(True)
    qh, kh, vh = heads(q, Nq), heads(k, Nk), heads(v, Nk)
    p = torch.softmax(qh @ kh.transpose(-1, -2) * d ** -0.5, dim=-1)
    o = p @ vh
    o.backward(g.float().view(B, Nq, H, d).permute(0, 2, 1, 3))
    back = lambda x, n: x.permute(0, 2, 1, 3).reshape(B * n, H * d)
    return back(o.detach(), Nq), back(qh.grad, Nq), back(kh.grad, Nk), back(vh.grad, Nk)


def main():
    dev = torch.device("cuda:0")
    kb = _hip.HipKernels()
    ops.set_kernel_backend(kb)
    T = torch.bfloat16
    lib = os.environ.get("COMAT_LIB_PATH", "comat_amd/lib/libcomat_hip.so")
    print(f"# {lib}  build {_hip.build_id()}")
    tot_f = tot_b = 0.0
    for n, (B, H, Nq, Nk, d, cf, cb) in enumerate(STEP):
        HD = H * d
        gen = torch.Generator(device=dev).manual_seed(1234 + n)
        q = torch.randn(B * Nq, HD, device=dev, generator=gen).to(T)
        k = torch.randn(B * Nk, HD, device=dev, generator=gen).to(T)
        v = torch.randn(B * Nk, HD, device=dev, generator=gen).to(T)
        g = torch.randn(B * Nq, HD, device=dev, generator=gen).to(T)
        o = torch.empty_like(q)
        lse = torch.empty(B, H, Nq, device=dev)
        dbuf = torch.empty(B, H, Nq, device=dev)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
        sc = d ** -0.5
        fwd = lambda: kb.flash_attn_fwd(q, k, v, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, sc)
        bwd = lambda: kb.flash_attn_bwd(q, k, v, o, g, lse, dbuf, dq, dk, dv, B, H, Nq, Nk, d, HD, HD, HD, HD, sc)
        fwd()
        tf, tb = replay_us(fwd), replay_us(bwd)
        tot_f += tf * cf
        tot_b += tb * cb
        err = ""
        if "--check" in sys.argv and n < 3:
            ro, rq, rk, rv = reference(q, k, v, g, B, H, Nq, Nk, d)
            err = "   max|err| O %.2e dQ %.2e dK %.2e dV %.2e" % tuple(
                (a.float() - b).abs().max().item() for a, b in ((o, ro), (dq, rq), (dk, rk), (dv, rv)))
        # a checksum of the outputs: builds whose arithmetic is unchanged print the same digits
        chk = sum(x.float().sum().item() for x in (o, dq, dk, dv))
        print(f"B={B} H={H:2d} Nq={Nq:4d} Nk={Nk:4d} d={d:3d}  fwd {tf:7.1f} us  bwd {tb:7.1f} us  x{cf:2d}  sum {chk:+.6e}{err}", flush=True)
    print(f"# C2 step: forward {tot_f / 1e3:.2f} ms + backward {tot_b / 1e3:.2f} ms = {(tot_f + tot_b) / 1e3:.2f} ms of attention "
          f"(main at round-4 end, in step: 8.38 + 20.0)")


if __name__ == "__main__":
    main()
