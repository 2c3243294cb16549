"""Fused-attention microbenchmark on the step's shapes: forward and backward (prep + dQ + dK/dV [+ reduce]) under the four
flash_trim x flash_tr variants.  python tools/mb_flash.py  (GPU box)"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from comat_amd import _hip, ops  # noqa: E402

SHAPES = [  # B, H, Nq, Nk, d
    (2, 8, 4096, 4096, 40), (2, 8, 1024, 1024, 80), (2, 8, 256, 256, 160), (2, 8, 64, 64, 160),
    (2, 8, 4096, 77, 40), (2, 8, 1024, 77, 80), (2, 8, 256, 77, 160), (1, 16, 577, 577, 64), (1, 12, 16, 577, 64),
    (1, 1, 4096, 4096, 512 // 4)]


def timeit(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def timeit_graph(fn, n=20):
    """n launches back to back from a hipGraph (no host gaps): what the call costs inside a replayed step"""
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):  # this stream's workspaces exist before the capture
        fn()
        fn()
    st.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    g.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


def merge_table(k, dev):
    """option flash_merge: dQ and dK/dV blocks in one launch (2) against the separate kernels (0), flash_kt at its default"""
    T = torch.bfloat16
    print("# backward, dQ + dK/dV in ONE launch (flash_merge = 2) against the separate kernels (0); replayed from a hipGraph")
    for (B, H, Nq, Nk, d) in SHAPES + [(2, 8, 64, 77, 160), (1, 8, 4096, 4096, 40), (1, 8, 1024, 1024, 80), (1, 8, 256, 256, 160)]:
        HD = H * d
        q = torch.randn(B * Nq, HD, device=dev).to(T)
        kk = torch.randn(B * Nk, HD, device=dev).to(T)
        v = torch.randn(B * Nk, HD, device=dev).to(T)
        g = torch.randn(B * Nq, HD, device=dev).to(T)
        o = torch.empty_like(q)
        lse = torch.empty(B, H, Nq, device=dev)
        dbuf = torch.empty(B, H, Nq, device=dev)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(kk), torch.empty_like(v)
        k.flash_attn_fwd(q, kk, v, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
        row = []
        for mg in (0, 2):
            _hip.set_option("flash_merge", mg)
            row.append(timeit_graph(lambda: k.flash_attn_bwd(q, kk, v, o, g, lse, dbuf, dq, dk, dv, B, H, Nq, Nk, d, HD, HD, HD, HD,
                                                             d ** -0.5)))
        blocks = ((Nq + 127) // 128 + (Nk + 127) // 128) * B * H
        print(f"flash merge B={B} H={H:2d} Nq={Nq:4d} Nk={Nk:4d} d={d:3d}  separate {row[0]:7.1f} us   one launch {row[1]:7.1f} us   "
              f"({(Nq + 127) // 128 * B * H} dQ blocks + {(Nk + 127) // 128 * B * H} x qsplit dK/dV blocks, {blocks} unsplit)", flush=True)
    _hip.set_option("flash_merge", 1)


def qs_table(k, dev):
    """option flash_qs: target block count of the dK / dV query split (cross-attention: one key block per (batch, head))"""
    T = torch.bfloat16
    print("# backward of the cross-attention shapes, us per call replayed from a hipGraph, by flash_qs (blocks the query split aims at) x flash_merge")
    for (B, H, Nq, Nk, d) in [(2, 8, 4096, 77, 40), (2, 8, 1024, 77, 80), (2, 8, 256, 77, 160), (2, 8, 64, 77, 160), (2, 20, 1024, 77, 64), (2, 10, 4096, 77, 64)]:
        HD = H * d
        q = torch.randn(B * Nq, HD, device=dev).to(T)
        kk = torch.randn(B * Nk, HD, device=dev).to(T)
        v = torch.randn(B * Nk, HD, device=dev).to(T)
        g = torch.randn(B * Nq, HD, device=dev).to(T)
        o = torch.empty_like(q)
        lse = torch.empty(B, H, Nq, device=dev)
        dbuf = torch.empty(B, H, Nq, device=dev)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(kk), torch.empty_like(v)
        k.flash_attn_fwd(q, kk, v, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5)
        cells = []
        for qs in (64, 128, 256, 512, 1024):
            for mg in (0, 1, 2):
                _hip.set_option("flash_qs", qs)
                _hip.set_option("flash_merge", mg)
                t = timeit_graph(lambda: k.flash_attn_bwd(q, kk, v, o, g, lse, dbuf, dq, dk, dv, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5))
                cells.append(f"qs={qs:4d}/mg={mg} {t:6.1f}")
        print(f"B={B} H={H:2d} Nq={Nq:4d} Nk={Nk} d={d:3d}: " + "  ".join(cells), flush=True)
    _hip.set_option("flash_qs", 512)
    _hip.set_option("flash_merge", 1)


def main():
    dev = torch.device("cuda:0")
    k = _hip.HipKernels()
    ops.set_kernel_backend(k)
    T = torch.bfloat16
    if "qs" in sys.argv[1:]:
        return qs_table(k, dev)
    if "merge" in sys.argv[1:]:
        return merge_table(k, dev)
    if "kt" in sys.argv[1:]:
        return kt_table(k, dev)
    for (B, H, Nq, Nk, d) in SHAPES:
        HD = H * d
        q = torch.randn(B * Nq, HD, device=dev).to(T)
        kk = torch.randn(B * Nk, HD, device=dev).to(T)
        v = torch.randn(B * Nk, HD, device=dev).to(T)
        g = torch.randn(B * Nq, HD, device=dev).to(T)
        o = torch.empty_like(q)
        lse = torch.empty(B, H, Nq, device=dev)
        dbuf = torch.empty(B, H, Nq, device=dev)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(kk), torch.empty_like(v)
        sc = d ** -0.5
        for trim in (0, 1):
            for tr in (0, 2):  # never / always (the default, 1, picks by head dim)
                _hip.set_option("flash_trim", trim)
                _hip.set_option("flash_tr", tr)
                tf = timeit(lambda: k.flash_attn_fwd(q, kk, v, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, sc))
                tb = timeit(lambda: k.flash_attn_bwd(q, kk, v, o, g, lse, dbuf, dq, dk, dv, B, H, Nq, Nk, d, HD, HD, HD, HD, sc))
                fl = 4.0 * B * H * Nq * Nk * d
                print(f"flash B={B} H={H} Nq={Nq} Nk={Nk} d={d:3d} trim={trim} tr={tr}  fwd {tf:8.1f} us {fl / tf / 1e6:7.1f} TF/s   "
                      f"bwd {tb:8.1f} us {2.5 * fl / tb / 1e6:7.1f} TF/s", flush=True)
    _hip.set_option("flash_trim", 1)
    _hip.set_option("flash_tr", 1)
    kt_table(k, dev)


def kt_table(k, dev):
    T = torch.bfloat16
    _hip.set_option("flash_merge", 0)
    print("# two 32-row tiles per iteration (flash_kt: 2 forward, 3 + dQ, 5 + dK/dV) against one, default trim / tr")
    for (B, H, Nq, Nk, d) in SHAPES:  # (head dims above 96 have one-tile kernels only: their columns repeat)
        HD = H * d
        q = torch.randn(B * Nq, HD, device=dev).to(T)
        kk = torch.randn(B * Nk, HD, device=dev).to(T)
        v = torch.randn(B * Nk, HD, device=dev).to(T)
        g = torch.randn(B * Nq, HD, device=dev).to(T)
        o = torch.empty_like(q)
        lse = torch.empty(B, H, Nq, device=dev)
        dbuf = torch.empty(B, H, Nq, device=dev)
        dq, dk, dv = torch.empty_like(q), torch.empty_like(kk), torch.empty_like(v)
        row = []
        for kt in (1, 2, 3, 5):
            _hip.set_option("flash_kt", kt)
            tf = timeit(lambda: k.flash_attn_fwd(q, kk, v, o, lse, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5))
            tb = timeit(lambda: k.flash_attn_bwd(q, kk, v, o, g, lse, dbuf, dq, dk, dv, B, H, Nq, Nk, d, HD, HD, HD, HD, d ** -0.5))
            row.append((tf, tb))
        print(f"flash kt B={B} H={H} Nq={Nq} Nk={Nk} d={d:3d}  fwd {row[0][0]:7.1f} -> {row[1][0]:7.1f} us   "
              f"bwd kt=1 {row[0][1]:7.1f}  kt=3 (dQ) {row[2][1]:7.1f}  kt=5 (dQ + dK/dV) {row[3][1]:7.1f} us", flush=True)
    _hip.set_option("flash_kt", 4)
    _hip.set_option("flash_merge", 1)


if __name__ == "__main__":
    main()
