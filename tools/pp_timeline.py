"""Cycle timeline of the ping-pong main loop (diagnostic build: `make -C comat_amd/csrc timeline`, run on the GPU box):
    COMAT_LIB_PATH=comat_amd/lib/libcomat_hip_tl.so python tools/pp_timeline.py
Waves 0 (group 0) and 4 (group 1) of workgroup 0 stamp s_memtime (100 MHz-independent shader clock ticks) at the start of every
MFMA part (S), after its last MFMA issued (E) and behind the barrier that ends it (B).  Printed per phase: memory part incl. its
barrier wait (S - previous B), MFMA part (E - S), closing barrier (B - E)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
from comat_amd import _hip, ops  # noqa: E402

TL_MAX = 1536


def stamps():
    buf = (C.c_uint * (2 * TL_MAX))()
    lib = C.CDLL(os.environ["COMAT_LIB_PATH"])
    rc = lib.cmt_dbg_pp_timeline(buf)
    assert rc == 0, rc
    return [list(buf[:TL_MAX]), list(buf[TL_MAX:])]


def report(name, cfg):
    torch.cuda.synchronize()
    tl = stamps()
    n = tl[0][TL_MAX - 1]
    print(f"== {name} cfg={cfg}: {n} stamps per wave")
    for gi in (0, 1):
        t = tl[gi][:min(n, TL_MAX - 1)]
        d = lambda a, b: (b - a) & 0xffffffff
        print(f" group {gi}: start->loop {d(t[0], t[1])}  total {d(t[0], t[-1])}  finish {d(t[-2], t[-1])}")
        ph = t[2:-2]
        rows = []
        prevB = t[1]
        for i in range(0, len(ph) - 6, 7):  # per phase: reads issued, DMA issued, vmcnt passed, barrier passed, fragments landed, MFMAs issued, closing barrier
            d1, d2, d3, d4, S, E, B = ph[i:i + 7]
            if d2 == 0 or d(d1, d2) > 1 << 30:
                d2 = d1  # a tail phase without DMA keeps the previous stamp
            rows.append((d(prevB, d1), d(d1, d2), d(d2, d3), d(d3, d4), d(d4, S), d(S, E), d(E, B)))
            prevB = B
        k = len(rows)
        print("  phase: reads | dma issue | vmcnt | barrier 1 | lgkmcnt | mfma | barrier 2   (first 12, then every 8th)")
        for i, r_ in enumerate(rows):
            if i < 12 or i % 8 == 0 or i >= k - 4:
                print("   %4d: " % i + " ".join("%6d" % v for v in r_))
        mid = rows[8:-4] if k > 16 else rows
        if mid:
            m = [sum(r_[c] for r_ in mid) / len(mid) for c in range(7)]
            print("  mean of the middle phases: reads %.0f  dma issue %.0f  vmcnt %.0f  barrier-1 %.0f  lgkmcnt %.0f  mfma %.0f  barrier-2 %.0f  (sum %.0f ticks per phase)"
                  % (*m, sum(m)))


def main():
    dev = torch.device("cuda:0")
    k = _hip.HipKernels()
    ops.set_kernel_backend(k)
    T = torch.bfloat16
    r = lambda *s: (torch.rand(*s, device=dev) * 2 - 1).to(T)
    for (M, N, K, cfg) in ((8192, 1280, 1280, 13), (8192, 320, 1280, 13), (128, 128, 2560, 13), (8192, 1280, 1280, 12), (256, 256, 2560, 12)):
        a, b = r(M, K), r(N, K) * K ** -0.5
        c = torch.empty((M, N), dtype=T, device=dev)
        _hip.set_option("g2_cfg", cfg)
        _hip.set_option("g2_splits", 1)
        for _ in range(3):
            k.gemm(a, b, c, M, N, K, K, K, N)
        report(f"gemm {M}x{N}x{K}", cfg)
    for (B, H, W, Cin, Cout, cfg) in ((1, 256, 256, 256, 256, 12), (2, 64, 64, 320, 320, 13)):
        x, w = r(B * H * W, Cin), r(Cout, 3, 3, Cin) * (9 * Cin) ** -0.5
        y = torch.empty((B * H * W, Cout), dtype=T, device=dev)
        _hip.set_option("g2_cfg", cfg)
        for _ in range(3):
            k.conv2d(x, w, y, B, H, W, Cin, H, W, Cout, 3, 3, 1, 1, mode=0, ups=1)
        report(f"conv {B}x{H}x{W} {Cin}->{Cout}", cfg)


if __name__ == "__main__":
    main()
