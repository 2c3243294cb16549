"""Per-shape plan tuning of the pipelined GEMM / conv kernel (gemm2.hip) on the GPU box.

    python tools/tune_gemm2.py c2 [c4 ...]  > gpurun_out/g2_tune.jsonl

1. builds the bench world of each configuration, runs one eager step with a recorder around the kernel backend and
   collects the distinct GEMM / K-segmented GEMM / conv problems of the step with their call counts;
2. replays every problem the pipelined kernel accepts on synthetic operands under each block tile (g2_cfg) and split
   count, 10 timed launches per variant (HIP events);
3. prints one JSON line per problem: key (kind, M, N, k-tiles, batch), calls per step, microseconds per variant, best.
tools/make_gemm2_plans.py turns the output into comat_amd/csrc/gemm2_plans.inc (the static plan table)."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
os.environ["COMAT_STEP_GRAPH"] = "0"
import bench  # noqa: E402
from comat_amd import _hip, ops  # noqa: E402

CFGS = {1: (128, 128), 2: (128, 64), 3: (256, 128), 4: (64, 128), 6: (64, 64), 7: (128, 128),
        8: (64, 64), 9: (128, 64), 10: (64, 128), 11: (128, 128),  # 8..11: 128-byte k-tiles
        12: (256, 256), 13: (256, 128)}  # wave tiles of 128 x 64, never split


class Recorder:
    def __init__(self, inner):
        self.inner, self.seen = inner, {}

    def __getattr__(self, name):
        fn = getattr(self.inner, name)
        if name not in ("gemm", "gemm_segments", "conv2d"):
            return fn

        def wrapped(*a, **kw):
            sig = None
            if name == "gemm" and a[0].dtype == torch.uint8:  # fp8 operands (C5): 64-byte k-tiles = 64 elements (a GEGLU epilogue is timed as a plain one)
                b = kw.get("batch", (1, 1))
                kt = kw.get("ktail")
                sig = ("gemm8", a[3], a[4], a[5] // 64, b[0], (a[5], kt[2] if kt is not None else 0), kw.get("R") is not None,
                       str(a[2].dtype))
            elif name == "conv2d" and a[0].dtype == torch.uint8 and kw.get("mode", 0) == 0:
                B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad = a[3:14]
                sig = ("conv8", B * Hout * Wout, Cout, KH * KW * Cin // 64, 1,
                       (B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, kw.get("ups", 1)), kw.get("R") is not None,
                       str(a[2].dtype))
            elif name == "gemm" and a[0].dtype == torch.bfloat16 and not kw.get("transA") and not kw.get("transB"):
                b = kw.get("batch", (1, 1))
                if b[1] == 1 and a[5] % 32 == 0 and a[3] >= 48:
                    # (a[2] is None for the GEGLU epilogue of a no-grad call: no pre-activations are stored)
                    sig = ("gemm", a[3], a[4], a[5] // 32, b[0], a[5], kw.get("R") is not None,
                           str(a[2].dtype) if a[2] is not None else "torch.bfloat16")
            elif name == "gemm_segments" and a[0][0][0].dtype == torch.bfloat16 and a[2] >= 48:
                ks = tuple(sg[2] for sg in a[0])
                if all(k_ % 32 == 0 for k_ in ks):
                    sig = ("seg", a[2], a[3], sum(ks) // 32, kw.get("batch", 1), ks, kw.get("R") is not None, str(a[1].dtype))
            elif name == "conv2d" and a[0].dtype == torch.bfloat16 and kw.get("mode", 0) == 0 and a[6] % 32 == 0:
                B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad = a[3:14]
                if B * Hout * Wout >= 48:
                    sig = ("conv", B * Hout * Wout, Cout, KH * KW * Cin // 32, 1,
                           (B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, kw.get("ups", 1)),
                           kw.get("R") is not None, str(a[2].dtype))
            if sig is not None:
                self.seen[sig] = self.seen.get(sig, 0) + 1
            return fn(*a, **kw)
        return wrapped


_side = None


def timeit(fn, n=16):
    """microseconds per launch: n back-to-back launches replayed from a hipGraph (an eager loop measures the host's
    ~6 us launch cadence for every kernel shorter than that)"""
    global _side
    if _side is None:
        _side = torch.cuda.Stream()
    with torch.cuda.stream(_side):
        fn()
        fn()
    _side.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=_side):
        for _ in range(n):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(2):  # best of two replays: a minimum over ~40 noisy variants otherwise picks the luckiest, not the fastest
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        g.replay()
        e.record()
        torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / n * 1e3)
    return best


def make_call(k, sig, dev):
    T = torch.bfloat16
    kind, M, N, nkt, batch, extra, has_r, out_dt = sig
    odt = torch.float32 if "float32" in out_dt else T
    r = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(T)
    r8 = lambda *s: torch.randint(0, 120, s, device=dev, dtype=torch.uint8)  # e4m3 bytes of small positive values
    one = torch.ones(max(batch, 1), device=dev)
    if kind == "gemm8":
        K, K2 = extra
        a, b = r8(M, K), r8(batch, N, K)
        c = torch.empty((batch, M, N), dtype=odt, device=dev)
        R = torch.zeros_like(c[0]) if has_r else None
        kt = (r(M, batch * K2), r(batch, N, K2), K2, batch * K2, K2, K2, N * K2) if K2 else None
        return lambda: k.gemm(a, b, c, M, N, K, K, K, N, R=R, ldr=N, beta=1.0 if has_r else 0.0, batch=(batch, 1), sB=(N * K, 0),
                              sC=(M * N, 0), scales=(one, one, 1), ktail=kt)
    if kind == "conv8":
        B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, ups = extra
        x, w = r8(B * Hin * Win, Cin), r8(Cout, KH, KW, Cin)
        y = torch.empty((B * Hout * Wout, Cout), dtype=odt, device=dev)
        R = torch.zeros_like(y) if has_r else None
        bias = torch.zeros(Cout, device=dev)
        return lambda: k.conv2d(x, w, y, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, mode=0, ups=ups, bias=bias, R=R,
                                beta=1.0 if has_r else 0.0, scales=(one, one))
    if kind == "gemm":
        K = extra
        a, b = r(batch, M, K), r(batch, N, K)
        c = torch.empty((batch, M, N), dtype=odt, device=dev)
        R = torch.zeros_like(c) if has_r else None
        return lambda: k.gemm(a, b, c, M, N, K, K, K, N, R=R, ldr=N, beta=1.0 if has_r else 0.0, batch=(batch, 1),
                              sA=(M * K, 0), sB=(N * K, 0), sC=(M * N, 0), sR=(M * N, 0))
    if kind == "seg":
        segs = [(r(batch, M, K_), r(batch, N, K_), K_, K_, K_, M * K_, N * K_) for K_ in extra]
        c = torch.empty((batch, M, N), dtype=odt, device=dev)
        R = torch.zeros_like(c) if has_r else None
        return lambda: k.gemm_segments(segs, c, M, N, N, R=R, ldr=N, beta=1.0 if has_r else 0.0, batch=batch, sC=M * N,
                                       sR=M * N)
    B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, ups = extra
    x, w = r(B * Hin * Win, Cin), r(Cout, KH, KW, Cin)
    y = torch.empty((B * Hout * Wout, Cout), dtype=odt, device=dev)
    R = torch.zeros_like(y) if has_r else None
    bias = torch.zeros(Cout, device=dev)
    return lambda: k.conv2d(x, w, y, B, Hin, Win, Cin, Hout, Wout, Cout, KH, KW, stride, pad, mode=0, ups=ups, bias=bias, R=R,
                            beta=1.0 if has_r else 0.0)


def main():
    dev = torch.device("cuda:0")
    k = _hip.HipKernels()
    seen = {}
    for cfg_name in sys.argv[1:] or ["c2"]:
        ops.set_kernel_backend(k)
        world, bs = (cfg_name[:-3], 4) if cfg_name.endswith("bs4") else (cfg_name, 1)  # "c2bs4": the batch-4 secondary line
        trainer, batch, fixed, scfg, _, _ = bench.build_world(dev, torch.bfloat16, 0, world, bs=bs)
        trainer.train_step(batch, **fixed)
        rec = Recorder(k)
        ops.set_kernel_backend(rec)
        trainer.train_step(batch, **fixed)
        torch.cuda.synchronize()
        ops.set_kernel_backend(k)
        for sig, n in rec.seen.items():
            seen[sig] = max(seen.get(sig, 0), n)
        del trainer, batch
        torch.cuda.empty_cache()
    if os.environ.get("TUNE_NEW_ONLY"):  # only the problems the plan table does not hold yet (a step whose shapes moved)
        import re
        have = set()
        for m in re.finditer(r"^\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), ", open(os.path.join(ROOT, "comat_amd", "csrc", "gemm2_plans.inc")).read(), re.M):
            have.add(tuple(int(v) for v in m.groups()))
        n0 = len(seen)
        code = {"gemm": 0, "seg": 0, "conv": 1, "gemm8": 2, "conv8": 3}
        seen = {sig: n for sig, n in seen.items() if (code[sig[0]], sig[1], sig[2], sig[3], sig[4]) not in have}
        print(f"# {n0} distinct problems, {len(seen)} of them not in gemm2_plans.inc", file=sys.stderr, flush=True)
    print(f"# {len(seen)} distinct problems", file=sys.stderr, flush=True)
    top = int(os.environ.get("TUNE_TOP", "0")) or len(seen)
    for sig, calls in sorted(seen.items(), key=lambda kv: -kv[1])[:top]:
        kind, M, N, nkt, batch = sig[:5]
        call = make_call(k, sig, dev)
        res = {}
        _hip.set_option("gemm2", 1)

        def sweep(c):
            bm, bn = CFGS[c]
            blocks = -(-M // bm) * -(-N // bn) * batch
            best = None
            for s in (1, 2, 3, 4, 6, 8, 12, 16):
                if s > 1 and (nkt // s < 8 or blocks * s > 1536 or blocks >= 512 or c >= 12):
                    continue
                _hip.set_option("g2_cfg", c)
                _hip.set_option("g2_splits", s)
                res[f"{c}:{s}"] = t = round(timeit(call), 2)
                best = t if best is None else min(best, t)
            return best

        k2 = {}
        for c in (1, 2, 3, 4, 6, 7):
            if c == 3 and -(-M // 256) * -(-N // 128) * batch < 64:  # 256 x 128: only with enough tiles (round 4: it wins at
                continue                                            # 512 x 10240 x 1280, 24.8 vs 32.1 us)
            k2[c] = sweep(c)
        for c in (12, 13):  # the 128 x 64 wave tiles: only where they get at least half a chip of blocks
            bm, bn = CFGS[c]
            if -(-M // bm) * -(-N // bn) * batch >= 120:
                k2[c] = sweep(c)
        floor = min(k2.values())
        for c4, twin in ((8, 6), (9, 2), (10, 4), (11, 1)):  # 128-byte k-tiles: only where the twin is in the running
            if twin in k2 and k2[twin] <= 1.3 * floor and floor < 60.0 and not kind.endswith("8"):  # (bf16 only)
                sweep(c4)
        if os.environ.get("TUNE_GENERAL", "0") != "0":  # the register-staged 64 x 64 kernel, for the table's comment column
            _hip.set_option("gemm2", 0)
            res["general"] = round(timeit(call), 2)
            _hip.set_option("gemm2", 1)
        _hip.set_option("g2_cfg", 0)
        _hip.set_option("g2_splits", 0)
        res["auto"] = round(timeit(call), 2)
        best = min((v, kk) for kk, v in res.items() if ":" in kk)
        print(json.dumps(dict(kind=kind, M=M, N=N, nkt=nkt, batch=batch, calls=calls, extra=sig[5], best=best[1],
                              best_us=best[0], us=res)), flush=True)
        del call
    _hip.set_option("g2_cfg", 0)
    _hip.set_option("g2_splits", 0)


if __name__ == "__main__":
    main()
