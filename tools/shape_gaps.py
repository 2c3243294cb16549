"""Where the MFMA kernels of a step lose their time: measured launch time against two floors, per problem.

    python tools/shape_gaps.py profiles/r03_z_bench_shapes.txt [--top 30] > profiles/r03_z_gap_table.txt

Input: the per-problem table `bench.py` writes under COMAT_BENCH_DUMP (in-step ms per step, calls, in-step us per launch,
replayed us per launch, problem signature).  For each problem:

    t_mfma = FLOP / dense peak (2.5 PFLOP/s bf16; 5 PFLOP/s for fp8 operands)
    t_hbm  = algorithmic bytes (every operand once, the output once) / 8 TB/s
    floor  = max(t_mfma, t_hbm)                       the roofline of the problem, whichever resource binds
    excess = calls x (measured - floor)               what a kernel AT its roofline would give back, per step

and a third column: the measured time of an EMPTY launch on this box (~2.2 us back to back in a graph, tools/mb_launch
numbers in DESIGN.md section 6) - problems whose floor is below it are launch-latency problems: no kernel change, only
fewer launches (grouping / fusion), gives their time back.  The table is sorted by excess; the header sums the step."""
import argparse
import re
import sys

PEAK = {"bf16": 2500e12, "fp8": 5000e12, "fp32": 2500e12 / 16}  # fp32 operands ride the bf16x3-free slow path: xf32 absent on gfx950
HBM = 8e12
EMPTY_LAUNCH_US = 2.2


def esz(dtype):
    return {"torch.bfloat16": 2, "torch.float32": 4, "torch.uint8": 1, "torch.float16": 2, "torch.float8_e4m3fn": 1}[dtype]


def parse(sig):
    """-> (flop, bytes, operand class) of one launch"""
    g = lambda pat: re.search(pat, sig)
    ints = lambda pat: tuple(int(v) for v in g(pat).groups())
    if sig.startswith("gemm_segments"):
        M, N = ints(r"M=(\d+) N=(\d+)")
        ks = [int(v) for v in g(r"K=([\d+]+)").group(1).split("+")]
        b = int(g(r" b=(\d+)").group(1))
        i, o = esz(g(r" in=(\S+)").group(1)), esz(g(r" out=(\S+)").group(1))
        r = M * N * o if " R=1" in sig else 0
        return 2.0 * M * N * sum(ks) * b, b * ((M + N) * sum(ks) * i + M * N * o + r), "fp8" if i == 1 else "bf16"
    if sig.startswith("gemm "):
        M, N, K = ints(r"M=(\d+) N=(\d+) K=(\d+)")
        b1, b2 = ints(r"b=\((\d+), (\d+)\)")
        i, o = esz(g(r" in=(\S+)").group(1)), esz(g(r" out=(\S+)").group(1))
        r = M * N * o if " R=1" in sig else 0
        cls = "fp8" if i == 1 else ("fp32" if i == 4 else "bf16")
        return 2.0 * M * N * K * b1 * b2, b1 * b2 * ((M + N) * K * i + M * N * o + r), cls
    if sig.startswith("conv"):
        B, Hi, Wi, Ci, Ho, Wo, Co, k, s = ints(r"B=(\d+) HWin=(\d+)x(\d+) Cin=(\d+) HWout=(\d+)x(\d+) Cout=(\d+) k=(\d+) s=(\d+)")
        mode = int(g(r"mode=(\d+)").group(1))
        i, o = esz(g(r" in=(\S+)").group(1)), esz(g(r" out=(\S+)").group(1))
        f = 2.0 * B * Ho * Wo * Co * k * k * Ci
        if mode == 1:
            f /= s * s
        r = B * Ho * Wo * Co * o if " R=1" in sig else 0
        return f, (B * Hi * Wi * Ci + Co * k * k * Ci) * i + B * Ho * Wo * Co * o + r, "fp8" if i == 1 else "bf16"
    if sig.startswith("flash_fwd") or sig.startswith("flash_bwd"):
        B, H, Nq, Nk, d = ints(r"B=(\d+) H=(\d+) Nq=(\d+) Nk=(\d+) d=(\d+)")
        e = esz(sig.split()[-1])
        if sig.startswith("flash_fwd"):  # q, k, v read, o written (+ the fp32 row statistics)
            return 4.0 * B * H * Nq * Nk * d, B * H * ((2 * Nq + 2 * Nk) * d * e + Nq * 4), "bf16"
        # q, k, v, o, dO read; dq, dk, dv written
        return 10.0 * B * H * Nq * Nk * d, B * H * ((5 * Nq + 4 * Nk) * d * e + Nq * 4), "bf16"
    if sig.startswith("tt_grouped"):
        f = by = 0.0
        for m, n, k, c in re.findall(r"(\d+)x(\d+)x(\d+)\*(\d+)", sig):
            m, n, k, c = int(m), int(n), int(k), int(c)
            f += 2.0 * m * n * k * c
            by += ((m + n) * k * 2 + 2 * m * n * 4) * c
        return f, by, "bf16"
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("table")
    ap.add_argument("--top", type=int, default=30)
    args = ap.parse_args()
    rows = []
    for line in open(args.table):
        if line.startswith("#") or not line.strip():
            continue
        p = line.split(None, 6)
        ms, calls, us, us_rep, kern, sig = float(p[0]), int(p[1]), float(p[3]), float(p[4]), p[5], p[6].strip()
        got = parse(sig)
        if got is None:
            continue
        f, by, cls = got
        t_m, t_h = f / PEAK[cls] * 1e6, by / HBM * 1e6
        rows.append(dict(ms=ms, calls=calls, us=us, us_rep=us_rep, kern=kern, sig=sig, t_m=t_m, t_h=t_h, floor=max(t_m, t_h),
                         bound="mfma" if t_m >= t_h else "hbm"))
    tot = sum(r["ms"] for r in rows)
    fl = sum(r["calls"] * r["floor"] for r in rows) / 1e3
    lat = sum(r["calls"] * max(r["floor"], EMPTY_LAUNCH_US) for r in rows) / 1e3
    n = sum(r["calls"] for r in rows)
    sub = [r for r in rows if r["floor"] < EMPTY_LAUNCH_US]
    print(f"# {args.table}: {n} MFMA-kernel launches per step, {tot:.1f} ms measured in step")
    print(f"#   at every problem's own roofline (max of FLOP / dense peak, algorithmic bytes / 8 TB/s): {fl:.1f} ms "
          f"({fl / tot:.1%} of measured)")
    print(f"#   same, but no launch shorter than an empty launch ({EMPTY_LAUNCH_US} us): {lat:.1f} ms")
    print(f"#   problems whose roofline time is BELOW an empty launch: {sum(r['calls'] for r in sub)} launches, "
          f"{sum(r['ms'] for r in sub):.1f} ms measured - only fewer launches give these back")
    for name, pred in (("under 2 us", lambda r: r["floor"] < 2), ("2-10 us", lambda r: 2 <= r["floor"] < 10),
                       ("10 us and more", lambda r: r["floor"] >= 10)):
        s = [r for r in rows if pred(r)]
        if s:
            m, f_ = sum(r["ms"] for r in s), sum(r["calls"] * r["floor"] for r in s) / 1e3
            print(f"#   roofline time {name:15s}: {sum(r['calls'] for r in s):5d} launches, measured {m:6.1f} ms, floor {f_:6.1f} ms, "
                  f"measured / floor {m / max(f_, 1e-9):5.1f}x")
    print("# excess ms/step  calls  measured us  replayed us  floor us  bound  x floor  problem")
    rows.sort(key=lambda r: -(r["ms"] - r["calls"] * r["floor"] / 1e3))
    for r in rows[:args.top]:
        ex = r["ms"] - r["calls"] * r["floor"] / 1e3
        print(f"{ex:10.2f} {r['calls']:6d} {r['us']:10.1f} {r['us_rep']:10.1f} {r['floor']:9.2f}  {r['bound']:5s} "
              f"{r['us'] / max(r['floor'], 1e-9):6.1f}  {r['sig'][:120]}")


if __name__ == "__main__":
    sys.exit(main())
