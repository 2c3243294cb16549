"""Why does the pipelined GEMM / conv kernel move 2.8x its algorithmic bytes?  A model of the L2-miss traffic of its tile order.

    python tools/xcd_traffic_model.py profiles/r03_z_bench_shapes.txt

MI355X has 8 XCDs with PRIVATE L2s; the kernel hands each XCD one contiguous chunk of the work items
`((z * tiles_m + tm) * tiles_n + tn) * splits + sp` (gemm_shared.h: xcd_chunk_map), i.e. a range of output tiles in
row-block-major order.  Whatever operand tiles a chunk touches are fetched into THAT XCD's L2 once (the chunks' working sets
fit: <= a few MB); an operand tile touched by k chunks is fetched k times.  So the fabric-side read traffic of a launch is

    sum over XCDs of  (distinct row blocks in its chunk) x |A tile panel|  +  (distinct column blocks) x |B tile panel|

against the algorithmic |A| + |B|.  `FETCH_SIZE` counts exactly these L2 misses (whether HBM or the 256 MB Infinity Cache
serves them), which is what `roofline.traffic` reports.  The model below evaluates that sum for every GEMM / K-segmented GEMM /
conv problem of a step (128 x 128 tiles, or 64 x 64 where a dimension is below 128; split-K slices of one tile share its
operands' rows / columns but not their k-range, so they do not add re-reads) for the kernel's order and for the
column-block-major order, and reports what a per-problem choice of the better one would save.

Round 3 reading (profiles/r03_z_xcd_traffic_model.txt): the model gives 31.4 MB per launch for the kernel's order against 40.4
MB measured and 14.2 MB algorithmic, i.e. it accounts for two thirds of the excess (the rest: 64-wide tiles where the plan
table picks them, conv halos, evictions).  The modelled excess is overwhelmingly WEIGHT panels fetched by several XCDs in the
short, wide problems of the deep UNet levels and of the feed-forward projections (M = 512 .. 2048 rows against N = 1280 ..
10240 columns: a 16 x 16-level 3x3 conv moves 123 MB for 32 MB of operands, its 29.5 MB of weights four times), where
row-block-major order gives every XCD a slice of ALL columns.  Column-block-major order for those problems (each XCD: all row
blocks of its share of the columns) halves their traffic (modelled average 31.4 -> 24.6 MB; 834 launches, 23.8 ms of the
step); it is a 3-line change of the index decode with bit-identical results - the next round's first kernel experiment
(DESIGN.md section 9)."""
import re
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
import shape_gaps as sg  # noqa: E402


def problem(sig):
    """-> (M, N, K, batch, a_bytes_per_row_block(BM), b_bytes_per_col_block(BN), c_bytes) or None"""
    g = lambda pat: re.search(pat, sig)
    if sig.startswith("gemm_segments"):
        M, N = (int(v) for v in g(r"M=(\d+) N=(\d+)").groups())
        K = sum(int(v) for v in g(r"K=([\d+]+)").group(1).split("+"))
        b = int(g(r" b=(\d+)").group(1))
    elif sig.startswith("gemm "):
        M, N, K = (int(v) for v in g(r"M=(\d+) N=(\d+) K=(\d+)").groups())
        b1, b2 = (int(v) for v in g(r"b=\((\d+), (\d+)\)").groups())
        b = b1 * b2
        if "tA=1" in sig or sg.esz(g(r" in=(\S+)").group(1)) != 2:
            return None
    elif sig.startswith("conv"):
        B, Hi, Wi, Ci, Ho, Wo, Co, k, s = (int(v) for v in g(
            r"B=(\d+) HWin=(\d+)x(\d+) Cin=(\d+) HWout=(\d+)x(\d+) Cout=(\d+) k=(\d+) s=(\d+)").groups())
        M, N, K, b = B * Ho * Wo, Co, k * k * Ci, 1
        # the A operand of a conv is the activation: a row block's k*k taps overlap, so its fetch is ~ the block's own pixels
        # (+ halo) rather than BM x K - count BM x Cin x (1 + halo) with a 25 % halo for 3x3
        return M, N, K, b, ("conv", Ci, k), 2 * M * N
    else:
        return None
    return M, N, K, b, None, 2 * M * N


def traffic(M, N, K, conv, tiles_first):
    """modelled L2-miss READ bytes of one launch with the items of each XCD chunk ordered tiles_first = 'n' (the kernel:
    column blocks fastest, row-block-major) or 'm' (row blocks fastest, column-block-major)"""
    BM = 128 if M >= 128 else 64
    BN = 128 if N >= 128 else 64
    tm, tn = -(-M // BM), -(-N // BN)
    a_row = (conv[1] * 2 * (1.25 if conv[2] == 3 else 1.0)) if conv else K * 2   # bytes of one A row
    b_col = K * 2                                                                  # bytes of one B column (a weight row)
    span = lambda lo, hi, blk, n: min(n, (hi + 1) * blk) - lo * blk                # real rows / columns of blocks lo..hi
    items = tm * tn
    q, r = divmod(items, 8)
    total, start = 0.0, 0
    for x in range(8):
        n_items = q + (1 if x < r else 0)
        if n_items == 0:
            continue
        lo, hi = start, start + n_items - 1
        start += n_items
        if tiles_first == "n":  # lin = tm_i * tn + tn_i
            rows = span(lo // tn, hi // tn, BM, M)
            cols = N if hi // tn > lo // tn else span(lo % tn, hi % tn, BN, N)
        else:                   # lin = tn_i * tm + tm_i
            cols = span(lo // tm, hi // tm, BN, N)
            rows = M if hi // tm > lo // tm else span(lo % tm, hi % tm, BM, M)
        total += rows * a_row + cols * b_col
    return total


def main():
    rows = []
    for line in open(sys.argv[1]):
        if line.startswith("#") or not line.strip():
            continue
        p = line.split(None, 6)
        ms, calls, us, kern, sig = float(p[0]), int(p[1]), float(p[3]), p[5], p[6].strip()
        if not kern.startswith("gemm2_kernel"):
            continue
        pr = problem(sig)
        if pr is None:
            continue
        M, N, K, b, conv, c_bytes = pr
        algo = sg.parse(sig)[1]
        t_n, t_m = b * traffic(M, N, K, conv, "n"), b * traffic(M, N, K, conv, "m")
        r_bytes = b * M * N * 2 if " R=1" in sig else 0
        rows.append(dict(ms=ms, calls=calls, us=us, sig=sig, algo=algo, now=t_n + b * c_bytes + r_bytes,
                         best=min(t_n, t_m) + b * c_bytes + r_bytes, swap=t_m < 0.9 * t_n))
    n = sum(r["calls"] for r in rows)
    algo = sum(r["calls"] * r["algo"] for r in rows) / n
    now = sum(r["calls"] * r["now"] for r in rows) / n
    best = sum(r["calls"] * r["best"] for r in rows) / n
    print(f"# {sys.argv[1]}: {n} launches of the pipelined kernel per step")
    print(f"#   algorithmic bytes per launch (every operand once): {algo / 1e6:6.1f} MB")
    print(f"#   modelled L2-miss bytes per launch, the kernel's tile order (row-block-major chunks): {now / 1e6:6.1f} MB   "
          f"(measured under rocprofv3 --pmc: 40.4 MB, profiles/r03_pmc_kernels.json)")
    print(f"#   modelled, with the better of the two orders chosen per problem:               {best / 1e6:6.1f} MB")
    sw = [r for r in rows if r["swap"]]
    print(f"#   problems where column-block-major order cuts the modelled traffic by > 10 %: {len(sw)} kinds, "
          f"{sum(r['calls'] for r in sw)} launches, {sum(r['ms'] for r in sw):.1f} ms of the step")
    print("# saved MB/launch  calls  us/launch  algorithmic MB  modelled now  modelled best  problem")
    for r in sorted(sw, key=lambda r: -(r["now"] - r["best"]) * r["calls"])[:25]:
        print(f"{(r['now'] - r['best']) / 1e6:10.1f} {r['calls']:6d} {r['us']:9.1f} {r['algo'] / 1e6:12.1f} {r['now'] / 1e6:12.1f} "
              f"{r['best'] / 1e6:12.1f}   {r['sig'][:110]}")


def validate():
    """the model against the two GEMM-family probes of the counter passes (profiles/r03_pmc_kernels.json, tools/pmc_targets.py)"""
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r03_pmc_kernels.json")
    if not os.path.exists(path):
        return
    d = json.load(open(path))
    probes = {"conv 3x3 B=2 64x64 320->320": (8192, 320, 2880, ("conv", 320, 3)),
              "GEGLU projection 8192x2560x320": (8192, 2560, 320, None)}
    print("# model against measured FETCH_SIZE x 2 of the counter probes (reads only):")
    for k, v in d.items():
        if isinstance(v, dict) and v.get("probe") in probes:
            M, N, K, conv = probes[v["probe"]]
            mod = traffic(M, N, K, conv, "n")
            print(f"#   {v['probe']:34s} measured {v['hbm_read_bytes'] / 1e6:6.1f} MB   modelled {mod / 1e6:6.1f} MB   "
                  f"algorithmic reads {((M * (conv[1] if conv else K)) + N * K) * 2 / 1e6:6.1f} MB")


if __name__ == "__main__":
    main()
    validate()
