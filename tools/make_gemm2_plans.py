"""tools/tune_gemm2.py output (JSON lines) -> comat_amd/csrc/gemm2_plans.inc (the static plan table of gemm2.hip).
    python tools/make_gemm2_plans.py [--base old_plans.inc] gpurun_out/g2_tune.jsonl [more.jsonl ...]
Per problem the fastest measured (tile, splits); among variants within 3 % of the fastest the one with the fewest slices
(less slab traffic, least sensitivity to what else runs on the chip)."""
import json
import re
import sys

rows = {}
args = sys.argv[1:]
if args and args[0] == "--base":  # keep the entries of an existing table (problems the given measurements do not cover)
    for m in re.finditer(r"^\s*\{(\d+), (\d+), (\d+), (\d+), (\d+), (\d+), (\d+)\},\s*// ([\d.]+) us \(general ([\d.]+)\), (\d+) calls",
                         open(args[1]).read(), re.M):
        v = m.groups()
        rows[tuple(int(x) for x in v[:5])] = (int(v[5]), int(v[6]), float(v[7]), int(v[9]), float(v[8]), args[1])
    args = args[2:]
for path in args:
    for line in open(path):
        if not line.startswith("{"):
            continue
        r = json.loads(line)
        key = ({"conv": 1, "gemm8": 2, "conv8": 3}.get(r["kind"], 0), r["M"], r["N"], r["nkt"], r["batch"])
        var = {k: v for k, v in r["us"].items() if ":" in k}
        best = min(var.values())
        ok = [(int(k.split(":")[1]), v, k) for k, v in var.items() if v <= best * 1.03]
        s, us, k = min(ok)
        cfg = int(k.split(":")[0])
        old = rows.get(key)
        if old is None or r["calls"] >= old[3] or path != old[5]:  # later files (newer measurements) win
            rows[key] = (cfg, s, us, max(r["calls"], old[3] if old else 0), r["us"].get("general", 0.0), path)
out = ["// gemm2_plans.inc - measured plans of the pipelined GEMM / conv kernel for the problems of the SD1.5 (C2) and SDXL (C4) steps:",
       "// tools/tune_gemm2.py on an MI355X -> tools/make_gemm2_plans.py.  {kind, M, N, k-tiles, batch, tile code, slices}; kind 0 GEMM, 1 conv,",
       "// 2 / 3 the same with fp8 operands (C5; k-tiles of 64 bytes either way)",
       "// tile codes: 1 128x128, 2 128x64, 3 256x128, 4 64x128, 6 64x64, 7 128x128 with 8 waves; 8..11 = 64x64, 128x64, 64x128, 128x128 with",
       "// 128-byte k-tiles; 12 256x256, 13 256x128 (wave tiles of 128x64, never split).  Trailing comment: us per launch with this plan,",
       "// with the general 64x64 kernel, calls per step.",
       "static const Plan2Entry g2_plans[] = {"]
for key in sorted(rows, key=lambda k: (-rows[k][3] * rows[k][2])):
    cfg, s, us, calls, gen, _ = rows[key]
    out.append(f"    {{{key[0]}, {key[1]}, {key[2]}, {key[3]}, {key[4]}, {cfg}, {s}}},  // {us:.1f} us (general {gen:.1f}), {calls} calls")
out.append("};")
open("comat_amd/csrc/gemm2_plans.inc", "w").write("\n".join(out) + "\n")
print(f"{len(rows)} plans; step total {sum(v[2] * v[3] for v in rows.values()) / 1e3:.1f} ms "
      f"(general kernel {sum(v[4] * v[3] for v in rows.values()) / 1e3:.1f} ms)")
