"""Per (kernel, grid size) averages of every counter in rocprofv3 rocpd databases, in first-launch order.
    python tools/pmc_dump.py a_results.db [b_results.db ...] [--match gemm2] > profiles/<name>.txt
Derived columns where their inputs are present: MFMA busy share, waves per launch, wait / active shares of wave cycles."""
import sqlite3
import sys
from collections import defaultdict


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    match = None
    if "--match" in sys.argv:
        match = sys.argv[sys.argv.index("--match") + 1]
        args = [a for a in args if a != match]
    rows = defaultdict(dict)
    meta = {}
    for db in args:
        cur = sqlite3.connect(db).cursor()
        for k, gs, c, v, st, d, n in cur.execute(
                "select kernel_name, grid_size, counter_name, avg(value), min(start), avg(duration), count(*) "
                "from counters_collection group by kernel_name, grid_size, counter_name"):
            if match and match not in k:
                continue
            rows[(k, gs)][c] = v
            m = meta.setdefault((k, gs), [st, d / 1e3, n])
            m[0] = min(m[0], st)
    for key in sorted(rows, key=lambda kk: meta[kk][0]):
        k, gs = key
        st, us, n = meta[key]
        c = rows[key]
        short = k.replace("(anonymous namespace)::", "").replace("void ", "")[:90]
        print(f"{short}  grid_threads={gs} launches={n} avg_us={us:.1f}")
        for name in sorted(c):
            print(f"    {name:36s} {c[name]:.5e}")
        d = []
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c and c["GRBM_GUI_ACTIVE"]:
            d.append(f"mfma_busy={c['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024.0 * c['GRBM_GUI_ACTIVE'] / 8.0):.3f}")
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"]:
            wc = c["SQ_WAVE_CYCLES"]
            for nm in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_LDS",
                       "SQ_WAIT_INST_LDS"):
                if nm in c:
                    d.append(f"{nm[3:].lower()}/wave_cycles={c[nm] / wc:.3f}")
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]):
            d.append(f"l2_hit={c['TCC_HIT_sum'] / (c['TCC_HIT_sum'] + c['TCC_MISS_sum']):.3f}")
        if d:
            print("    -> " + "  ".join(d))


if __name__ == "__main__":
    main()
