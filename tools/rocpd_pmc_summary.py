"""Per-kernel averages of the PMC counters in rocprofv3 rocpd databases (one pass per file), with kernel durations.
    python tools/rocpd_pmc_summary.py a_results.db b_results.db ... > profiles/<name>.txt"""
import sqlite3
import sys
from collections import defaultdict


def main():
    out = defaultdict(dict)
    dur = {}
    for db in sys.argv[1:]:
        cur = sqlite3.connect(db).cursor()
        for k, c, v, n in cur.execute("select kernel_name, counter_name, avg(value), count(*) from counters_collection "
                                      "group by kernel_name, counter_name"):
            out[k][c] = (v, n)
        for k, d, n in cur.execute("select name, avg(end-start), count(*) from kernels group by name"):
            dur.setdefault(k, (d, n))
    print("# per-kernel PMC averages (rocprofv3 --pmc, separate passes for FETCH_SIZE / WRITE_SIZE); avg duration in us")
    for k in sorted(out, key=lambda k: -dur.get(k, (0, 0))[0]):
        if "comat" not in k and "anonymous" not in k:
            continue
        d, n = dur.get(k, (0.0, 0))
        print(f"{k[:120]}\n    launches={n} avg_us={d / 1e3:.1f}")
        for c, (v, m) in sorted(out[k].items()):
            print(f"    {c:32s} {v:.4e}")


if __name__ == "__main__":
    main()
