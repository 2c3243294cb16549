"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into the text table kept under profiles/.
    python tools/rocpd_summary.py <results.db> [steps]  > profiles/<name>.txt
(rocprofv3 --kernel-trace --stats writes <name>_results.db; this prints per-kernel totals, like --stats's CSV.)"""
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    db = sys.argv[1]
    steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3, min(end-start)/1e3, "
                            "max(end-start)/1e3 from kernels group by name order by 3 desc"))
    tot = sum(r[2] for r in rows)
    n = sum(r[1] for r in rows)
    print(f"# rocprofv3 --kernel-trace summary of {db}")
    print(f"# total kernel time {tot:.2f} ms over {n} dispatches; {steps:g} bench steps in the trace "
          f"-> {tot / steps:.2f} ms of kernels per step")
    print(f"# {'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>9} {'min_us':>8} {'max_us':>9}  kernel")
    for name, cnt, ms, avg, mn, mx in rows:
        print(f"  {ms:10.2f} {100 * ms / tot:6.2f} {cnt:7d} {avg:9.1f} {mn:8.1f} {mx:9.1f}  {name[:150]}")
    # template instantiations of one kernel folded together (bench.py's roofline.avg_launch_ms is quoted per family)
    from pmc_to_json import family
    fam = {}
    for name, cnt, ms, avg, mn, mx in rows:
        f = fam.setdefault(family(name), [0, 0.0])
        f[0] += cnt
        f[1] += ms
    print("#\n# by kernel family (all template instantiations)")
    print(f"# {'total_ms':>10} {'pct':>6} {'calls':>7} {'avg_us':>9}  family")
    for name, (cnt, ms) in sorted(fam.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"  {ms:10.2f} {100 * ms / tot:6.2f} {cnt:7d} {ms / cnt * 1e3:9.1f}  {name}")


if __name__ == "__main__":
    main()
