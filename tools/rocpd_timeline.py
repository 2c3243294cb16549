"""GPU busy / idle analysis of a rocprofv3 kernel trace (rocpd sqlite): union of kernel intervals vs wall time, idle-gap
histogram, per-queue busy time, and what the critical queue looks like.  Settles "host-bound or GPU-bound".
    python tools/rocpd_timeline.py <results.db> [skip_fraction]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    cur = sqlite3.connect(db).cursor()
    rows = list(cur.execute("select start, end, queue_id, name from kernels order by start"))
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    cut = t0 + (t1 - t0) * skip  # steady state only
    rows = [r for r in rows if r[0] >= cut]
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    wall = (t1 - t0) / 1e6
    busy, cur_end, gaps = 0, t0, []
    for s, e, q, n in rows:
        if s > cur_end:
            gaps.append((s - cur_end) / 1e3)
            busy += 0
            cur_s = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    ksum = sum(e - s for s, e, _, _ in rows) / 1e6
    print(f"window {wall:.1f} ms, {len(rows)} dispatches; union-busy {busy / 1e6:.1f} ms ({100 * busy / 1e6 / wall:.1f} %), "
          f"sum of kernel durations {ksum:.1f} ms, idle {wall - busy / 1e6:.1f} ms in {len(gaps)} gaps")
    import collections
    h = collections.Counter()
    tot = collections.Counter()
    for g in gaps:
        b = 1 if g < 2 else 5 if g < 5 else 10 if g < 10 else 20 if g < 20 else 50 if g < 50 else 100 if g < 100 else 1000
        h[b] += 1
        tot[b] += g
    for b in sorted(h):
        print(f"  gaps < {b:5d} us: {h[b]:6d}  total {tot[b] / 1e3:8.2f} ms")
    perq = collections.Counter()
    for s, e, q, n in rows:
        perq[q] += e - s
    for q, v in perq.most_common():
        print(f"  queue {q}: {v / 1e6:.1f} ms of kernels")


if __name__ == "__main__":
    main()
