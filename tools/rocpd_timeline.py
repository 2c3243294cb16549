"""GPU busy / idle analysis of a rocprofv3 kernel trace (rocpd sqlite): union of kernel intervals vs wall time, idle-gap
histogram, per-queue busy time and gaps, how many kernels run concurrently, and WHICH kernels own the wall time
(every instant of the window is shared equally among the kernels running at that instant: the per-family sums add up to
the busy time, so a family that only ever runs in the shadow of another stream's kernels gets half of its duration).
Settles "host-bound or GPU-bound" and shows what the critical path of a graph-replayed step is made of.
    python tools/rocpd_timeline.py <results.db> [skip_fraction] [steps_in_window]"""
import collections
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def main():
    db = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    steps = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
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
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end = e
    ksum = sum(e - s for s, e, _, _ in rows) / 1e6
    print(f"window {wall:.1f} ms, {len(rows)} dispatches; union-busy {busy / 1e6:.1f} ms ({100 * busy / 1e6 / wall:.1f} %), "
          f"sum of kernel durations {ksum:.1f} ms, idle {wall - busy / 1e6:.1f} ms in {len(gaps)} gaps")
    if steps:
        print(f"per step ({steps:g} steps in the window): wall {wall / steps:.1f} ms, busy {busy / 1e6 / steps:.1f} ms, "
              f"kernel sum {ksum / steps:.1f} ms, {len(rows) / steps:.0f} dispatches")

    def hist(gs, indent="  "):
        h, tot = collections.Counter(), collections.Counter()
        for g in gs:
            b = 1 if g < 1 else 2 if g < 2 else 5 if g < 5 else 10 if g < 10 else 20 if g < 20 else 50 if g < 50 else \
                100 if g < 100 else 1000 if g < 1000 else 10 ** 9
            h[b] += 1
            tot[b] += g
        for b in sorted(h):
            print(f"{indent}gaps < {b:10d} us: {h[b]:6d}  total {tot[b] / 1e3:8.2f} ms")

    print("all queues together (GPU idle):")
    hist(gaps)
    perq = collections.defaultdict(list)
    for r in rows:
        perq[r[2]].append(r)
    print("per queue:")
    for q, rs in sorted(perq.items(), key=lambda kv: -sum(e - s for s, e, _, _ in kv[1])):
        b = sum(e - s for s, e, _, _ in rs) / 1e6
        qg = [(rs[i + 1][0] - rs[i][1]) / 1e3 for i in range(len(rs) - 1) if rs[i + 1][0] > rs[i][1]]
        print(f"  queue {q}: {len(rs)} kernels, {b:.1f} ms busy, gaps {sum(qg) / 1e3:.1f} ms "
              f"(median {sorted(qg)[len(qg) // 2] if qg else 0:.2f} us)")
        hist(qg, "      ")

    # concurrency sweep: share every instant among the kernels running in it
    from pmc_to_json import family
    ev = []
    for i, (s, e, q, n) in enumerate(rows):
        ev.append((s, 1, i))
        ev.append((e, 0, i))
    ev.sort()
    running, last = set(), ev[0][0]
    conc = collections.Counter()
    own = collections.Counter()
    for t, kind, i in ev:
        if t > last and running:
            dt = t - last
            conc[min(len(running), 4)] += dt
            share = dt / len(running)
            for j in running:
                own[family(rows[j][3])] += share
        last = t
        if kind:
            running.add(i)
        else:
            running.discard(i)
    print("concurrency (ms of the window with k kernels running): " +
          ", ".join(f"{k}{'+' if k == 4 else ''}: {v / 1e6:.1f}" for k, v in sorted(conc.items())))
    dur = collections.Counter()
    cnt = collections.Counter()
    for s, e, q, n in rows:
        dur[family(n)] += e - s
        cnt[family(n)] += 1
    div = steps or 1.0
    print(f"wall-time ownership by kernel family (ms{' per step' if steps else ''}; own = shared-instant attribution, "
          "dur = plain sum of durations):")
    for f, v in own.most_common(28):
        print(f"  own {v / 1e6 / div:8.2f}  dur {dur[f] / 1e6 / div:8.2f}  n {cnt[f] / div:7.0f}  {f}")


if __name__ == "__main__":
    main()
