#!/usr/bin/env python3
"""From a rocprofv3 rocpd database of the bench command: how the two half-streams of one ENTER+EXIT fill the timeline — time with
0 / 1 / >= 2 kernels in flight over the last `last` dispatches, per-stream busy time, and the longest idle gaps.
usage: overlap_analysis.py DB LAST"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); last = int(sys.argv[2])
cols = [c[1] for c in db.execute("pragma table_info(kernels)")]
qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
sel = f"name, start, end, {qcol}" if qcol else "name, start, end, 0"
rows = db.execute(f"select {sel} from kernels order by start desc limit ?", (last,)).fetchall()[::-1]
t0, t1 = rows[0][1], max(r[2] for r in rows)
ev = sorted([(r[1], 1) for r in rows] + [(r[2], -1) for r in rows])
depth, prev, hist = 0, t0, {}
for t, dlt in ev:
    hist[min(depth, 2)] = hist.get(min(depth, 2), 0) + (t - prev)
    depth += dlt; prev = t
span = t1 - t0
print(f"dispatches {len(rows)}  span {span / 1e6:.3f} ms   in flight: 0 kernels {hist.get(0, 0) / 1e6:.3f} ms, 1 kernel {hist.get(1, 0) / 1e6:.3f} ms, >= 2 kernels {hist.get(2, 0) / 1e6:.3f} ms")
by = {}
for n, s, e, q in rows:
    by.setdefault(q, []).append((s, e, n))
for q, v in by.items():
    busy = sum(e - s for s, e, _ in v)
    gaps = [v[i + 1][0] - v[i][1] for i in range(len(v) - 1)]
    print(f"  queue {q}: {len(v)} kernels, busy {busy / 1e6:.3f} ms, positive gaps {sum(g for g in gaps if g > 0) / 1e6:.3f} ms (median {sorted(gaps)[len(gaps) // 2] / 1e3:.2f} us)")
cls = {}
for n, s, e, q in rows:
    k = n.split("<")[0].split("::")[-1][:22]
    c = cls.setdefault(k, [0, 0.0]); c[0] += 1; c[1] += (e - s)
for k, (c, t) in sorted(cls.items(), key=lambda kv: -kv[1][1]):
    print(f"  {k:24s} {c:5d} x {t / c / 1e3:8.1f} us = {t / 1e6:7.3f} ms")
