#!/usr/bin/env python3
"""From a rocprofv3 rocpd database: idle gaps between consecutive kernels of the timed steps (single-stream schedule)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); last = int(sys.argv[2])
rows = db.execute("select name, start, end from kernels order by start desc limit ?", (last,)).fetchall()[::-1]
span = rows[-1][2] - rows[0][1]
busy = sum(r[2] - r[1] for r in rows)
gaps = [rows[i + 1][1] - rows[i][2] for i in range(len(rows) - 1)]
gaps_pos = [g for g in gaps if g > 0]
print(f"dispatches {len(rows)}  span {span/1e6:.3f} ms  sum of kernel durations {busy/1e6:.3f} ms  sum of positive gaps {sum(gaps_pos)/1e6:.3f} ms")
gs = sorted(gaps_pos)
if gs:
    print(f"gap median {gs[len(gs)//2]/1e3:.2f} us  p90 {gs[int(len(gs)*0.9)]/1e3:.2f} us  max {gs[-1]/1e3:.2f} us")
by = {}
for i in range(len(rows) - 1):
    k = rows[i][0].split('<')[0].split('::')[-1][:24]
    by.setdefault(k, []).append(gaps[i])
for k, v in by.items():
    print(f"  after {k:26s} n={len(v):4d} avg gap {sum(v)/len(v)/1e3:7.2f} us")
