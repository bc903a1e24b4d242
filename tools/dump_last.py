#!/usr/bin/env python3
import sqlite3, sys
db = sqlite3.connect(sys.argv[1]); last = int(sys.argv[2])
cols = [d[0] for d in db.execute("select * from kernels limit 1").description]
g = "grid_size_x" if "grid_size_x" in cols else ("grid_x" if "grid_x" in cols else None)
q = f"select name, {g if g else 0}, start, end from kernels order by start desc limit ?"
rows = db.execute(q, (last,)).fetchall()[::-1]
for name, grid, st, en in rows:
    k = name.split('(')[0].replace('void ecfft::', '')[:40]
    print(f"{k:42s} grid {grid:8d}  {(en-st)/1e3:8.2f} us")
