#!/usr/bin/env python3
"""Per (kernel, grid size) averages from rocprofv3 rocpd databases: durations from a --kernel-trace db, counters from
--pmc dbs.  Tree construction launches have small / odd grids, the timed transforms have the large ones, so grouping by
grid separates them without knowing launch counts.
usage: pmc_by_grid.py --min-grid G trace.db [pmc1.db pmc2.db ...]"""
import sqlite3, sys
from collections import defaultdict

def short(name):
    return name.split('(')[0].replace('void ecfft::', '').replace('ecfft::', '')[:46]

def gridcol(cols):
    for c in ("grid_size_x", "grid_x", "grid_size"):
        if c in cols: return c
    return None

args = sys.argv[1:]
min_grid = 0
if args[0] == "--min-grid":
    min_grid = int(args[1]); args = args[2:]
trace, pmcs = args[0], args[1:]
db = sqlite3.connect(trace)
cols = [d[0] for d in db.execute("select * from kernels limit 1").description]
g = gridcol(cols)
dur = defaultdict(list)
for name, grid, st, en in db.execute(f"select name, {g}, start, end from kernels"):
    if grid >= min_grid: dur[(short(name), grid)].append((en - st) / 1e3)
ctr = defaultdict(lambda: defaultdict(list))
for p in pmcs:
    d = sqlite3.connect(p)
    cc = [x[0] for x in d.execute("select * from counters_collection limit 1").description]
    if "--schema" in sys.argv: print(cc)
    k = "kernel_name" if "kernel_name" in cc else "name"
    c = "counter_name" if "counter_name" in cc else "counter"
    v = "value" if "value" in cc else "counter_value"
    did = "dispatch_id" if "dispatch_id" in cc else "id"
    gg = gridcol(cc)
    if gg is None:
        print("no grid column in", p, cc); continue
    for name, grid, counter, val in d.execute(f"select {k}, {gg}, {c}, sum({v}) from counters_collection group by {did}, {c}"):
        if grid >= min_grid: ctr[(short(name), grid)][counter].append(val)
names = sorted({c for kk in ctr.values() for c in kk})
print("kernel | grid | n | avg_us | " + " | ".join(names))
for key in sorted(dur, key=lambda k: -sum(dur[k])):
    row = [key[0], str(key[1]), str(len(dur[key])), f"{sum(dur[key]) / len(dur[key]):.2f}"]
    for c in names:
        vals = ctr.get(key, {}).get(c)
        row.append(f"{sum(vals) / len(vals):.4g}" if vals else "-")
    print(" | ".join(row))
