#!/bin/bash
# usage: tools/trace_case.sh TAG FIELD LOG_N OP [REPS] -> per-dispatch listing of the LAST rep (kernel, grid, us) + gaps
set -u
TAG=$1; shift
OUT=gpurun_out/trace_$TAG
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace -d $OUT/trace -o t -- python tools/prof_case.py $* > $OUT/trace.log 2>&1
python - "$OUT/trace/t_results.db" <<'PY' > $OUT/dispatches.txt
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
cols = [d[0] for d in db.execute("select * from kernels limit 1").description]
g = "grid_size_x" if "grid_size_x" in cols else "grid_x"
w = "workgroup_size_x" if "workgroup_size_x" in cols else None
rows = db.execute(f"select name, {g}, start, end, stream_id from kernels order by start").fetchall() if "stream_id" in cols else [r + (0,) for r in db.execute(f"select name, {g}, start, end from kernels order by start").fetchall()]
# last rep = after the last gap > 200 us
cut = 0
for i in range(1, len(rows)):
    if rows[i][2] - rows[i - 1][3] > 200000: cut = i
sel = rows[cut:]
t0 = sel[0][2]
tot = 0
for name, grid, st, en, sid in sel:
    k = name.split('(')[0].replace('void ecfft::', '')[:44]
    print(f"{(st - t0) / 1e3:9.1f} us  +{(en - st) / 1e3:7.2f}  grid {grid:8d}  s{sid}  {k}")
    tot += en - st
print(f"launches {len(sel)}  span {(sel[-1][3] - t0) / 1e3:.1f} us  sum of kernel time {tot / 1e3:.1f} us")
PY
rm -rf $OUT/trace
tail -80 $OUT/dispatches.txt
