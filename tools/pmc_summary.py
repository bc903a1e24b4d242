#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 PMC counters from a rocpd database (*_results.db).
  pmc_summary.py <db>                          markdown table: kernel, counter, dispatches, avg, total
  pmc_summary.py <db> --kernel SUBSTR --last N  JSON: averages over the LAST N dispatches of the kernels whose
                                                name contains SUBSTR (the timed steps come after tree construction)"""
import json
import sqlite3
import sys


def cols_of(db):
    cur = db.execute("select * from counters_collection limit 1")
    cols = [d[0] for d in cur.description]
    k = "kernel_name" if "kernel_name" in cols else "name"
    c = "counter_name" if "counter_name" in cols else "counter"
    v = "value" if "value" in cols else "counter_value"
    d = "dispatch_id" if "dispatch_id" in cols else ("id" if "id" in cols else None)
    return cols, k, c, v, d


def main():
    db = sqlite3.connect(sys.argv[1])
    cols, kcol, ccol, vcol, dcol = cols_of(db)
    if "--schema" in sys.argv:
        print(cols)
        for r in db.execute("select * from counters_collection limit 3"):
            print(r)
        return
    if "--kernel" in sys.argv:
        sub = sys.argv[sys.argv.index("--kernel") + 1]
        last = int(sys.argv[sys.argv.index("--last") + 1])
        out = {}
        for (counter,) in db.execute(f"select distinct {ccol} from counters_collection").fetchall():
            rows = db.execute(f"select {dcol}, sum({vcol}) from counters_collection where {kcol} like ? and {ccol} = ? group by {dcol} order by {dcol} desc limit ?",
                              (f"%{sub}%", counter, last)).fetchall()
            if rows:
                out[counter] = {"dispatches": len(rows), "avg": sum(r[1] for r in rows) / len(rows)}
        print(json.dumps({"kernel": sub, "last": last, "counters": out}))
        return
    rows = db.execute(f"select {kcol}, {ccol}, count(*), avg({vcol}), sum({vcol}) from counters_collection group by {kcol}, {ccol} order by 5 desc").fetchall()
    print("| kernel | counter | samples | avg per sample | total |")
    print("|---|---|---:|---:|---:|")
    for k, c, n, avg, tot in rows:
        k = k if len(k) < 90 else k[:87] + "..."
        print(f"| `{k}` | {c} | {n} | {avg:.4g} | {tot:.6g} |")


if __name__ == "__main__":
    main()
