#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 PMC counters from a rocpd database (*_results.db).
Usage: pmc_summary.py <db> [--schema]   -> markdown table: kernel, dispatches, avg of each counter."""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    if "--schema" in sys.argv:
        for name, sql in db.execute("select name, sql from sqlite_master where type='view' and name in ('pmc_events','kernels','counters_collection')"):
            print(name, "::", sql, "\n")
        for v in ("pmc_events", "counters_collection"):
            try:
                cur = db.execute(f"select * from {v} limit 2")
                print(v, [d[0] for d in cur.description])
                for r in cur.fetchall():
                    print("   ", r)
            except sqlite3.Error as e:
                print(v, "ERR", e)
        return
    cur = db.execute("select * from counters_collection limit 1")
    cols = [d[0] for d in cur.description]
    kcol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
    ccol = "counter_name" if "counter_name" in cols else "counter"
    vcol = "value" if "value" in cols else "counter_value"
    rows = db.execute(f"select {kcol}, {ccol}, count(*), avg({vcol}), sum({vcol}) from counters_collection group by {kcol}, {ccol} order by 5 desc").fetchall()
    print("| kernel | counter | dispatches | avg per dispatch | total |")
    print("|---|---|---:|---:|---:|")
    for k, c, n, avg, tot in rows:
        k = k if len(k) < 90 else k[:87] + "..."
        print(f"| `{k}` | {c} | {n} | {avg:.4g} | {tot:.6g} |")


if __name__ == "__main__":
    main()
