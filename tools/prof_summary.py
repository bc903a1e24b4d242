#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (``rocprofv3 --kernel-trace --stats`` output, *_results.db)
into the per-kernel table committed under profiles/: calls, total / average / min / max duration."""
import sqlite3
import sys


def hot(db_path, sub, last):
    """average duration of the LAST `last` dispatches of kernels whose name contains `sub` (the timed steps)"""
    import json
    db = sqlite3.connect(db_path)
    rows = db.execute("select end - start from kernels where name like ? order by start desc limit ?", (f"%{sub}%", last)).fetchall()
    d = [r[0] for r in rows]
    print(json.dumps({"kernel": sub, "dispatches": len(d), "avg_us": sum(d) / max(len(d), 1) / 1e3, "total_ms": sum(d) / 1e6}))


def main(db_path, title):
    db = sqlite3.connect(db_path)
    rows = db.execute(
        "select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        "from kernels group by name order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# {title}\n")
    print("| kernel | calls | total ms | avg us | min us | max us | % |")
    print("|---|---:|---:|---:|---:|---:|---:|")
    for name, calls, tot, avg, mn, mx in rows:
        short = name if len(name) < 110 else name[:107] + "..."
        print(f"| `{short}` | {calls} | {tot / 1e6:.3f} | {avg / 1e3:.2f} | {mn / 1e3:.2f} | {mx / 1e3:.2f} | {100.0 * tot / total:.1f} |")


if __name__ == "__main__":
    if "--kernel" in sys.argv:
        hot(sys.argv[1], sys.argv[sys.argv.index("--kernel") + 1], int(sys.argv[sys.argv.index("--last") + 1]))
    else:
        main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else sys.argv[1])
