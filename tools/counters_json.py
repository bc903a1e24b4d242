#!/usr/bin/env python3
"""Per kernel CLASS counters of the timed launches, from rocprofv3 rocpd databases of `tools/prof_case.py FIELD LOG_N both REPS`.
usage: counters_json.py OUT_DIR FIELD LOG_N REPS     (OUT_DIR holds launches.json, trace/t_results.db, pmc*/p_results.db)
The timed launches of a class are its LAST reps x launches_per_rep dispatches (tree construction comes first).  Writes
OUT_DIR/counters.json: per class, launches per step, average duration (kernel-trace pass, kernels may overlap across the two
streams of a transform) and average counters per launch (PMC passes; rocprofv3 serialises dispatches there, so `pmc_avg_us`
is the SOLO duration).  FETCH_SIZE / WRITE_SIZE are in KiB; HBM bytes = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 — the factor 2
is the gfx950 correction of MI355X_MICROARCH.md (FETCH_SIZE tallies 128-byte requests at 64 bytes)."""
import glob
import hashlib
import json
import os
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """sha256 over the kernel sources the library is built from: bench.py compares it with the running tree and flags a
    counters file taken from other kernels as stale"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "ecfft_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".inc")) and f not in ("host_curve.h", "transport.h", "wire_parse.h"):      # device code + launch logic (not the host-only headers)
            with open(os.path.join(d, f), "rb") as fh:
                h.update(f.encode()); h.update(fh.read())
    return h.hexdigest()[:16]

CLASSES = {"k_stages_lds": ["k_stages_lds<", "k_stages_row256<"],
           "k_stages_col": ["k_stages_col<", "k_stages_col_mid<", "k_stages_col_enter<", "k_stages_col256<", "k_stages_col_mid256<", "k_stages_col_enter256<"],
           "k_enter_low": ["k_enter_low<"], "k_exit_low": ["k_exit_low<"], "k_decompose_stage": ["k_decompose_stage<"],
           "k_recombine_stage": ["k_recombine_stage<"]}


def klass(name):
    for c, pats in CLASSES.items():
        if any(p in name for p in pats):
            return c
    return None


def main():
    out, field, log_n, reps = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
    launches = json.load(open(os.path.join(out, "launches.json")))
    res = {"workload": f"{field} n=2^{log_n} ENTER+EXIT (tools/prof_case.py {field} {log_n} both {reps})", "reps": reps,
           "kernel_source_hash": kernel_source_hash(), "classes": {}}
    db = sqlite3.connect(os.path.join(out, "trace", "t_results.db"))
    by = {}
    for name, st, en in db.execute("select name, start, end from kernels order by start"):
        c = klass(name)
        if c:
            by.setdefault(c, []).append((en - st) / 1e3)
    for c, n_per in launches.items():
        if c not in by:
            continue
        d = by[c][-reps * n_per:]
        res["classes"][c] = {"launches_per_step": n_per, "avg_us": sum(d) / len(d), "sum_us_per_step": sum(d) / reps}
    for p in sorted(glob.glob(os.path.join(out, "pmc*", "p_results.db"))):
        d = sqlite3.connect(p)
        cc = [x[0] for x in d.execute("select * from counters_collection limit 1").description]
        k = "kernel_name" if "kernel_name" in cc else "name"
        cn = "counter_name" if "counter_name" in cc else "counter"
        v = "value" if "value" in cc else "counter_value"
        did = "dispatch_id" if "dispatch_id" in cc else "id"
        per = {}                                   # class -> counter -> list per dispatch (dispatch order)
        for name, disp, counter, val in d.execute(f"select {k}, {did}, {cn}, sum({v}) from counters_collection group by {did}, {cn} order by {did}"):
            c = klass(name)
            if c:
                per.setdefault(c, {}).setdefault(counter, []).append(val)
        # solo durations of the same pass
        dur = {}
        try:
            for name, st, en in d.execute("select name, start, end from kernels order by start"):
                c = klass(name)
                if c:
                    dur.setdefault(c, []).append((en - st) / 1e3)
        except sqlite3.Error:
            pass
        for c, ctrs in per.items():
            if c not in res["classes"]:
                continue
            n = reps * res["classes"][c]["launches_per_step"]
            for counter, vals in ctrs.items():
                vv = vals[-n:]
                res["classes"][c][counter] = sum(vv) / len(vv)
            if c in dur and "pmc_avg_us" not in res["classes"][c]:
                dd = dur[c][-n:]
                res["classes"][c]["pmc_avg_us"] = sum(dd) / len(dd)
    for c, r in res["classes"].items():
        if "FETCH_SIZE" in r and "WRITE_SIZE" in r:
            r["hbm_bytes_per_launch"] = (2.0 * r["FETCH_SIZE"] + r["WRITE_SIZE"]) * 1024.0
        if "GRBM_GUI_ACTIVE" in r and r.get("pmc_avg_us"):
            r["clock_mhz_profiled"] = r["GRBM_GUI_ACTIVE"] / 8.0 / r["pmc_avg_us"]        # counter is summed over the 8 XCDs
    json.dump(res, open(os.path.join(out, "counters.json"), "w"), indent=1)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
