#!/usr/bin/env python3
"""Round 6: where does a step's time go, per kernel CLASS, against each class's VALU issue floor?  Reads a counters.json of
tools/prof_counters.sh (kernel-trace + PMC passes; in the PMC passes rocprofv3 serialises dispatches, so `pmc_avg_us` is the SOLO
duration of a launch) and prints, per class: launches per step, solo time per launch and per step, the VALU busy time of a launch
(4 x SQ_ACTIVE_INST_VALU quad-cycles / 1024 SIMDs / profiled clock = the time the launch would take if every SIMD issued VALU
instructions back to back: its ISSUE FLOOR), floor / solo, waiting wave-cycles, HBM rate.  Last line: the sum of the solo times and of
the floors per step next to the measured step time (pass it as the third argument, in ms) — the gap between the sum of the floors and the step
is what overlap across streams, fill / drain and waiting leave on the table.
usage: class_breakdown.py COUNTERS.json [POLYS_PER_STEP] [MEASURED_MS_PER_STEP]"""
import json
import sys

d = json.load(open(sys.argv[1]))
polys = int(sys.argv[2]) if len(sys.argv) > 2 else 1
meas = float(sys.argv[3]) if len(sys.argv) > 3 else None
print(d["workload"])
print(f"{'class':<20} {'launches':>8} {'solo us':>9} {'solo ms/step':>12} {'floor us':>9} {'floor ms/step':>13} {'floor/solo':>10} {'wait/wave-cyc':>13} {'HBM TB/s':>9} {'LDS confl/act':>13}")
tot_solo = tot_floor = 0.0
for c, r in sorted(d["classes"].items(), key=lambda kv: -kv[1].get("pmc_avg_us", 0) * kv[1]["launches_per_step"]):
    if "pmc_avg_us" not in r or "SQ_ACTIVE_INST_VALU" not in r:
        continue
    n, solo = r["launches_per_step"], r["pmc_avg_us"]
    mhz = r.get("clock_mhz_profiled") or 2400.0
    floor = 4.0 * r["SQ_ACTIVE_INST_VALU"] / 1024.0 / mhz            # us
    wait = r.get("SQ_WAIT_ANY", 0.0) / max(r.get("SQ_WAVE_CYCLES", 1.0), 1.0)
    hbm = r.get("hbm_bytes_per_launch", 0.0) / (solo * 1e-6) / 1e12
    lds = r.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(r.get("SQ_ACTIVE_INST_LDS", 1.0), 1.0)
    tot_solo += n * solo / 1e3; tot_floor += n * floor / 1e3
    print(f"{c:<20} {n:>8} {solo:>9.1f} {n * solo / 1e3:>12.3f} {floor:>9.1f} {n * floor / 1e3:>13.3f} {floor / solo:>10.2f} {wait:>13.2f} {hbm:>9.2f} {lds:>13.2f}")
line = f"sum over classes: solo {tot_solo:.3f} ms, VALU issue floor {tot_floor:.3f} ms per step"
if polys > 1:
    line += f" = {tot_solo / polys:.3f} / {tot_floor / polys:.3f} ms per polynomial pair ({polys} per step)"
if meas:
    line += f"; measured step {meas:.3f} ms" + (f" = {meas / polys:.3f} per pair" if polys > 1 else "") + f": floor / measured = {tot_floor / meas:.2f}, solo sum / measured = {tot_solo / meas:.2f}"
print(line)
