"""Evidence hygiene: the committed rocprofv3 summaries of the bench command must describe the kernels the
bench times.  For every round directory profiles/rNN (NN >= 5) that holds a bench trace:
  * bench_trace/kernel_hot.json (dominant kernel over the timed launches of `bench.py`) and counters_secp256k1_20.json (the same
    kernel class under tools/prof_case.py) must agree on the average launch duration within 15 %;
  * every "avg X us" that profiles/README.md quotes next to that round's kernel_hot.json must be the number in the file."""
import glob
import json
import os
import re

import pytest

from conftest import ROOT

ROUNDS = sorted(d for d in glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]")) if int(os.path.basename(d)[1:]) >= 5
                and os.path.exists(os.path.join(d, "bench_trace", "kernel_hot.json")))


@pytest.mark.skipif(not ROUNDS, reason="no bench trace committed for round >= 5 yet")
@pytest.mark.parametrize("rdir", ROUNDS, ids=[os.path.basename(r) for r in ROUNDS])
def test_bench_trace_agrees_with_the_counter_passes(rdir):
    hot = json.load(open(os.path.join(rdir, "bench_trace", "kernel_hot.json")))
    ctr = json.load(open(os.path.join(rdir, "counters_secp256k1_20.json")))
    fam = hot["kernel"].split("<")[0]
    assert fam in ctr["classes"], (fam, list(ctr["classes"]))
    a, b = hot["avg_us"], ctr["classes"][fam]["avg_us"]
    assert hot["dispatches"] > 0 and abs(a - b) <= 0.15 * b, f"{rdir}: kernel_hot.json avg {a:.1f} us vs counters avg {b:.1f} us"
    # the trace's own line of the bench must be the headline workload, with the latency section off
    cmd = open(os.path.join(rdir, "bench_trace", "command.txt")).read()
    assert "bench.py" in cmd


@pytest.mark.skipif(not ROUNDS, reason="no bench trace committed for round >= 5 yet")
@pytest.mark.parametrize("rdir", ROUNDS, ids=[os.path.basename(r) for r in ROUNDS])
def test_readme_quotes_the_numbers_of_the_files(rdir):
    name = os.path.basename(rdir)
    readme = open(os.path.join(ROOT, "profiles", "README.md")).read()
    m = re.search(r"\* `%s/`.*?(?=\n\* `r[0-9][0-9]/`|\Z)" % name, readme, re.S)
    assert m, f"profiles/README.md has no section for {name}/"
    sec = m.group(0)
    hot = json.load(open(os.path.join(rdir, "bench_trace", "kernel_hot.json")))
    quoted = re.findall(r"kernel_hot\.json`.{0,200}?avg ([0-9.]+) us", sec, re.S)
    assert quoted, f"README section of {name}/ does not quote kernel_hot.json's average"
    for q in quoted:
        assert abs(float(q) - hot["avg_us"]) <= 0.02 * hot["avg_us"] + 0.05, (q, hot["avg_us"])
