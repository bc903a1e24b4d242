"""CPU-only checks of bench.py's host logic: the closed forms of SURVEY 8(d), the evidence-based `roofline` builder on synthetic
profiler classes (with and without a committed counters file) and the cgroup-aware core count of the socket baseline."""
import json
import os
import sys
import types

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)
import bench  # noqa: E402


def test_closed_forms_match_survey():
    n = 1 << 20
    we, wx = bench.w_mul(n)
    assert we + wx == pytest.approx(2.506e9, rel=1e-3)
    be, bx = bench.b_alg(n, 32)
    assert (be + bx) / 2**30 == pytest.approx(80.0, rel=2e-3)            # 80.0 GiB
    xe, xx = bench.executed_mul(n)
    assert xe + xx == pytest.approx(1.27e9, rel=5e-3)
    we, _ = bench.w_mul(1 << 24)
    assert we == pytest.approx(1.89e10, rel=5e-3)                        # config 5


class _FakeField:
    elem_bytes = 32

    def shader_clock_mhz(self, device):
        return 2300.0

    def mul_ceiling(self, waves, device):
        return 1.9e11 if waves == 4 else 2.0e11


def _classes():
    return [{"name": "k_stages_lds", "launches": 950, "ms": 60.0, "alg_bytes": 4.0e11},
            {"name": "k_stages_col", "launches": 1290, "ms": 50.0, "alg_bytes": 2.2e11},
            {"name": "k_enter_low", "launches": 20, "ms": 10.0, "alg_bytes": 7.0e10},
            {"name": "k_exit_low", "launches": 20, "ms": 23.0, "alg_bytes": 1.5e11},
            {"name": "pointwise", "launches": 0, "ms": 0.0, "alg_bytes": 0.0}]


def test_roofline_uses_committed_counters_and_never_exceeds_one():
    args = types.SimpleNamespace(field="secp256k1", log_n=20, steps=10)
    r = bench.build_roofline(args, _FakeField(), 1 << 20, _classes(), 8.4e-3, 0)
    assert r["counters_source"] and os.path.exists(os.path.join(ROOT, r["counters_source"]))
    assert r["bound"] in ("valu", "hbm") and r["kernel"] == "k_stages_lds"
    fracs = [r["frac"], r["valu"]["frac"], r["valu_issue"]["frac"], r["whole_job"]["hbm_frac"]] + [c["hbm_frac"] for c in r["per_class"]]
    assert all(0.0 < f <= 1.0 for f in fracs), fracs
    assert r["traffic"] == pytest.approx(r["achieved"] * 1e9 * r["avg_launch_us"] * 1e-6, rel=1e-6)      # achieved = traffic / launch time
    assert "frac" not in r["effective"]                                   # the stage-streaming figure is not a fraction
    json.dumps(r)


def test_roofline_without_a_counters_file_degrades_gracefully():
    args = types.SimpleNamespace(field="m31", log_n=17, steps=10)        # no committed pass for this workload
    r = bench.build_roofline(args, _FakeField(), 1 << 17, _classes(), 1e-3, 0)
    assert r["counters_source"] is None and r["achieved"] is None and r["frac"] is None and r["traffic"] is None
    assert "whole_job" not in r and "valu_issue" not in r and 0 < r["valu"]["frac"]
    json.dumps(r)


def test_cpu_quota_parsing(tmp_path, monkeypatch):
    q = bench._cpu_quota()
    assert q is None or q > 0
    assert bench._socket_cores() >= 1
