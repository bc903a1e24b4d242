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


def test_ab_switches_empty_the_counter_derived_fields(monkeypatch):
    """VERDICT r03 evidence hygiene: the committed PMC counters describe the default code path; with an A/B switch in the
    environment the line must not carry counter-derived numbers of other kernels (mixed epochs with counters_stale = false)"""
    monkeypatch.setenv("ECFFT_NO_MFMA", "1")
    args = types.SimpleNamespace(field="secp256k1", log_n=20, steps=10)
    r = bench.build_roofline(args, _FakeField(), 1 << 20, _classes(), 8.4e-3, 0)
    assert r["counters_skipped_for_switches"] == ["ECFFT_NO_MFMA"] and r["counters_source"] is None and r["counters_stale"] is False
    assert r["achieved"] is None and r["frac"] is None and r["traffic"] is None
    assert r["valu"]["of_which_on_matrix_cores"] is None and r["valu"]["frac"] is None and "valu_issue" not in r
    assert all("hbm_bytes_per_launch" not in c and "valu_insts_per_launch" not in c for c in r["per_class"])
    assert r["frac_alg"] > 0                                              # the event-time figures stay


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
    # round 3: both fractions are emitted under their own names; the algorithmic one may exceed 1 and says so in the note
    assert r["frac_alg"] == pytest.approx(r["achieved_alg"] / bench.HBM_PEAK_GBS) and r["achieved_alg"] == pytest.approx(r["effective"]["GBs"])
    assert r["compulsory"]["bytes_per_step"] == 32 * 26 * (1 << 20) and r["whole_job"]["traffic_over_compulsory"] > 1.0
    assert isinstance(r["counters_stale"], bool) and r["running_kernel_hash"] == bench.kernel_source_hash()
    assert 0.0 < r["valu_machine"]["busy_frac"] <= 1.0 and 0.0 <= r["mfma"]["busy_frac"] < 1.0
    assert r["valu"]["of_which_on_matrix_cores"] + r["valu"]["valu_mul_per_step"] == pytest.approx(r["valu"]["executed_mul_per_step"])
    assert "peak_at_8_waves_per_simd" not in r["valu"]
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


@pytest.mark.gpu
@pytest.mark.parametrize("transport", ["callback", "rccl-stub"])
def test_gpus2_headline_is_the_split_transform(transport):
    """`--gpus N > 1` (the driver's SCALE command): the HEADLINE is the north_star partitioning — ONE ENTER+EXIT with the evaluation
    domain split over the ranks, "scaling": "strong" — the N independent polynomials sit under `replicas`, the split EXTEND
    (configs[3]) under `split`.  Two ranks share this box's GPU: through the gloo callback transport, and
    through the RCCL code path bound to the test-only stand-in library (real RCCL refuses two ranks on one device); on an N-GPU
    node the same command runs over RCCL / xGMI."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ECFFT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1")
    if transport == "rccl-stub":
        stub = os.path.join(ROOT, "tests", "stub_rccl", "librccl_stub.so")
        if not os.path.exists(stub):
            subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", stub, os.path.join(ROOT, "tests", "stub_rccl", "stub_rccl.cpp"), "-lrt"], check=True)
        env.update(ECFFT_BENCH_TRANSPORT="rccl", ECFFT_BENCH_RCCL_LIB=stub)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "12",
           "--split-log-n", "12", "--split-log-e", "12", "--cpu-log-n", "0", "--batch", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["headline"] == "split" and d["scaling"] == "strong" and d["roofline"] is not None
    sp = d["split"]
    assert sp["ranks"] == 2 and sp["status"] == 0 and ("stand-in" in sp["transport"]) == (transport == "rccl-stub")
    assert d["value"] == sp["enter_exit"]["value"] and d["ms_per_step"] == sp["enter_exit"]["ms_per_step"]
    assert "one transform" in d["config"]["workload"] and d["config"]["split_exit"] == "gather" and d["config"]["steps"] == 2
    rp = d["replicas"]
    assert rp["scaling"] == "weak" and rp["value"] > 0 and "independent" in rp["parallelism"]
    for key, exch in (("enter_exit", 1), ("extend", 4)):
        o = sp[key]
        assert o["round_trip_ok"] is True and o["scaling"] == "strong" and o["n_gpus"] == 2 and o["ranks_seen_by_transport"] == 2
        ph = o["phases"]
        assert ph["exchanges_per_step"] >= exch and ph["bytes_sent_per_step_per_rank"] > 0 and ph["comm_ms_per_step"] >= 0
        assert "sharded" in o["config"]["tables"]


@pytest.mark.gpu
def test_the_drivers_scale_command_with_8_ranks_dry_run():
    """the command the driver launches on an 8-GPU node — `python -m torch.distributed.run --nproc-per-node 8 bench.py
    --gpus 8 --steps K --warmup W`, nothing else on the line: n = 2^20 split ENTER+EXIT (9 exchanges per step), e = 2^22 split EXTEND
    (4), three kinds of contexts per rank — run END TO END with 8 ranks sharing this box's one GPU: gloo process group, the library's
    RCCL transport bound to the stand-in library.  Functional only (times mean nothing).  The first real 8-GPU run must not be the
    first 8-rank run.  Needs ~25 GB of HBM for the eight ranks' contexts: skipped on a busy / small device."""
    import socket
    import subprocess
    import torch
    free, _total = torch.cuda.mem_get_info(0)
    if free < 40 << 30:
        pytest.skip(f"only {free >> 30} GiB of HBM free")
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    stub = os.path.join(ROOT, "tests", "stub_rccl", "librccl_stub.so")
    if not os.path.exists(stub):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", stub, os.path.join(ROOT, "tests", "stub_rccl", "stub_rccl.cpp"), "-lrt"], check=True)
    env = dict(os.environ, ECFFT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", ECFFT_BENCH_TRANSPORT="rccl", ECFFT_BENCH_RCCL_LIB=stub)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["headline"] == "split" and d["scaling"] == "strong" and d["steps"] == 2
    assert "configs[2]" in d["config"]["workload"] and d["config"]["n"] == 1 << 20
    sp = d["split"]
    assert sp["ranks"] == 8 and sp["status"] == 0 and "stand-in" in sp["transport"]
    ee, ex = sp["enter_exit"], sp["extend"]
    assert ee["round_trip_ok"] is True and ee["ranks_seen_by_transport"] == 8 and ee["phases"]["exchanges_per_step"] == 9
    assert ex["round_trip_ok"] is True and ex["ranks_seen_by_transport"] == 8 and ex["phases"]["exchanges_per_step"] == 4
    assert d["value"] == ee["value"] and d["replicas"]["scaling"] == "weak" and d["replicas"]["value"] > 0


@pytest.mark.gpu
def test_a_stuck_split_part_costs_the_split_object_not_the_line():
    """The split part is the only code a one-GPU lease cannot run over multi-rank RCCL.  If it hangs (or raises) on a real
    node, the replica measurement that precedes it must still be printed: a watchdog emits the line with `split.error`."""
    import socket
    import subprocess
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, ECFFT_BENCH_BACKEND="gloo", MASTER_ADDR="127.0.0.1", ECFFT_SPLIT_TIMEOUT_S="0.01")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--log-n", "12",
           "--split-log-n", "16", "--split-log-e", "0", "--cpu-log-n", "0", "--batch", "0"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["roofline"] is not None
    assert d["scaling"] == "weak" and "replicas" not in d          # nothing promoted: the replica line stays the headline
    assert "did not finish" in d["split"]["error"] and "enter_exit" not in d["split"]


@pytest.mark.gpu
def test_single_gpu_line_is_unchanged_by_the_split_option():
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--log-n", "12", "--cpu-log-n", "0", "--batch", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert "split" not in d and d["n_gpus"] == 1 and d["roofline"]["frac_alg"] > 0
