#!/usr/bin/env python3
"""bench.py — BASELINE.json metric on MI355X: secp256k1 Fp field-mul/s and ENTER+EXIT wall time at n=2^20.

One "step" = one ENTER followed by one EXIT of a degree-(n-1) polynomial with synthetic (seeded,
uniform) coefficients that are already resident in HBM when the timed region starts.  field-mul/s
uses the ALGORITHMIC multiplication count of the reference's recursion:
    W_mul(ENTER) = 2nL(L-1) + nL,   W_mul(EXIT) = 4nL(L-1) + 4.5nL,   L = log2 n
independent of how this implementation re-associates the arithmetic.

Multi-GPU (driver launches `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`):
the path shards over independent polynomials, one per GPU, no data-path collective (weak scaling);
RCCL is used only for the barrier and the max-over-ranks of the timed region.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured float4 copy)
P_SECP = 2**256 - 2**32 - 977


def w_mul(n):
    L = n.bit_length() - 1
    return 2 * n * L * (L - 1) + n * L, 4 * n * L * (L - 1) + 4.5 * n * L


def executed_mul(n):
    """multiplies the HIP path really executes (normalised butterflies, merged innermost stage):
    ENTER (L^2 + L/2 + 1) n, EXIT (2 L^2 + 2) n"""
    L = n.bit_length() - 1
    return (L * L + 0.5 * L + 1) * n, (2 * L * L + 2) * n


def b_alg(n, s):
    L = n.bit_length() - 1
    enter = s * (2 * n * L * (L - 1) + 8 * (n - 1 - L) + 3 * n * L + 2 * (n - 1))
    exit_ = s * (4 * n * L * (L - 1) + 32 * (n - 1 - L) + 8.5 * n * L + 8.5 * (n - 1))
    return enter, exit_


def synth(field, n, seed):
    """seeded uniform field elements in the crate's in-memory representation (host numpy)"""
    rng = np.random.default_rng(seed)
    if field == "m31":
        return rng.integers(0, 2**31 - 1, n, dtype=np.uint32)
    # uniform 256-bit words, rejection of values >= p is a 2^-223 event: clamp by clearing on collision
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    bad = (a[:, 3] == 0xFFFFFFFFFFFFFFFF) & (a[:, 2] == 0xFFFFFFFFFFFFFFFF) & (a[:, 1] == 0xFFFFFFFFFFFFFFFF) & (a[:, 0] >= 0xFFFFFFFEFFFFFC2F)
    a[bad, 0] = 0
    return a   # any canonical 256-bit pattern < p is a valid Montgomery-form element


def cpu_baseline(field, log_n):
    """the oracle (C restatement of the reference's recursive single-threaded algorithm, incl. its
    per-REDC batch inversion) timed on this host on a bounded sample: one ENTER + EXIT at 2^log_n."""
    from oracle import oracle
    F = oracle.field(field)
    n = 1 << log_n
    t = F.build_fftree(n)
    c = synth(field, n, 0xC0FFEE)
    t0 = time.perf_counter(); ev = t.enter(c); t1 = time.perf_counter(); back = t.exit(ev); t2 = time.perf_counter()
    assert np.array_equal(back, c)
    we, wx = w_mul(n)
    return {"value": (we + wx) / (t2 - t0), "unit": "field-mul/s", "cores": 1, "kind": "port",
            "sample": f"{field} n=2^{log_n} one ENTER ({t1 - t0:.3f}s) + one EXIT ({t2 - t1:.3f}s), oracle/ C restatement of the reference's "
                      f"recursive algorithm (per-call vectors from a per-thread size-class cache), 1 thread",
            "host_cpu": _cpu_name()}


def _socket_cores():
    """physical cores of one socket that this process may run on (falls back to the affinity mask size)"""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("cpu cores"):
                    return max(1, min(avail, int(line.split(":", 1)[1])))
    except (OSError, ValueError):
        pass
    return max(1, avail)


def _cpu_quota():
    """CPUs this container may actually use at once (cgroup v2 cpu.max / v1 cfs quota), or None when unlimited"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()[:2]
            return None if q == "max" else float(q) / float(per)
    except (OSError, ValueError):
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:
            q = float(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            per = float(f.read())
        return None if q <= 0 else q / per
    except (OSError, ValueError):
        return None


def cpu_baseline_socket(field, log_n):
    """single-socket figure: the same single-threaded oracle on every physical core of one socket at once,
    one independent polynomial per thread (the reference has no threading of its own; this is its best case on a socket)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import oracle
    F = oracle.field(field)
    n = 1 << log_n
    t = F.build_fftree(n)
    phys = _socket_cores()
    quota = _cpu_quota()
    cores = phys if quota is None else max(1, min(phys, int(quota)))      # more threads than the cgroup quota only time-slice
    inputs = [synth(field, n, 0xC0FFEE + 1 + i) for i in range(cores)]

    def one(c):                      # ctypes releases the GIL inside the C calls
        return np.array_equal(t.exit(t.enter(c)), c)
    with ThreadPoolExecutor(max_workers=cores) as ex:
        t0 = time.perf_counter(); ok = list(ex.map(one, inputs)); dt = time.perf_counter() - t0
    assert all(ok)
    we, wx = w_mul(n)
    val = cores * (we + wx) / dt
    return {"value": val, "unit": "field-mul/s", "cores": cores, "kind": "port",
            "socket_physical_cores": phys, "cgroup_cpu_quota": quota,
            "value_extrapolated_to_socket": val * phys / cores,
            "sample": f"{field} n=2^{log_n}: {cores} independent ENTER+EXIT round trips, one per thread, {dt:.3f}s wall; the socket has {phys} physical "
                      f"cores" + (f" but this container's cgroup allows {quota:g} CPUs, so {cores} threads ran and value_extrapolated_to_socket scales "
                                  f"the measured figure linearly to {phys} (the transforms are independent and scale linearly up to the quota)" if quota and quota < phys else "")
                      + "; allocations served by per-thread caches (no mmap / page-fault contention)", "host_cpu": _cpu_name()}


def _cpu_name():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--log-n", type=int, default=20)
    ap.add_argument("--field", default="secp256k1", choices=["secp256k1", "m31"])
    ap.add_argument("--cpu-log-n", type=int, default=15, help="size of the bounded CPU-baseline sample (0 = skip)")
    ap.add_argument("--no-profile", action="store_true", help="skip the per-launch HIP-event pass")
    ap.add_argument("--mode", default="enter-exit", choices=["enter-exit", "extend-split", "enter-exit-split"],
                    help="extend-split: BASELINE configs[3] — ONE EXTEND of 2^log-n evaluations split over the ranks with RCCL all-to-all; "
                         "enter-exit-split: ONE ENTER+EXIT of 2^log-n coefficients split over the ranks (strong scaling)")
    ap.add_argument("--batch", type=int, default=8, help="also report throughput with this many polynomials per launch (0 = skip)")
    ap.add_argument("--split-log-n", type=int, default=20, help="--gpus N > 1: size of the ONE ENTER+EXIT split over the ranks reported under `split` (0 = skip)")
    ap.add_argument("--split-log-e", type=int, default=22, help="--gpus N > 1: size of the ONE EXTEND split over the ranks reported under `split` (0 = skip)")
    ap.add_argument("--stripe-min-gain", type=int, default=None, help="--gpus N > 1: ecfft_comm_set_link_striping threshold in bytes for the split part (default: the library's — striping OFF since round 6; 4194304 = the projection's threshold; 0 = always when it helps; -1 = never)")
    ap.add_argument("--split-exit", default="auto", choices=["auto", "gather", "shard"],
                    help="--gpus N > 1: form of the split EXIT — gather: full (replicated) context, ONE all-gather, top levels redundant; shard: "
                         "EXIT-shard context, split top levels; auto: gather up to n = 2^21, shard above")
    args = ap.parse_args()

    import torch
    import ecfft_amd
    global _STRIPE_MIN_GAIN
    _STRIPE_MIN_GAIN = args.stripe_min_gain

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with {args.gpus} processes (WORLD_SIZE={world})")
    ndev = torch.cuda.device_count()
    backend = os.environ.get("ECFFT_BENCH_BACKEND", "nccl")      # "gloo" lets several ranks share one GPU (functional test only)
    if backend == "nccl" and world > ndev:
        raise SystemExit(f"{world} ranks but {ndev} GPU(s): one process per GPU")
    local_rank = local_rank % max(ndev, 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"

    if args.mode in ("extend-split", "enter-exit-split"):
        fn = extend_split if args.mode == "extend-split" else enter_exit_split
        line = fn(args, torch, dist, ecfft_amd, rank, local_rank, world, red_dev, args.log_n)
        if rank == 0:
            print(json.dumps(line))
        if dist is not None:
            dist.barrier(); dist.destroy_process_group()
        return

    n = 1 << args.log_n
    F = ecfft_amd.FIELDS[args.field]
    # host-side input first, so that nothing but the upload separates the device's first work (context, tree build) from the warm-up steps:
    # an idle gap of tens of ms lets the chip drop its clocks, and the first 3 - 4 steps after one run 5 - 30 % slow (tools/ramp_check.py)
    host = synth(args.field, n, 0x5EED0000 + 2 + rank)
    view = host.view(np.int64) if args.field == "secp256k1" else host.view(np.int32)
    # the process's first device work (HIP context, queues, module load: 0.1 - 0.2 s on this stack) is not tree construction:
    # it is timed on its own so that tree_build_s is what ecfft_build_fftree costs a process that already uses its GPU
    t_init0 = time.perf_counter()
    torch.zeros(1, device=f"cuda:{local_rank}"); ecfft_amd.device_info(local_rank); torch.cuda.synchronize()
    hip_init_s = time.perf_counter() - t_init0
    t_build0 = time.perf_counter()
    tree = F.build_fftree(n, device=local_rank)
    if tree is None:
        raise SystemExit("n exceeds the curve's 2-adicity")
    torch.cuda.synchronize()
    build_s = time.perf_counter() - t_build0

    coeffs = torch.from_numpy(view).cuda()          # resident in HBM before the timed region

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def step():
        ev = tree.enter(coeffs)
        return ev, tree.exit(ev)

    for _ in range(args.warmup):
        ev, back = step()
    barrier()
    if args.warmup:
        assert torch.equal(back, coeffs), "EXIT(ENTER(c)) != c"

    # ---- timed region: exactly K steps --------------------------------------------------------
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ev, back = step()
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.equal(back, coeffs), "EXIT(ENTER(c)) != c"

    # ---- per-kernel HIP-event pass over the same K steps (events on the launch stream) -----------
    roofline = None
    split = {}
    if not args.no_profile and args.steps > 0:
        tree.profile(True)
        te0 = time.perf_counter()
        for _ in range(args.steps):
            ev = tree.enter(coeffs)
        torch.cuda.synchronize(); te1 = time.perf_counter()
        for _ in range(args.steps):
            back = tree.exit(ev)
        torch.cuda.synchronize(); te2 = time.perf_counter()
        classes = tree.profile_read()
        tree.profile(False)
        split = {"enter_ms": (te1 - te0) * 1e3 / args.steps, "exit_ms": (te2 - te1) * 1e3 / args.steps,
                 "instrumented_ms_per_step": (te2 - te0) * 1e3 / args.steps}
        roofline = build_roofline(args, F, n, classes, elapsed / args.steps, local_rank)

    # ---- extra: batched throughput (B independent polynomials share every launch; not the headline value) -----
    batched = None
    if args.batch > 1 and args.steps > 0:
        B = args.batch
        big = torch.from_numpy(np.concatenate([view] * B)).cuda() if args.field == "m31" else torch.from_numpy(np.concatenate([view] * B, axis=0)).cuda()
        evb = tree.enter(big, count=B); backb = tree.exit(evb, count=B)        # warm-up (also grows the scratch)
        barrier()
        tb0 = time.perf_counter()
        for _ in range(args.steps):
            evb = tree.enter(big, count=B); backb = tree.exit(evb, count=B)
        barrier()
        tb = time.perf_counter() - tb0
        assert torch.equal(backb, big), "batched EXIT(ENTER(c)) != c"
        we_, wx_ = w_mul(n)
        batched = {"batch": B, "ms_per_step": tb * 1e3 / args.steps, "ms_per_transform_pair": tb * 1e3 / args.steps / B,
                   "field_mul_per_s_per_gpu": (we_ + wx_) * B * args.steps / tb}
        del big, evb, backb

    # ---- extra: the latency regime (BASELINE.json configs[1], n = 2^16, and 2^17 = the rank-local chunk of an 8-GPU split at 2^20):
    # one transform, device buffers, median of 15 — reported beside the headline, never part of `value` ----
    latency = None
    if world == 1 and args.field == "secp256k1" and args.log_n == 20 and args.steps > 0 and not os.environ.get("ECFFT_BENCH_NO_LATENCY"):
        latency = {}
        for ln in (16, 17):
            ts = F.build_fftree(1 << ln, device=local_rank)
            hs = synth(args.field, 1 << ln, 0x5EED0000 + ln)
            xs = torch.from_numpy(hs.view(np.int64)).cuda()
            for _ in range(3):
                ys = ts.exit(ts.enter(xs))
            assert torch.equal(ys, xs), "EXIT(ENTER(c)) != c in the latency regime"
            tl = []
            for _ in range(15):
                torch.cuda.synchronize(); t0 = time.perf_counter()
                ys = ts.exit(ts.enter(xs))
                torch.cuda.synchronize(); tl.append(time.perf_counter() - t0)
            wl = sum(w_mul(1 << ln))
            med = sorted(tl)[len(tl) // 2]
            latency[f"2^{ln}"] = {"enter_exit_ms": med * 1e3, "field_mul_per_s": wl / med}
            del ts, xs, ys
        latency["note"] = ("ONE ENTER+EXIT per size (median of 15, device buffers, round trip asserted): launches with fewer tiles than CUs run the "
                           "256-element register / 16x16x64 matrix-core kernels (DESIGN.md 5.1); 2^16 is BASELINE.json configs[1]")

    out = None
    if rank == 0:
        we, wx = w_mul(n)
        value = (we + wx) * args.steps * world / elapsed
        out = {
            "metric": f"{args.field} Fp field-mul/s, ENTER+EXIT at n=2^{args.log_n}",
            "value": value, "unit": "field-mul/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u256" if args.field == "secp256k1" else "u32", "data": "synthetic",
            "value_note": "reference-equivalent throughput: the ALGORITHMIC multiply count of the reference's recursion (SURVEY 8(d)) per second; "
                          "the kernels execute about half as many multiplies (normalised butterflies) — see executed_field_mul_per_s",
            "executed_field_mul_per_s": sum(executed_mul(n)) * args.steps * world / elapsed,
            "config": {"workload": f"{args.field}::Fp n=2^{args.log_n} ENTER+EXIT (BASELINE.json configs[2])" if args.log_n == 20 and args.field == "secp256k1"
                       else f"{args.field}::Fp n=2^{args.log_n} ENTER+EXIT",
                       "n": n, "field": args.field, "parallelism": f"{world} independent polynomial(s), one per GPU, no collective",
                       "inputs": "device-resident (HBM) before the timed region", "tree_build_s": build_s, "hip_runtime_init_s": hip_init_s,
                       "steps": args.steps, "warmup": args.warmup},
            "roofline": roofline,
            "cpu_baseline": None,
            "batched": batched,
            "latency_regime": latency,
        }
        out.update(split)

    # ---- --gpus N > 1: besides the replica line, the north_star partitioning itself — ONE transform with its evaluation domain
    # split over the ranks (grouped ncclSend/ncclRecv at the top log2 N levels, sharded tables), same process group.  The replica
    # measurement above is complete at this point: a watchdog makes sure a fault in the split part (this is the only code of the
    # repository that a one-GPU lease cannot run over real multi-rank RCCL) costs the `split` object and not the whole line. -----
    if world > 1 and (args.split_log_n or args.split_log_e):
        import threading
        del tree, coeffs, ev, back
        torch.cuda.empty_cache()
        _tr = os.environ.get("ECFFT_BENCH_TRANSPORT", "rccl" if backend == "nccl" else "callback")
        split_obj = {"ranks": world, "transport": ("rccl" + (" (stand-in library, functional test)" if os.environ.get("ECFFT_BENCH_RCCL_LIB") else "")) if _tr == "rccl"
                     else f"callback over {backend} (functional test)"}
        limit = float(os.environ.get("ECFFT_SPLIT_TIMEOUT_S", "240"))
        finished = threading.Event()

        store = None
        try:
            store = dist.distributed_c10d._get_default_store()
        except Exception:  # pragma: no cover - private API moved: the time limit below still holds
            store = None

        def peer_failed():
            try:
                return store is not None and store.check(["ecfft_split_failed"])
            except Exception:  # pragma: no cover
                return False

        def watchdog():
            # a rank whose split part raised says so in the process group's store: its peers, blocked in an exchange that
            # will never complete, abort at once instead of waiting for the time limit
            t_end = time.monotonic() + limit
            while not finished.wait(min(1.0, max(limit, 0.01))):
                if peer_failed() or time.monotonic() >= t_end:
                    break
            else:
                return
            if finished.is_set():
                return
            # a peer is gone or stuck: ncclCommAbort on this rank's communicator(s) (ecfft_comm_abort) makes the blocked sharded call
            # return an error, so the process leaves through the ordinary path below with split.status = 1 — every rank has the same
            # watchdog, so nobody is left for torchrun to reap.  Only when that fails too (callback transport) the process is ended.
            split_obj["error"] = ("a peer's split part failed; " if peer_failed() else "") + f"the split part did not finish within {limit:.0f} s (ECFFT_SPLIT_TIMEOUT_S); replica line kept"
            split_obj["status"] = 1
            aborted = [c.abort() for c in list(_LIVE_COMMS)]
            split_obj["communicators_aborted"] = sum(1 for a in aborted if a)
            if any(aborted) and finished.wait(45):
                return
            if rank == 0:
                out["split"] = split_obj
                print(json.dumps(out), flush=True)
            os._exit(0)

        threading.Thread(target=watchdog, daemon=True).start()
        try:
            if args.split_log_n:
                split_obj["enter_exit"] = enter_exit_split(args, torch, dist, ecfft_amd, rank, local_rank, world, red_dev, args.split_log_n)
            if args.split_log_e:
                split_obj["extend"] = extend_split(args, torch, dist, ecfft_amd, rank, local_rank, world, red_dev, args.split_log_e)
        except Exception as ex:  # a rank that fails here leaves its peers in an exchange: their watchdogs abort those
            split_obj.setdefault("error", f"{type(ex).__name__}: {ex}")
            split_obj["status"] = 1
            sys.stderr.write(f"bench.py rank {rank}: split part failed: {split_obj['error']}\n")
            try:
                if store is not None:
                    store.set("ecfft_split_failed", str(rank))     # the peers' watchdogs see it within a second
            except Exception:  # pragma: no cover
                pass
            for c in list(_LIVE_COMMS):
                c.abort()                        # nothing of this rank may sit in an exchange when the process group is torn down
        split_obj.setdefault("status", 0)
        finished.set()
        if rank == 0:
            out["split"] = split_obj              # N = 1 lines carry no such key (byte-compatible with earlier rounds)
            ee = split_obj.get("enter_exit")
            if split_obj["status"] == 0 and ee and ee.get("round_trip_ok") and args.split_log_n == args.log_n:
                # the N > 1 headline is the north_star partitioning — ONE transform with its domain split over the
                # ranks (strong scaling: total work fixed); the N independent polynomials (trivial weak scaling) move
                # under `replicas`.  Only a split part that ran and round-tripped is promoted: a node where it fails keeps the replica line.
                out["replicas"] = {k: out[k] for k in ("metric", "value", "unit", "ms_per_step", "scaling", "executed_field_mul_per_s") if k in out}
                out["replicas"]["parallelism"] = out["config"]["parallelism"]
                out["metric"] = ee["metric"]; out["value"] = ee["value"]; out["ms_per_step"] = ee["ms_per_step"]; out["scaling"] = "strong"
                out["executed_field_mul_per_s"] = sum(executed_mul(1 << args.split_log_n)) * args.steps / (ee["ms_per_step"] * 1e-3 * args.steps)
                out["config"] = dict(ee["config"], workload=ee["config"]["workload"] + (" (BASELINE.json configs[2], domain split over the GPUs)" if args.split_log_n == 20 and args.field == "secp256k1" else ""),
                                     field=args.field, inputs="device-resident (HBM), block-distributed over the ranks before the timed region",
                                     tree_build_s=build_s, hip_runtime_init_s=hip_init_s)
                out["phases"] = ee["phases"]
                out["headline"] = "split"
                out["roofline_note"] = "roofline: per-launch evidence of the replica pass (one whole transform per GPU), kept for reference"
            else:
                out["headline"] = "replicas"

    if rank == 0:
        if world == 1 and args.cpu_log_n > 0:
            out["cpu_baseline"] = cpu_baseline(args.field, args.cpu_log_n)
            out["gpu_over_cpu"] = value / out["cpu_baseline"]["value"]
            out["cpu_baseline_socket"] = cpu_baseline_socket(args.field, args.cpu_log_n)
            out["gpu_over_cpu_socket"] = value / out["cpu_baseline_socket"]["value_extrapolated_to_socket"]
            out["gpu_over_cpu_threads_measured"] = value / out["cpu_baseline_socket"]["value"]     # against what really ran (cores = the cgroup quota)
            out["gpu_over_cpu_socket_note"] = ("gpu_over_cpu_socket: GPU value / CPU figure extrapolated to every physical core of the socket; "
                                               "gpu_over_cpu_threads_measured: GPU value / the measured multi-thread figure (cpu_baseline_socket.cores threads)")
        print(json.dumps(out))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


_LIVE_COMMS = []          # communicators of the split part: the watchdog aborts them when the part is stuck
_STRIPE_MIN_GAIN = None   # --stripe-min-gain


def _make_comm(D, dist, world):
    """RCCL communicator (one rank per GPU); ECFFT_BENCH_BACKEND=gloo -> host-staged callback transport (functional test only)"""
    # ECFFT_BENCH_TRANSPORT=rccl with ECFFT_BENCH_BACKEND=gloo and ECFFT_BENCH_RCCL_LIB=<tests/stub_rccl/librccl_stub.so>: the RCCL code
    # path with several ranks on ONE GPU (dry run of the driver's SCALE command, tests/test_bench_host.py) — variables of this harness;
    # the library itself reads none, the stand-in is handed over with ecfft_comm_set_rccl_library
    backend = os.environ.get("ECFFT_BENCH_BACKEND", "nccl")
    transport = os.environ.get("ECFFT_BENCH_TRANSPORT", "rccl" if backend == "nccl" else "callback")
    if transport == "rccl" and os.environ.get("ECFFT_BENCH_RCCL_LIB") and not _LIVE_COMMS:
        try:
            D.Comm.set_rccl_library(os.environ["ECFFT_BENCH_RCCL_LIB"])
        except Exception:      # already bound by an earlier communicator of this process
            pass
    c = D.Comm.rccl() if transport == "rccl" else D.Comm.callback()
    if _STRIPE_MIN_GAIN is not None:
        c.set_link_striping((1 << 64) - 1 if _STRIPE_MIN_GAIN < 0 else _STRIPE_MIN_GAIN)
    _LIVE_COMMS.append(c)
    return c


def _split_report(comm, steps):
    st = comm.stats()
    rep = {"comm_ms_per_step": st["comm_ms"] / max(steps, 1), "exchanges_per_step": st["exchanges"] / max(steps, 1),
           "bytes_sent_per_step_per_rank": st["bytes_sent"] / max(steps, 1)}
    if comm.world == 1:
        # one rank: every "exchange" is a device-to-device copy of the rank's own chunk through the transport — not communication
        rep["self_copy_ms_per_step"] = rep["comm_ms_per_step"]
        rep["comm_ms_per_step"] = 0.0
        rep["note"] = "world = 1: the exchanges are self send / receives; their time is self_copy_ms_per_step, no inter-GPU communication is measured"
    return rep


def extend_split(args, torch, dist, ecfft_amd, rank, local_rank, world, red_dev="cuda", log_e=None):
    """BASELINE configs[3]: one EXTEND of e = 2^log_n evaluations (tree T_2e) with the evaluation domain split over the
    ranks: block-distributed input, four grouped ncclSend/ncclRecv exchanges (block<->cyclic) around the top log2(P) stages,
    all below the C ABI (ecfft_extend_sharded).  Strong scaling: total work is fixed.  Checked by S0->S1->S0 round trip."""
    from ecfft_amd import distributed as D
    log_e = args.log_n if log_e is None else log_e
    e = 1 << log_e
    F = ecfft_amd.FIELDS[args.field]
    # sharded EXTEND-only context: this rank's 1/world share of the tables of T_2e, nothing else (ecfft_build_extend_shard)
    t_b = time.perf_counter()
    tree = F.build_extend_shard(e, world, rank, device=local_rank)
    if tree is None:
        raise SystemExit("2e exceeds the curve's 2-adicity")
    torch.cuda.synchronize(); build_s = time.perf_counter() - t_b
    c = e // world
    host = synth(args.field, e, 0x5EED0004)[rank * c:(rank + 1) * c]          # this rank's block of the same global vector
    x = torch.from_numpy(host.view(np.int64) if args.field == "secp256k1" else host.view(np.int32)).cuda()
    comm = _make_comm(D, dist, world)
    comm_world = comm.world

    def run(v, moiety):
        return tree.extend_sharded(comm, v, e, moiety)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        y = run(x, ecfft_amd.Moiety.S1)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        y = run(x, ecfft_amd.Moiety.S1)
    barrier()
    elapsed = time.perf_counter() - t0
    back = run(y, ecfft_amd.Moiety.S0)
    ok = bool(torch.equal(back, x))
    # second pass with per-exchange events: communication vs compute per step
    comm.stats(True)
    barrier(); t1 = time.perf_counter()
    for _ in range(args.steps):
        y = run(x, ecfft_amd.Moiety.S1)
    barrier(); inst = time.perf_counter() - t1
    phases = _split_report(comm, args.steps)
    phases["instrumented_ms_per_step"] = inst * 1e3 / max(args.steps, 1)
    phases["compute_ms_per_step"] = phases["instrumented_ms_per_step"] - phases["comm_ms_per_step"] - phases.get("self_copy_ms_per_step", 0.0)
    comm.stats(False)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); elapsed = float(tt.item())
        fl = torch.tensor([1 if ok else 0], device=red_dev); dist.all_reduce(fl, op=dist.ReduceOp.MIN); ok = bool(fl.item())
    L = log_e
    return {"metric": f"{args.field} Fp field-mul/s, one EXTEND of 2^{L} evaluations split over {world} GPU(s)",
            "value": 4 * e * L * args.steps / elapsed, "unit": "field-mul/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u256" if args.field == "secp256k1" else "u32", "data": "synthetic",
            "config": {"workload": f"{args.field}::Fp EXTEND e=2^{L} on T_2^{L + 1}" + (" (BASELINE.json configs[3])" if L == 22 and args.field == "secp256k1" else ""), "e": e,
                       "parallelism": f"evaluation domain block-split over {world} GPU(s): ecfft_extend_sharded, 4 grouped ncclSend/ncclRecv exchanges per EXTEND",
                       "tables": "sharded: each GPU holds its 1/world share of T_2e's EXTEND tables (ecfft_build_extend_shard)",
                       "table_bytes_per_gpu": tree.device_bytes, "context_build_s": build_s},
            "phases": phases, "round_trip_ok": ok, "ranks_seen_by_transport": comm_world}


def enter_exit_split(args, torch, dist, ecfft_amd, rank, local_rank, world, red_dev="cuda", log_n=None):
    """ONE ENTER followed by ONE EXIT of n = 2^log_n coefficients with the evaluation domain block-split over the ranks
    (tests/split_model.py is the Python model of it: local low levels, split EXTENDs + table_fma + one all-to-all per top level).  Strong
    scaling.  Checked by EXIT(ENTER(c)) == c on every rank."""
    from ecfft_amd import distributed as D
    log_n = args.log_n if log_n is None else log_n
    n = 1 << log_n
    F = ecfft_amd.FIELDS[args.field]
    c = n // world
    host = synth(args.field, n, 0x5EED0005)[rank * c:(rank + 1) * c]
    x = torch.from_numpy((host.view(np.int64) if args.field == "secp256k1" else host.view(np.int32)).reshape(c, -1).copy()).cuda()
    comm = _make_comm(D, dist, world)
    comm_world = comm.world
    t_b = time.perf_counter()
    if world > 1:
        # sharded contexts: the chain up to n/world + this rank's share of the top trees (the EXIT one is a collective build)
        t_enter = F.build_enter_shard(n, world, rank, device=local_rank)
        exit_form = args.split_exit if args.split_exit != "auto" else ("gather" if log_n <= 21 else "shard")
        if exit_form == "gather":
            # round 4: at these sizes the split top levels of an EXIT are latency bound (tools/split_project.py), so the EXIT runs on a
            # FULL context (1.8 GiB of tables at 2^20, replicated): one all-gather, then every top level redundantly — 1 exchange
            t_exit = F.build_fftree(n, device=local_rank)
            tables = ("ENTER sharded: chain up to n/world + the rank's share of the log2(world) top trees (ecfft_build_enter_shard); "
                      "EXIT full chain replicated: one all-gather, top levels redundant on the block of the rank's chunk (n <= 2^21)")
        else:
            t_exit = F.build_exit_shard(n, comm, device=local_rank)
            tables = "sharded: chain up to n/world + the rank's share of the log2(world) top trees (ecfft_build_enter_shard / ecfft_build_exit_shard)"
    else:
        t_enter = t_exit = F.build_fftree(n, device=local_rank)
        tables = "full chain (world = 1)"
    if t_enter is None or t_exit is None:
        raise SystemExit("n exceeds the curve's 2-adicity")
    torch.cuda.synchronize(); build_s = time.perf_counter() - t_b
    table_bytes = t_enter.device_bytes + (t_exit.device_bytes if t_exit is not t_enter else 0)

    def step():
        ev = t_enter.enter_sharded(comm, x, n)
        return t_exit.exit_sharded(comm, ev, n)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(max(args.warmup, 1)):
        back = step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        back = step()
    barrier()
    elapsed = time.perf_counter() - t0
    ok = bool(torch.equal(back.reshape(x.shape), x))
    comm.stats(True)
    barrier(); t1 = time.perf_counter()
    for _ in range(args.steps):
        back = step()
    barrier(); inst = time.perf_counter() - t1
    phases = _split_report(comm, args.steps)
    phases["instrumented_ms_per_step"] = inst * 1e3 / max(args.steps, 1)
    phases["compute_ms_per_step"] = phases["instrumented_ms_per_step"] - phases["comm_ms_per_step"] - phases.get("self_copy_ms_per_step", 0.0)
    comm.stats(False)
    if dist is not None:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=red_dev); dist.all_reduce(tt, op=dist.ReduceOp.MAX); elapsed = float(tt.item())
        fl = torch.tensor([1 if ok else 0], device=red_dev); dist.all_reduce(fl, op=dist.ReduceOp.MIN); ok = bool(fl.item())
    we, wx = w_mul(n)
    return {"metric": f"{args.field} Fp field-mul/s, one ENTER+EXIT at n=2^{log_n} split over {world} GPU(s)",
            "value": (we + wx) * args.steps / elapsed, "unit": "field-mul/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": elapsed * 1e3 / args.steps, "higher_is_better": True, "scaling": "strong",
            "vs_baseline": None, "dtype": "u256" if args.field == "secp256k1" else "u32", "data": "synthetic",
            "config": {"workload": f"{args.field}::Fp n=2^{log_n} ENTER+EXIT, one transform", "n": n,
                       "parallelism": f"coefficient/evaluation vector block-split over {world} GPU(s): ecfft_enter_sharded / ecfft_exit_sharded; levels above n/P use split EXTENDs and one re-blocking exchange per level",
                       "tables": tables, "table_bytes_per_gpu": table_bytes, "context_build_s": build_s,
                       "split_exit": (args.split_exit if args.split_exit != "auto" else ("gather" if log_n <= 21 else "shard")) if world > 1 else "none (world = 1)",
                       "steps": args.steps, "warmup": args.warmup},
            "phases": phases, "round_trip_ok": ok, "ranks_seen_by_transport": comm_world}


def kernel_source_hash():
    """the hash tools/counters_json.py stamps into a counters file (same code, kept in one place there)"""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    try:
        import counters_json
        return counters_json.kernel_source_hash()
    except Exception:  # pragma: no cover
        return None
    finally:
        sys.path.pop(0)


def compulsory_bytes(n, s):
    """what ENTER+EXIT cannot avoid moving: each transform's input and output once (4 n s) and every table element
    the pair needs once (ENTER u EXIT = 11 m per tree, 22 n for the chain)"""
    return s * (4 * n + 22 * n)


INSTR_PER_MUL = {"secp256k1": 169, "m31": 6}     # VALU instructions of the kernels' table multiply-add (tools/gen_mulmod_asm.py; field_m31.h)
N_SIMD = 256 * 4


# environment switches that change which kernels / code paths run (A/B tools): the committed PMC counters do not apply then
AB_SWITCHES = ("ECFFT_NO_MFMA", "ECFFT_NO_LOW16", "ECFFT_LOW32", "ECFFT_NO_FULL_CYCLIC", "ECFFT_NO_SMALL_TILES", "ECFFT_NO_ROW256", "ECFFT_NO_COL256",
               "ECFFT_SMALL_MIN_LOGC", "ECFFT_SMALL_TILES_MAX", "ECFFT_SMALL_LOW_MAX", "ECFFT_LIB")


def _counters(field, log_n):
    """per-class rocprofv3 counters of this workload from the newest committed PMC pass (profiles/r*/counters_<field>_<log n>.json,
    produced by tools/prof_counters.sh), else None"""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", f"counters_{field}_{log_n}.json")))
    if not files:
        return None, None
    try:
        with open(files[-1]) as f:
            return json.load(f), os.path.relpath(files[-1], ROOT)
    except (OSError, ValueError):
        return None, None


def build_roofline(args, F, n, classes, step_s, device):
    """`roofline` object of the bench line.  What binds this path is decided from evidence, not assumed:
      * HBM: REAL bytes per launch from rocprofv3 PMC counters (2 x FETCH_SIZE + WRITE_SIZE, gfx950 correction) over the
        launch time measured NOW with HIP events -> `achieved` / `frac` against the 8 TB/s line (top level = dominant kernel);
      * VALU: wave-instructions per launch from SQ_INSTS_VALU over the same time and the shader clock measured NOW under
        the multiply load -> issue cycles per instruction per SIMD, against the cycles the bare multiply chain needs at full
        occupancy on this chip (`valu_issue`), and executed field multiplies per second against that chain (`valu`);
      * the stage-streaming figure of SURVEY 8(d) (algorithmic bytes / time) is kept as `effective`: the fused kernels run
        >= 10 stages per HBM round trip, so it can exceed the HBM line — it is a figure of merit, never a fraction."""
    live = [c for c in classes if c["launches"]]
    if not live:
        return None
    dom = max(live, key=lambda c: c["ms"])
    ctr, src = _counters(args.field, args.log_n)
    # the committed counters describe the DEFAULT build and code path: with an A/B switch set (or another library) they belong to
    # other kernels than the ones this run times, so every counter-derived field stays empty instead of mixing epochs
    switches = [k for k in AB_SWITCHES if os.environ.get(k)]
    if switches:
        ctr, src = None, None
    cls_ctr = (ctr or {}).get("classes", {})
    per_class = []
    tot_bytes = tot_insts = 0.0
    complete = bool(cls_ctr)
    for c in live:
        lps = c["launches"] / args.steps
        us = c["ms"] * 1e3 / c["launches"]
        row = {"name": c["name"], "launches_per_step": lps, "avg_launch_us": us, "event_ms_per_step": c["ms"] / args.steps,
               "effective_GBs": c["alg_bytes"] / (c["ms"] * 1e-3) / 1e9, "alg_bytes_per_launch": c["alg_bytes"] / c["launches"]}
        k = cls_ctr.get(c["name"])
        if k and "hbm_bytes_per_launch" in k:
            row["hbm_bytes_per_launch"] = k["hbm_bytes_per_launch"]
            row["hbm_counter_GBs"] = k["hbm_bytes_per_launch"] / (us * 1e-6) / 1e9
            row["hbm_frac"] = row["hbm_counter_GBs"] / HBM_PEAK_GBS
            row["hbm_counter_GBs_solo"] = k["hbm_bytes_per_launch"] / (k.get("pmc_avg_us", us) * 1e-6) / 1e9
            tot_bytes += k["hbm_bytes_per_launch"] * lps
        else:
            complete = False
        if k and "SQ_INSTS_VALU" in k:
            row["valu_insts_per_launch"] = k["SQ_INSTS_VALU"]
            tot_insts += k["SQ_INSTS_VALU"] * lps
            if k.get("SQ_WAVE_CYCLES"):
                row["wave_cycles_waiting_frac"] = k.get("SQ_WAIT_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
                row["wave_cycles_issue_stalled_frac"] = k.get("SQ_WAIT_INST_ANY", 0.0) / k["SQ_WAVE_CYCLES"]
        else:
            complete = False
        per_class.append(row)
    drow = next(r for r in per_class if r["name"] == dom["name"])
    be, bx = b_alg(n, F.elem_bytes)
    cur_hash = kernel_source_hash()
    ctr_hash = (ctr or {}).get("kernel_source_hash")
    stale = bool(ctr) and (ctr_hash is None or cur_hash is None or ctr_hash != cur_hash)
    out = {"kernel": dom["name"], "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "achieved": drow.get("hbm_counter_GBs"), "frac": drow.get("hbm_frac"), "traffic": drow.get("hbm_bytes_per_launch"),
           # the SURVEY 8(d) formula, as the contract words it: ALGORITHMIC (stage-streaming) bytes of one launch of the dominant
           # kernel / its live launch time / 8 TB/s.  NOT a utilisation (>= 10 stages share one HBM round trip): see `effective`
           "achieved_alg": drow["effective_GBs"], "frac_alg": drow["effective_GBs"] / HBM_PEAK_GBS,
           "counters_kernel_hash": ctr_hash, "running_kernel_hash": cur_hash, "counters_stale": stale,
           "launches_per_step": drow["launches_per_step"], "avg_launch_us": drow["avg_launch_us"],
           "share_of_event_time": dom["ms"] / max(sum(c["ms"] for c in live), 1e-12),
           "effective": {"GBs": drow["effective_GBs"], "alg_bytes_per_launch": drow["alg_bytes_per_launch"],
                         "whole_job_GBs": (be + bx) / step_s / 1e9,
                         "alg_bytes_check": {"profiler_sum_per_step": sum(c["alg_bytes"] for c in live) / args.steps, "closed_form": be + bx},
                         "note": "ALGORITHMIC bytes of the stage-streaming model (SURVEY 8(d)) / measured time; >= 10 stages share one HBM "
                                 "round trip in the fused kernels, so this can exceed the HBM line and is not a fraction of it"},
           "per_class": per_class, "counters_source": src, "counters_skipped_for_switches": switches or None,
           "note": "achieved / frac / traffic: HBM bytes really moved per launch of the dominant kernel (rocprofv3 PMC, committed under "
                   "profiles/) over its launch time measured in this run with HIP events; achieved_alg / frac_alg: the same launch time under "
                   "SURVEY 8(d)'s algorithmic bytes; counters_stale = the PMC pass was taken from other kernel sources than the ones running"}
    try:
        clock_mhz = F.shader_clock_mhz(device)
        ceil4, ceil8 = F.mul_ceiling(4, device), F.mul_ceiling(8, device)
        xe_, xx_ = executed_mul(n)
        out["clock_mhz_under_multiply_load"] = clock_mhz
        tot_mfma = sum((cls_ctr.get(c["name"]) or {}).get("SQ_INSTS_VALU_MFMA_I8", 0.0) * c["launches"] / args.steps for c in live)
        # a 1024-element phase = 512 MFMAs.  In an EXTEND it stands for 7 sweeps x 1024 multiplies (14 per MFMA, mfma_blk16.h); the
        # low16 phases of k_enter_low / k_exit_low (n/2 MFMAs each per transform, launches of >= 2^18 elements) stand for levels 1..4 =
        # 19 n resp. 34 n multiplies of executed_mul's closed forms (38 / 68 per MFMA)
        # round 4: launches of < 2^18 elements run 256-element tiles whose phase is 256 v_mfma_i32_16x16x64_i8 (7 multiplies per MFMA) and
        # whose low16 phases are n MFMAs per transform; between 2^18 and 2^19 the two regimes mix and the split is approximate
        small = n < (1 << 18)
        low_on = tot_mfma > 0 and n >= (1 << 8) and not os.environ.get("ECFFT_NO_LOW16") and args.field != "m31"
        low16 = (n if small else n // 2) if low_on else 0
        on_mfma = (7.0 if small else 14.0) * max(tot_mfma - 2 * low16, 0.0) + ((19.0 + 34.0) * n if low_on else 0.0)
        valu_mul = max(xe_ + xx_ - on_mfma, 0.0)
        out["valu"] = {"executed_mul_per_step": xe_ + xx_, "of_which_on_matrix_cores": on_mfma, "valu_mul_per_step": valu_mul,
                       "achieved": valu_mul / step_s, "peak": max(ceil4, ceil8), "unit": "mul/s",
                       "frac": valu_mul / step_s / max(ceil4, ceil8), "peak_at_4_waves_per_simd": ceil4, "peak_at_max_occupancy": ceil8,
                       "note": "peak = ecfft_mul_ceiling: the kernels' OWN 169-instruction table multiply as a bare dependent chain on the whole "
                               "chip (8 workgroups per CU requested; the 94-VGPR chain fits 5 waves per SIMD) - a ceiling of this "
                               "implementation's multiply, not of the machine: see valu_machine for the instruction-issue utilisation"}
        if switches:     # no counters for this code path: how the multiplies split between the pipes is not known
            out["valu"].update({"of_which_on_matrix_cores": None, "valu_mul_per_step": None, "achieved": None, "frac": None,
                                "note": out["valu"]["note"] + "; counter-derived fields are empty: " + ", ".join(switches) + " set"})
        if tot_insts:
            ipm = INSTR_PER_MUL[args.field]
            floor = clock_mhz * 1e6 * N_SIMD * 64 / (max(ceil4, ceil8) * ipm)     # cycles per wave-instruction per SIMD of the bare chain
            measured = step_s * clock_mhz * 1e6 * N_SIMD / tot_insts
            out["valu_issue"] = {"valu_wave_insts_per_step": tot_insts, "cycles_per_inst_per_simd": measured,
                                 "floor_cycles_per_inst_per_simd": floor, "frac": floor / measured,
                                 "note": "SQ_INSTS_VALU summed over every launch of a step (PMC pass) against step time x measured clock x 1024 SIMDs; "
                                         "floor = the same quantity for the bare multiply chain (integer mad / carry instructions issue at ~4 cycles, "
                                         "tools/ubench/clock.hip)"}
    except Exception as ex:  # pragma: no cover
        out["valu"] = {"error": str(ex)}
    # machine-level pipe utilisation from the SQ counters (independent of any multiply model): busy cycles of the VALU and of
    # the matrix cores over step time x measured clock x 1024 SIMDs.  SQ_ACTIVE_INST_VALU counts 4-cycle quads, SQ_VALU_MFMA_BUSY_CYCLES cycles.
    try:
        clk = out.get("clock_mhz_under_multiply_load")
        act = sum((cls_ctr.get(c["name"]) or {}).get("SQ_ACTIVE_INST_VALU", 0.0) * c["launches"] / args.steps for c in live)
        mbusy = sum((cls_ctr.get(c["name"]) or {}).get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * c["launches"] / args.steps for c in live)
        if clk and act:
            simd_cycles = step_s * clk * 1e6 * N_SIMD
            out["valu_machine"] = {"busy_frac": 4.0 * act / simd_cycles, "SQ_ACTIVE_INST_VALU_per_step": act,
                                   "note": "4 x SQ_ACTIVE_INST_VALU (quad-cycles the VALU was issuing) / (step time x measured shader clock x 1024 SIMDs)"}
            out["mfma"] = {"busy_frac": mbusy / simd_cycles, "SQ_VALU_MFMA_BUSY_CYCLES_per_step": mbusy,
                           "instructions_per_step": sum((cls_ctr.get(c["name"]) or {}).get("SQ_INSTS_VALU_MFMA_I8", 0.0) * c["launches"] / args.steps for c in live),
                           "note": "v_mfma_i32_32x32x32_i8 (1024-element tiles) / v_mfma_i32_16x16x64_i8 (256-element tiles of small launches) of the innermost 16-point maps (mfma_blk16.h); the matrix pipe is a side channel here, "
                                   "the integer VALU still binds"}
    except Exception:  # pragma: no cover
        pass
    comp = compulsory_bytes(n, F.elem_bytes)
    out["compulsory"] = {"bytes_per_step": comp, "ms_at_peak": comp / (HBM_PEAK_GBS * 1e9) * 1e3,
                         "note": "in + out of both transforms + every needed table element once (SURVEY 8(d)); the floor of any implementation"}
    if tot_bytes and complete:
        out["whole_job"] = {"hbm_counter_GBs": tot_bytes / step_s / 1e9, "hbm_frac": tot_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                            "hbm_bytes_per_step": tot_bytes, "traffic_over_compulsory": tot_bytes / comp}
    # what binds: the larger of the two measured fractions
    hb = (out.get("whole_job") or {}).get("hbm_frac") or out.get("frac") or 0.0
    vb = max((out.get("valu_issue") or {}).get("frac") or 0.0, (out.get("valu") or {}).get("frac") or 0.0)
    out["bound"] = "valu" if vb >= hb else "hbm"
    if switches:
        out["bound"] = None                     # no counters for this code path: nothing measured decides it
    out["bound_evidence"] = {"hbm_frac_of_8TBs": hb, "valu_frac_of_multiply_chain": vb}
    return out


if __name__ == "__main__":
    main()
