#!/usr/bin/env python3
"""Per-rank PROJECTION of ONE secp256k1 ENTER+EXIT split over P GPUs (the north-star partitioning) without the
hardware — VERDICT r03 item 3.

1. P sharded ENTER contexts and P sharded EXIT contexts (collective build) are driven as the P ranks of one process on one
   GPU with a data-moving emulated exchange: the split ENTER and the split EXIT of arbitrary evaluations must equal the single-GPU
   transforms bit for bit.
2. Rank 0's two contexts are then timed ON THEIR OWN over the measurement transport (ecfft_comm_init_projection): every exchange with
   a remote peer is a kernel on the stream that spins for `delay` microseconds (+ message bytes / link bandwidth), the rank's own
   buffers stand in for the peers' data.  The stream's timeline is that of a rank whose peers answer after exactly the modelled
   time, so  T(delay) - T(0)  is the EXPOSED communication time of the schedule as built (nothing overlaps an exchange today:
   it equals exchanges x delay; an overlapped schedule would show less), and T(0) is the per-rank compute incl. pack / unpack.

usage: split_project.py [log_n] [delay_us] [link GB/s]      (default 20 25 48)"""
import os
import sys
import threading
import time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ecfft_amd  # noqa: E402
from ecfft_amd import distributed as D  # noqa: E402
from ecfft_amd import fftree as FT  # noqa: E402
FT.use_hooks_library().__enter__()      # the projection transport is a measurement hook: hooks build only (include/ecfft_hip_hooks.h)

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
delay = float(sys.argv[2]) if len(sys.argv) > 2 else 25.0
gbps = float(sys.argv[3]) if len(sys.argv) > 3 else 48.0
worlds = [int(w) for w in os.environ.get("WORLDS", "2,4,8").split(",")]
reps = 7
F = ecfft_amd.FIELDS["secp256k1"]
L = FT.lib()
n = 1 << log_n
rng = np.random.default_rng(0x5EED0420)
a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
x = torch.from_numpy(a.view(np.int64)).cuda()


def med_ms(fn):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    return sorted(ts)[len(ts) // 2]


full = F.build_fftree(n)
want_ev = full.enter(x)
want_co = full.exit(x)                      # EXIT of arbitrary evaluations
for _ in range(2):
    full.exit(full.enter(x))
single_enter, single_exit = med_ms(lambda: full.enter(x)), med_ms(lambda: full.exit(x))
single = med_ms(lambda: full.exit(full.enter(x)))
del full
torch.cuda.empty_cache()
print(f"secp256k1 n=2^{log_n}: single GPU ENTER {single_enter:.3f} ms, EXIT {single_exit:.3f} ms, ENTER+EXIT {single:.3f} ms")
print(f"projection: {delay:.0f} us per exchange with a remote peer; link model {gbps:.0f} GB/s per peer (one message per xGMI link)")
print("  P | per-rank compute ms (ENTER + EXIT) | launches | exchanges (ENTER + EXIT) | MB sent/rank | T(delay) ms | exposed ms | exposed/exchange us | "
      "T(delay + bytes/bw) ms | speed-up vs 1 GPU at delay / at delay+bw | bit-exact")
rows = []
full_rows = []
for P in worlds:
    c = n // P
    board, bar = {}, threading.Barrier(P)

    def make_exchange(rank):
        def exchange(user, ns, speer, sptr, sbytes, nr, rpeer, rptr, rbytes, stream):
            L.ecfft_device_sync(0)
            board[rank] = [(speer[i], sptr[i], sbytes[i]) for i in range(ns)]
            bar.wait()
            for i in range(nr):
                src = [q for q in board[rpeer[i]] if q[0] == rank]
                k = sum(1 for j in range(i) if rpeer[j] == rpeer[i])
                L.ecfft_device_copy(rptr[i], src[k][1], rbytes[i], 2)
            L.ecfft_device_sync(0)
            bar.wait()
            return 0
        return exchange

    ok, keep = {}, {}

    def run(rank):
        comm = D.Comm.callback(world=P, rank=rank, device=0, exchange=make_exchange(rank))
        esh = F.build_enter_shard(n, P, rank)
        xsh = F.build_exit_shard(n, comm)
        mine = x[rank * c:(rank + 1) * c].clone()
        ev = esh.enter_sharded(comm, mine, n)
        co = xsh.exit_sharded(comm, mine, n)
        L.ecfft_device_sync(0)
        ok[rank] = bool(torch.equal(ev, want_ev[rank * c:(rank + 1) * c]) and torch.equal(co, want_co[rank * c:(rank + 1) * c]))
        if rank == 0:
            keep["esh"], keep["xsh"] = esh, xsh
        bar.wait()

    th = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    exact = all(ok.get(r, False) for r in range(P))
    torch.cuda.empty_cache()
    esh, xsh = keep["esh"], keep["xsh"]
    mine = x[:c].clone()
    res = {}
    for tag, d_us, bw in (("t0", 0.0, 0.0), ("td", delay, 0.0), ("tb", delay, gbps)):
        comm = D.Comm.projection(P, 0, 0, d_us, bw)
        run1 = lambda: xsh.exit_sharded(comm, esh.enter_sharded(comm, mine, n), n)      # noqa: E731
        run1(); run1()
        res[tag] = med_ms(run1)
        if tag == "t0":
            res["t0_enter"] = med_ms(lambda: esh.enter_sharded(comm, mine, n))
            res["t0_exit"] = med_ms(lambda: xsh.exit_sharded(comm, mine, n))
            comm.stats(True)
            esh.enter_sharded(comm, mine, n); st_e = comm.stats()
            xsh.exit_sharded(comm, mine, n); st_x = comm.stats()
            comm.stats(False)
            launches = 0
            for sh in (esh, xsh):
                sh.profile(True)
            esh.enter_sharded(comm, mine, n); xsh.exit_sharded(comm, mine, n); torch.cuda.synchronize()
            for sh in (esh, xsh):
                launches += sum(v["launches"] for v in sh.profile_read()); sh.profile(False)
        del comm
    nx = st_e["exchanges"] + st_x["exchanges"]
    exposed = res["td"] - res["t0"]
    rows.append((P, res, launches, st_e, st_x, exposed, exact))
    print(f"  {P} | {res['t0']:.3f} ({res['t0_enter']:.3f} + {res['t0_exit']:.3f}) | {launches:.0f} | {nx:.0f} ({st_e['exchanges']:.0f} + {st_x['exchanges']:.0f}) | "
          f"{(st_e['bytes_sent'] + st_x['bytes_sent']) / 1e6:.1f} | {res['td']:.3f} | {exposed:.3f} | {exposed * 1e3 / max(nx, 1):.1f} | {res['tb']:.3f} | "
          f"{single / res['td']:.2f}x / {single / res['tb']:.2f}x | {exact}")
    del xsh
    keep.pop("xsh", None)
    torch.cuda.empty_cache()
    # FULL context (tables replicated): the split EXIT is one all-gather + every top level redundantly (api_exit_split, n <= 2^21)
    fullc = F.build_fftree(n)
    resf = {}
    for tag, d_us, bw in (("t0", 0.0, 0.0), ("td", delay, 0.0), ("tb", delay, gbps)):
        comm = D.Comm.projection(P, 0, 0, d_us, bw)
        runx = lambda: fullc.exit_sharded(comm, mine, n)      # noqa: E731
        runx(); runx()
        resf[tag] = med_ms(runx)
        if tag == "t0":
            comm.stats(True); runx(); stf = comm.stats(); comm.stats(False)
        del comm
    # the path `bench.py --gpus P` times by default: ENTER on the shard context, EXIT in the form --split-exit auto picks
    # (n <= 2^21: the all-gather form on the full context; above: the EXIT-shard context of the first table)
    resd = None
    if log_n <= 21:
        resd = {}
        for tag, d_us, bw in (("t0", 0.0, 0.0), ("td", delay, 0.0), ("tb", delay, gbps)):
            comm = D.Comm.projection(P, 0, 0, d_us, bw)
            rund = lambda: fullc.exit_sharded(comm, esh.enter_sharded(comm, mine, n), n)      # noqa: E731
            rund(); rund()
            resd[tag] = med_ms(rund)
            del comm
        resd["exchanges"] = st_e["exchanges"] + stf["exchanges"]
        resd["mb"] = (st_e["bytes_sent"] + stf["bytes_sent"]) / 1e6
    full_rows.append((P, resf, stf, resd))
    del fullc, esh, keep
    torch.cuda.empty_cache()
print("(compute = rank 0's whole stream with zero-cost exchanges: local levels on the 2^%d chunk + its share of the log2 P top levels + pack / unpack;" % (log_n,))
print(" exchanges with world = P include the self pieces of the group all-to-alls; 'exposed' is measured on the stream, not computed)")
print("FULL contexts (tables replicated; default for n <= 2^21): split EXIT = one all-gather, then every top level redundantly on the block that contains the rank's chunk")
print("  P | EXIT per-rank compute ms | exchanges | MB sent/rank | EXIT T(delay) ms | EXIT T(delay + bytes/bw) ms | vs single-GPU EXIT %.3f ms" % single_exit)
for P, r, st, _ in full_rows:
    print(f"  {P} | {r['t0']:.3f} | {st['exchanges']:.0f} | {st['bytes_sent'] / 1e6:.1f} | {r['td']:.3f} | {r['tb']:.3f} | {single_exit / r['td']:.2f}x / {single_exit / r['tb']:.2f}x")
if any(rd for *_, rd in full_rows):
    print("THE PATH bench.py --gpus P TIMES (--split-exit auto): ENTER on the ENTER-shard context + EXIT as one all-gather on the full context")
    print("  P | per-rank compute ms | exchanges | MB sent/rank | T(delay) ms | exposed ms | T(delay + bytes/bw) ms | speed-up vs 1 GPU at delay / at delay+bw")
    for P, _, _, rd in full_rows:
        if rd:
            print(f"  {P} | {rd['t0']:.3f} | {rd['exchanges']:.0f} | {rd['mb']:.1f} | {rd['td']:.3f} | {rd['td'] - rd['t0']:.3f} | {rd['tb']:.3f} | {single / rd['td']:.2f}x / {single / rd['tb']:.2f}x")
