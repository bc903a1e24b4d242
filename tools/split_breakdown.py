#!/usr/bin/env python3
"""Where a rank's time goes in the split ENTER / EXIT: per kernel class (launches, ms from HIP events) of rank 0's shard contexts over the
projection transport with zero-cost exchanges, next to the same classes of a single-GPU transform of the rank-local chunk size.
usage: split_breakdown.py [log_n] [P ...]"""
import os
import sys
import threading
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import ecfft_amd  # noqa: E402
from ecfft_amd import distributed as D  # noqa: E402
from ecfft_amd import fftree as FT  # noqa: E402
FT.use_hooks_library().__enter__()      # the projection transport is a measurement hook: hooks build only (include/ecfft_hip_hooks.h)

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
worlds = [int(w) for w in sys.argv[2:]] or [2, 8]
F = ecfft_amd.FIELDS["secp256k1"]
L = FT.lib()
n = 1 << log_n
rng = np.random.default_rng(7)
a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
x = torch.from_numpy(a.view(np.int64)).cuda()


def classes(ctx, fn, reps=5):
    fn(); fn()
    ctx.profile(True)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    rows = [(r["name"], r["launches"] / reps, r["ms"] / reps) for r in ctx.profile_read() if r["launches"]]
    ctx.profile(False)
    return rows


def show(tag, rows):
    tot = sum(r[2] for r in rows)
    print(f"  {tag}: {sum(r[1] for r in rows):.0f} launches, {tot:.3f} ms of kernel time: " + ", ".join(f"{nm} {ln:.0f} x = {ms:.3f}" for nm, ln, ms in rows))


for P in worlds:
    c = n // P
    print(f"P = {P}: chunk 2^{log_n - int(np.log2(P))}")
    loc = F.build_fftree(c)
    xc = x[:c].clone()
    show("single-GPU ENTER of the chunk size", classes(loc, lambda: loc.enter(xc)))
    show("single-GPU EXIT  of the chunk size", classes(loc, lambda: loc.exit(xc)))
    del loc
    board, bar, keep = {}, threading.Barrier(P), {}

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

    def run(rank):
        comm = D.Comm.callback(world=P, rank=rank, device=0, exchange=make_exchange(rank))
        esh = F.build_enter_shard(n, P, rank)
        xsh = F.build_exit_shard(n, comm)
        if rank == 0:
            keep["esh"], keep["xsh"] = esh, xsh
        bar.wait()

    th = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    esh, xsh = keep["esh"], keep["xsh"]
    comm = D.Comm.projection(P, 0, 0, 0.0, 0.0)
    show("split ENTER, rank 0", classes(esh, lambda: esh.enter_sharded(comm, xc, n)))
    show("split EXIT,  rank 0", classes(xsh, lambda: xsh.exit_sharded(comm, xc, n)))
    del esh, xsh, keep, comm
    torch.cuda.empty_cache()
