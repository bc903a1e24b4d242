"""P sharded EXTEND-only contexts driven as the P ranks of ONE process on one GPU (exchange = barrier + device-to-device copies):
per-rank COMPUTE time of a split EXTEND from the library's HIP events (the exchanges are emulated, so only the compute side
means anything), against the single-GPU EXTEND of the same size.  python tools/shard_emulate.py [field log_e world]"""
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

field = sys.argv[1] if len(sys.argv) > 1 else "secp256k1"
log_e = int(sys.argv[2]) if len(sys.argv) > 2 else 22
P = int(sys.argv[3]) if len(sys.argv) > 3 else 8
reps = 5
F = ecfft_amd.FIELDS[field]
e, c = 1 << log_e, (1 << log_e) // P
rng = np.random.default_rng(5)
if field == "m31":
    x = torch.from_numpy(rng.integers(0, 2**31 - 1, e, dtype=np.uint32).view(np.int32)).cuda()
else:
    a = rng.integers(0, 2**64, size=(e, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
    x = torch.from_numpy(a.view(np.int64)).cuda()
full = F.build_fftree(2 * e)
want = full.extend(x, ecfft_amd.Moiety.S1)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(reps):
    want = full.extend(x, ecfft_amd.Moiety.S1)
torch.cuda.synchronize(); single_ms = (time.perf_counter() - t0) * 1e3 / reps
del full
board, bar, L = {}, threading.Barrier(P), FT.lib()
turn = threading.Lock()      # one rank computes at a time, so a rank's HIP events time only its own kernels


def make_exchange(rank):
    def exchange(user, ns, speer, sptr, sbytes, nr, rpeer, rptr, rbytes, stream):
        L.ecfft_device_sync(0)
        turn.release()
        board[rank] = [(speer[i], sptr[i], sbytes[i]) for i in range(ns)]
        bar.wait()
        for i in range(nr):
            src = [q for q in board[rpeer[i]] if q[0] == rank]
            k = sum(1 for j in range(i) if rpeer[j] == rpeer[i])
            L.ecfft_device_copy(rptr[i], src[k][1], rbytes[i], 2)
        L.ecfft_device_sync(0)
        bar.wait()
        turn.acquire()
        return 0
    return exchange


res = {}


def run(rank):
    comm = D.Comm.callback(world=P, rank=rank, device=0, exchange=make_exchange(rank))
    shard = F.build_extend_shard(e, P, rank)
    mine = x[rank * c:(rank + 1) * c].clone()
    def ext(src, cyc):
        with turn:
            out = shard.extend_sharded(comm, src, e, ecfft_amd.Moiety.S1, cyclic_in=cyc, cyclic_out=cyc)
            L.ecfft_device_sync(0)
        return out

    for cyc in (False, True):
        src = x[rank::P].contiguous() if cyc else mine
        out = ext(src, cyc)      # warm-up
        ok = torch.equal(out, want[rank::P] if cyc else want[rank * c:(rank + 1) * c])
        if rank == 0:
            shard.profile(True)
        for _ in range(reps):
            ext(src, cyc)
        if rank == 0:
            cl = shard.profile_read(); shard.profile(False)
            res[cyc] = (ok, {v["name"]: (v["launches"] / reps, v["ms"] / reps) for v in cl if v["launches"]}, shard.device_bytes)
        else:
            res[(cyc, rank)] = ok


th = [threading.Thread(target=run, args=(r,)) for r in range(P)]
[t.start() for t in th]
[t.join() for t in th]
print(f"{field} EXTEND e=2^{log_e}: single GPU {single_ms:.3f} ms")
for cyc in (False, True):
    ok, cl, nb = res[cyc]
    allok = ok and all(v for k, v in res.items() if isinstance(k, tuple) and k[0] == cyc)
    tot = sum(v[1] for v in cl.values())
    print(f"  world {P} emulated, {'cyclic' if cyc else 'block'} in/out: rank-0 compute {tot:.3f} ms in {sum(v[0] for v in cl.values()):.0f} launches "
          f"({', '.join(f'{k} {v[0]:.0f}x {v[1] * 1e3:.0f}us' for k, v in cl.items())}); tables {nb / 2**20:.0f} MiB; bit-exact vs single GPU: {allok}")
