"""HBM held by a full context (tables of the chain T_1..T_2e + scratch) vs a sharded EXTEND-only context
(ecfft_build_extend_shard), and the build time of each.  python tools/shard_bytes.py [field log_e world ...]"""
import sys
import time
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecfft_amd  # noqa: E402

cases = [("secp256k1", 22, 8), ("secp256k1", 22, 2), ("m31", 24, 8)]
if len(sys.argv) > 3:
    cases = [(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))]
for field, log_e, world in cases:
    F = ecfft_amd.FIELDS[field]
    e = 1 << log_e
    t = time.time(); full = F.build_fftree(2 * e); tf = time.time() - t
    fb = full.device_bytes
    del full
    t = time.time(); sh = F.build_extend_shard(e, world, world - 1); ts = time.time() - t
    sb = sh.device_bytes
    print(f"{field} e=2^{log_e} world={world}: full context {fb / 2**20:.1f} MiB (build {tf:.2f} s)   shard context {sb / 2**20:.1f} MiB (build {ts:.2f} s)   ratio {fb / sb:.1f}x", flush=True)

for field, log_n, world in [("secp256k1", 22, 8), ("m31", 25, 8)] if len(sys.argv) <= 3 else []:
    F = ecfft_amd.FIELDS[field]
    n = 1 << log_n
    t = time.time(); full = F.build_fftree(n); tf = time.time() - t
    fb = full.device_bytes
    del full
    t = time.time(); sh = F.build_enter_shard(n, world, world - 1); ts = time.time() - t
    sb = sh.device_bytes
    print(f"{field} ENTER n=2^{log_n} world={world}: full context {fb / 2**20:.1f} MiB (build {tf:.2f} s)   ENTER-shard context {sb / 2**20:.1f} MiB (build {ts:.2f} s)   ratio {fb / sb:.1f}x", flush=True)

if len(sys.argv) <= 3:
    # EXIT-only shard contexts are a collective build: world ranks as threads of this process (exchange = barrier + D2D copies)
    import threading
    from ecfft_amd import distributed as D
    from ecfft_amd import fftree as FT
    for field, log_n, world in [("secp256k1", 22, 8), ("m31", 25, 8)]:
        F = ecfft_amd.FIELDS[field]
        n = 1 << log_n
        board, bar, L, out = {}, threading.Barrier(world), FT.lib(), {}

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
            comm = D.Comm.callback(world=world, rank=rank, device=0, exchange=make_exchange(rank))
            t = time.time(); sh = F.build_exit_shard(n, comm); out[rank] = (sh.device_bytes, time.time() - t)
            bar.wait()
        th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
        [t.start() for t in th]; [t.join() for t in th]
        full = F.build_fftree(n); fb = full.device_bytes; del full
        sb, ts = out[world - 1]
        print(f"{field} EXIT n=2^{log_n} world={world}: full context {fb / 2**20:.1f} MiB   EXIT-shard context {sb / 2**20:.1f} MiB per rank "
              f"(collective build of all {world} ranks on this one GPU: {ts:.2f} s)   ratio {fb / sb:.1f}x", flush=True)
