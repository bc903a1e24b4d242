"""Multi-process tests of the N>1 paths on CPU (gloo): the split-EXTEND orchestration
(tests/split_model.py: block<->cyclic all_to_all_single + local stage calls) at world sizes 2 and 4."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_extend_sharded_gloo(world, oracle_mod):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "DIST_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_sharded_transforms_with_hip_ops(world):
    """the whole multi-GPU path with the real HIP local ops: `world` ranks sharing cuda:0, gloo for the all-to-alls
    (staged through host memory); results must equal the single-GPU transforms bit for bit"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker_gpu.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DIST_GPU_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_sharded_transforms_over_rccl_multi_gpu():
    """the same worker with ONE RANK PER GPU over the real RCCL transport (grouped ncclSend / ncclRecv over xGMI): full and sharded
    contexts, block and cyclic layouts, against the single-GPU transforms.  Needs >= 2 GPUs (skipped on the one-GPU lease)."""
    import torch
    ndev = torch.cuda.device_count()
    if ndev < 2:
        pytest.skip("one GPU: RCCL refuses several ranks on one device (covered with world = 1 below and with gloo above)")
    world = 1 << (min(ndev, 8).bit_length() - 1)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", ECFFT_WORKER_RCCL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker_gpu.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DIST_GPU_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def _stub_rccl():
    """tests/stub_rccl/librccl_stub.so (built by __graft_entry__.build(); rebuilt here if missing)"""
    lib = os.path.join(ROOT, "tests", "stub_rccl", "librccl_stub.so")
    if not os.path.exists(lib):
        subprocess.run(["/opt/rocm/bin/hipcc", "-O2", "-fPIC", "-shared", "-o", lib, os.path.join(ROOT, "tests", "stub_rccl", "stub_rccl.cpp"), "-lrt"], check=True)
    return lib


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4])
def test_rccl_transport_with_several_ranks_through_the_stub_library(world):
    """round 4: the RcclTransport code path (grouped ncclSend / ncclRecv with several peers per group, sub-group peer lists of the
    split ENTER / EXIT levels, Transport::vote, the collective ecfft_build_exit_shard) with world = 2 and 4 on ONE GPU: librccl is
    replaced by a test-only stand-in (tests/stub_rccl: the entry points transport.h binds, bytes staged through POSIX shared
    memory between the processes) handed to the library with ecfft_comm_set_rccl_library.  Same worker and checks as the real multi-GPU test."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", ECFFT_WORKER_RCCL="stub", ECFFT_WORKER_RCCL_LIB=_stub_rccl())
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker_gpu.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DIST_GPU_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_comm_abort_unblocks_a_rank_whose_peer_never_arrives():
    """ecfft_comm_abort (ncclCommAbort): rank 0 of a two-rank communicator starts a split EXTEND whose peer never makes the call; a
    second host thread aborts the communicator and the blocked call returns an error instead of hanging (stub library: the peer
    process attaches and then just sleeps; tests/abort_worker.py)."""
    env = dict(os.environ, ECFFT_WORKER_RCCL_LIB=_stub_rccl(), ECFFT_STUB_RCCL_TIMEOUT_S="60")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "abort_worker.py"), "main"], env=env, capture_output=True, text=True, timeout=300)
    assert "RETURNED_ERROR" in r.stdout and "abort -> True" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.gpu
def test_rccl_worker_with_one_rank():
    """the multi-GPU worker in its RCCL mode (nccl process group, Comm.rccl) with world = 1: keeps the script the multi-GPU test
    launches exercised on the one-GPU lease"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", ECFFT_WORKER_RCCL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dist_worker_gpu.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "DIST_GPU_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.gpu
def test_sharded_transforms_over_rccl_single_rank():
    """the C++ sharded path over the REAL RCCL transport (librccl loaded at run time, ncclGetUniqueId / ncclCommInitRank,
    grouped ncclSend / ncclRecv on the HIP stream).  The GPU box has one GPU, so world = 1: every exchange is a self
    send / receive and log P = 0, but communicator creation, the pack / unpack operators and all four exchange calls of a
    split EXTEND really run on RCCL.  Results must equal the single-GPU transforms."""
    import torch
    import ecfft_amd
    from ecfft_amd import distributed as D
    comm = D.Comm.rccl(device=0, world=1, rank=0)
    assert comm.world == 1 and comm.rank == 0
    comm.stats(True)
    for field, n in (("secp256k1", 1 << 12), ("m31", 1 << 15)):
        tree = ecfft_amd.FIELDS[field].build_fftree(2 * n)
        rng = np.random.default_rng(3)
        if field == "m31":
            x = torch.from_numpy(rng.integers(0, 2**31 - 1, n, dtype=np.uint32).view(np.int32)).cuda()
        else:
            a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
            x = torch.from_numpy(a.view(np.int64)).cuda()
        for moiety in (ecfft_amd.Moiety.S1, ecfft_amd.Moiety.S0):
            assert torch.equal(tree.extend_sharded(comm, x, n, moiety), tree.extend(x, moiety))
        assert torch.equal(tree.enter_sharded(comm, x, n), tree.enter(x))
        assert torch.equal(tree.exit_sharded(comm, x, n), tree.exit(x))
        shard = ecfft_amd.FIELDS[field].build_extend_shard(n, 1, 0)             # EXTEND-only context, world = 1: the whole tables
        for moiety in (ecfft_amd.Moiety.S1, ecfft_amd.Moiety.S0):
            assert torch.equal(shard.extend_sharded(comm, x, n, moiety), tree.extend(x, moiety))
    st = comm.stats()
    assert st["exchanges"] >= 16 and st["bytes_sent"] > 0 and st["comm_ms"] > 0      # 4 per EXTEND x 2 moieties x 2 fields


@pytest.mark.gpu
@pytest.mark.parametrize("field,e,log_p", [("secp256k1", 1 << 12, 1), ("secp256k1", 1 << 13, 3), ("m31", 1 << 15, 2), ("secp256k1", 1 << 8, 2),
                                           ("secp256k1", 1 << 22, 3)])   # last: BASELINE configs[3] size, P = 8
def test_split_extend_building_blocks_on_one_gpu(oracle_mod, field, e, log_p):
    """the HIP shard kernels (cyclic top stages with strided tables, block-local fused stages with
    k >= log P) emulating P ranks sequentially on one GPU == the single-GPU EXTEND, bit for bit"""
    import ecfft_amd
    P = 1 << log_p
    Fp = ecfft_amd.FIELDS[field]
    t = Fp.build_fftree(2 * e)
    rng = np.random.default_rng(5)
    if field == "m31":
        x = rng.integers(0, 2**31 - 1, e, dtype=np.uint32)
    elif e > (1 << 16):                                   # any 256-bit pattern below p is a valid element (top bit cleared)
        x = rng.integers(0, 2**64, size=(e, 4), dtype=np.uint64); x[:, 3] >>= np.uint64(1)
    else:
        x = oracle_mod.field(field).from_ints([int.from_bytes(rng.bytes(32), "little") % (2**256 - 2**32 - 977) for _ in range(e)])
    for moiety in (ecfft_amd.Moiety.S1, ecfft_amd.Moiety.S0):
        expect = t.extend(x, moiety)
        c = e // P
        cyc = [np.ascontiguousarray(x[r::P]) for r in range(P)]                 # block -> cyclic
        for r in range(P):
            t.extend_top_cyclic(cyc[r], e, moiety, log_p, r, False)
        full = np.empty_like(x)
        for r in range(P):
            full[r::P] = cyc[r]
        blk = [np.ascontiguousarray(full[r * c:(r + 1) * c]) for r in range(P)]  # cyclic -> block
        for r in range(P):
            t.extend_local_block(blk[r], e, moiety, log_p)
        full = np.concatenate(blk)
        cyc = [np.ascontiguousarray(full[r::P]) for r in range(P)]
        for r in range(P):
            t.extend_top_cyclic(cyc[r], e, moiety, log_p, r, True)
        for r in range(P):
            full[r::P] = cyc[r]
        assert np.array_equal(full, expect)


@pytest.mark.gpu
@pytest.mark.parametrize("field,e,log_p", [("secp256k1", 1 << 12, 1), ("secp256k1", 1 << 13, 3), ("m31", 1 << 15, 2), ("m31", 1 << 20, 3), ("secp256k1", 1 << 8, 2),
                                           ("secp256k1", 1 << 22, 3)])   # last: BASELINE configs[3] — e = 2^22 over P = 8, all four layouts
def test_extend_shard_context_on_one_gpu(field, e, log_p):
    """ecfft_build_extend_shard: P sharded EXTEND-only contexts (each holding only its rank's table entries) driven as P
    ranks of one process — a callback transport whose exchange is a barrier + device-to-device copies between the ranks'
    buffers — == the single-GPU EXTEND of a full context, bit for bit; every other call on such a context is refused."""
    import threading
    import torch
    import ecfft_amd
    from ecfft_amd import distributed as D
    from ecfft_amd import fftree as FT
    F = ecfft_amd.FIELDS[field]
    P, c = 1 << log_p, e >> log_p
    full_tree = F.build_fftree(2 * e)
    rng = np.random.default_rng(5)
    if field == "m31":
        x = torch.from_numpy(rng.integers(0, 2**31 - 1, e, dtype=np.uint32).view(np.int32)).cuda()
    else:
        a = rng.integers(0, 2**64, size=(e, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
        x = torch.from_numpy(a.view(np.int64)).cuda()
    want = {m: full_tree.extend(x, m) for m in (ecfft_amd.Moiety.S1, ecfft_amd.Moiety.S0)}
    torch.cuda.synchronize()
    board, bar, L = {}, threading.Barrier(P), FT.lib()

    def make_exchange(rank):
        def exchange(user, ns, speer, sptr, sbytes, nr, rpeer, rptr, rbytes, stream):
            try:
                L.ecfft_device_sync(0)
                board[rank] = [(speer[i], sptr[i], sbytes[i]) for i in range(ns)]
                bar.wait(timeout=60)
                for i in range(nr):
                    src = [q for q in board[rpeer[i]] if q[0] == rank]
                    k = sum(1 for j in range(i) if rpeer[j] == rpeer[i])       # k-th message from that peer
                    assert src[k][2] == rbytes[i]
                    assert L.ecfft_device_copy(rptr[i], src[k][1], rbytes[i], 2) == 0
                L.ecfft_device_sync(0)
                bar.wait(timeout=60)
                return 0
            except Exception as ex:     # noqa: BLE001 — reported through the return code
                print("exchange failed:", ex, flush=True)
                bar.abort()
                return 1
        return exchange

    got, errs = {}, []

    def run(rank):
        try:
            comm = D.Comm.callback(world=P, rank=rank, device=0, exchange=make_exchange(rank))
            shard = F.build_extend_shard(e, P, rank)
            assert shard is not None and shard.n == 2 * e
            mine = x[rank * c:(rank + 1) * c].clone()
            for m in want:
                got[(rank, int(m))] = shard.extend_sharded(comm, mine, e, m)
                cyc = x[rank::P].contiguous()                                   # cyclic shard: local j' = global j' * P + rank
                got[(rank, int(m), "cb")] = shard.extend_sharded(comm, cyc, e, m, cyclic_in=True)
                got[(rank, int(m), "bc")] = shard.extend_sharded(comm, mine, e, m, cyclic_out=True)
                got[(rank, int(m), "cc")] = shard.extend_sharded(comm, cyc, e, m, cyclic_in=True, cyclic_out=True)
            assert L.ecfft_extend(shard._h, mine.data_ptr(), mine.data_ptr(), c, 1, 1, 1, None) == FT.ERR_BAD_ARG
            assert L.ecfft_enter_sharded(shard._h, comm._h, mine.data_ptr(), mine.data_ptr(), e, None) == FT.ERR_BAD_ARG
            half = mine[: c // 2].clone()                                       # a different e than the context was built for
            assert L.ecfft_extend_sharded(shard._h, comm._h, half.data_ptr(), half.data_ptr(), e // 2, 1, None) == FT.ERR_BAD_ARG
        except Exception as ex:         # noqa: BLE001
            errs.append((rank, repr(ex)))
            bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    torch.cuda.synchronize()
    for m, w in want.items():
        for r in range(P):
            assert torch.equal(got[(r, int(m))], w[r * c:(r + 1) * c]), (field, e, log_p, r, m)
            assert torch.equal(got[(r, int(m), "cb")], w[r * c:(r + 1) * c]), ("cyclic in", field, e, log_p, r, m)
            assert torch.equal(got[(r, int(m), "bc")], w[r::P]), ("cyclic out", field, e, log_p, r, m)
            assert torch.equal(got[(r, int(m), "cc")], w[r::P]), ("cyclic in and out", field, e, log_p, r, m)


def _thread_ranks(P, body):
    """run body(rank, make_comm) on P threads; make_comm() gives a callback communicator whose exchange is a barrier +
    device-to-device copies between the ranks' buffers (all ranks share cuda:0)"""
    import threading
    from ecfft_amd import distributed as D
    from ecfft_amd import fftree as FT
    board, bar, L = {}, threading.Barrier(P), FT.lib()

    def make_exchange(rank):
        def exchange(user, ns, speer, sptr, sbytes, nr, rpeer, rptr, rbytes, stream):
            try:
                L.ecfft_device_sync(0)
                board[rank] = [(speer[i], sptr[i], sbytes[i]) for i in range(ns)]
                bar.wait(timeout=120)
                for i in range(nr):
                    src = [q for q in board[rpeer[i]] if q[0] == rank]
                    k = sum(1 for j in range(i) if rpeer[j] == rpeer[i])
                    assert src[k][2] == rbytes[i]
                    assert L.ecfft_device_copy(rptr[i], src[k][1], rbytes[i], 2) == 0
                L.ecfft_device_sync(0)
                bar.wait(timeout=120)
                return 0
            except Exception as ex:     # noqa: BLE001 — reported through the return code
                print("exchange failed:", ex, flush=True)
                bar.abort()
                return 1
        return exchange

    errs = []

    def run(rank):
        try:
            body(rank, lambda: D.Comm.callback(world=P, rank=rank, device=0, exchange=make_exchange(rank)))
        except Exception as ex:         # noqa: BLE001
            import traceback
            errs.append((rank, repr(ex), traceback.format_exc()))
            bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(P)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs


@pytest.mark.gpu
@pytest.mark.parametrize("field,e,P", [("secp256k1", 1 << 13, 4), ("m31", 1 << 16, 8), ("secp256k1", 1 << 12, 2)])
def test_full_context_fuses_its_cyclic_stages(field, e, P, monkeypatch, hooks_lib):
    """round 3: a FULL context gathers the stride-P entries of the cyclic stages into compact tables on first use, so its split
    EXTEND runs the log P cyclic stages as one fused column pass each way (no k_decompose_stage / k_recombine_stage launch) like a
    shard context does; ECFFT_NO_FULL_CYCLIC=1 keeps the one-stage launches.  All four layouts, both forms == the single-GPU EXTEND."""
    import torch
    import ecfft_amd
    Fp = ecfft_amd.FIELDS[field]
    c = e // P
    rng = np.random.default_rng(0xF0CC + P)
    if field == "m31":
        x = torch.from_numpy(rng.integers(0, 2**31 - 1, e, dtype=np.uint32).view(np.int32)).cuda()
    else:
        a = rng.integers(0, 2**64, size=(e, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
        x = torch.from_numpy(a.view(np.int64)).cuda()
    ref = Fp.build_fftree(2 * e)
    expect = {m: ref.extend(x, m) for m in (ecfft_amd.Moiety.S1, ecfft_amd.Moiety.S0)}
    torch.cuda.synchronize()
    one_stage = {}

    for fused in (True, False):
        if not fused:
            monkeypatch.setenv("ECFFT_NO_FULL_CYCLIC", "1")
        counts = {}

        def body(rank, make_comm):
            comm = make_comm()
            t = Fp.build_fftree(2 * e)                      # the env switch is read when a context is built
            t.profile(True)
            for m in (ecfft_amd.Moiety.S1, ecfft_amd.Moiety.S0):
                for ci in (False, True):
                    for co in (False, True):
                        mine = (x[rank::P] if ci else x[rank * c:(rank + 1) * c]).contiguous().clone()
                        got = t.extend_sharded(comm, mine, e, m, cyclic_in=ci, cyclic_out=co)
                        want = expect[m][rank::P] if co else expect[m][rank * c:(rank + 1) * c]
                        assert torch.equal(got.reshape(want.shape), want), (rank, int(m), ci, co)
            cls = {k["name"]: k["launches"] for k in t.profile_read()}
            held = t.device_bytes
            t.trim()                                        # the gathered tables are given back by ecfft_ctx_trim (pinned temporaries stay)
            counts[rank] = (cls["k_decompose_stage"] + cls["k_recombine_stage"], held - t.device_bytes)
            mine = x[rank * c:(rank + 1) * c].contiguous().clone()               # after the trim the tables are gathered again
            got = t.extend_sharded(comm, mine, e, ecfft_amd.Moiety.S1)
            assert torch.equal(got.reshape(mine.shape), expect[ecfft_amd.Moiety.S1][rank * c:(rank + 1) * c])

        _thread_ranks(P, body)
        one_stage[fused] = counts
    logp = P.bit_length() - 1
    for rank in range(P):
        assert one_stage[True][rank][0] == 0 and one_stage[True][rank][1] > one_stage[False][rank][1]
        assert one_stage[False][rank][0] == 8 * 2 * logp            # 8 calls x (log P decompose + log P recombine) launches


@pytest.mark.gpu
@pytest.mark.parametrize("field,n,P", [("secp256k1", 1 << 12, 2), ("secp256k1", 1 << 13, 4), ("m31", 1 << 16, 8), ("m31", 1 << 20, 4), ("secp256k1", 1 << 9, 8),
                                       ("secp256k1", 1 << 17, 8), ("m31", 1 << 22, 16), ("m31", 1 << 21, 2)])    # chunk 2^20 runs the two-halves schedule
# (the BASELINE metric's size, secp256k1 2^20 over P = 8, is held against the CPU oracle directly: tests/test_gpu_parity.py::test_shard_contexts_2e20_over_8_ranks_vs_oracle)
def test_enter_shard_context_on_one_gpu(field, n, P):
    """ecfft_build_enter_shard: P sharded ENTER-only contexts (chain up to n/P + the rank's share of the log2 P top trees) driven
    as the ranks of one process == the single-GPU ENTER of a full context, bit for bit; smaller HBM footprint; other calls refused"""
    import torch
    import ecfft_amd
    from ecfft_amd import fftree as FT
    F = ecfft_amd.FIELDS[field]
    c = n // P
    full_tree = F.build_fftree(n)
    rng = np.random.default_rng(6)
    if field == "m31":
        x = torch.from_numpy(rng.integers(0, 2**31 - 1, n, dtype=np.uint32).view(np.int32)).cuda()
    else:
        a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
        x = torch.from_numpy(a.view(np.int64)).cuda()
    want = full_tree.enter(x)
    full_bytes = full_tree.device_bytes
    torch.cuda.synchronize()
    got, L = {}, FT.lib()

    def body(rank, make_comm):
        comm = make_comm()
        shard = F.build_enter_shard(n, P, rank)
        assert shard is not None and shard.n == n
        mine = x[rank * c:(rank + 1) * c].clone()
        got[rank] = shard.enter_sharded(comm, mine, n)
        got[("bytes", rank)] = shard.device_bytes
        assert L.ecfft_enter(shard._h, mine.data_ptr(), mine.data_ptr(), c, 1, None) == FT.ERR_BAD_ARG
        assert L.ecfft_exit_sharded(shard._h, comm._h, mine.data_ptr(), mine.data_ptr(), n, None) == FT.ERR_BAD_ARG
        assert L.ecfft_extend_sharded(shard._h, comm._h, mine.data_ptr(), mine.data_ptr(), n // 2, 1, None) == FT.ERR_BAD_ARG

    _thread_ranks(P, body)
    torch.cuda.synchronize()
    for r in range(P):
        assert torch.equal(got[r], want[r * c:(r + 1) * c]), (field, n, P, r)
        if n >= 1 << 16:
            assert got[("bytes", r)] < full_bytes


@pytest.mark.gpu
@pytest.mark.parametrize("field,n,P", [("secp256k1", 1 << 12, 2), ("secp256k1", 1 << 13, 4), ("m31", 1 << 16, 8), ("m31", 1 << 20, 4), ("secp256k1", 1 << 9, 8),
                                       ("secp256k1", 1 << 17, 8), ("m31", 1 << 22, 16), ("m31", 1 << 21, 2)])    # chunk 2^20 runs the two-halves schedule
# (the BASELINE metric's size, secp256k1 2^20 over P = 8, is held against the CPU oracle directly: tests/test_gpu_parity.py::test_shard_contexts_2e20_over_8_ranks_vs_oracle)
def test_exit_shard_context_on_one_gpu(field, n, P):
    """ecfft_build_exit_shard (collective, distributed build of z0z0_rem_xnn_s): P sharded EXIT-only contexts as the ranks of one
    process == the single-GPU EXIT of a full context on arbitrary evaluations, bit for bit; other calls refused"""
    import torch
    import ecfft_amd
    from ecfft_amd import fftree as FT
    F = ecfft_amd.FIELDS[field]
    c = n // P
    full_tree = F.build_fftree(n)
    rng = np.random.default_rng(8)
    if field == "m31":
        x = torch.from_numpy(rng.integers(0, 2**31 - 1, n, dtype=np.uint32).view(np.int32)).cuda()
    else:
        a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
        x = torch.from_numpy(a.view(np.int64)).cuda()
    want = full_tree.exit(x)
    full_bytes = full_tree.device_bytes
    torch.cuda.synchronize()
    got, L = {}, FT.lib()

    def body(rank, make_comm):
        comm = make_comm()
        shard = F.build_exit_shard(n, comm)
        assert shard is not None and shard.n == n
        mine = x[rank * c:(rank + 1) * c].clone()
        got[rank] = shard.exit_sharded(comm, mine, n)
        got[("bytes", rank)] = shard.device_bytes
        assert L.ecfft_exit(shard._h, mine.data_ptr(), mine.data_ptr(), c, 1, None) == FT.ERR_BAD_ARG
        assert L.ecfft_enter_sharded(shard._h, comm._h, mine.data_ptr(), mine.data_ptr(), n, None) == FT.ERR_BAD_ARG

    _thread_ranks(P, body)
    torch.cuda.synchronize()
    for r in range(P):
        assert torch.equal(got[r], want[r * c:(r + 1) * c]), (field, n, P, r)
        if n >= 1 << 16 and P > 2:       # P = 2: the pair level runs on the full tree T_n (round 4), the context is the whole chain
            assert got[("bytes", r)] < full_bytes
        if P == 2 and n >= 1 << 16:
            assert got[("bytes", r)] <= 1.15 * full_bytes       # the whole chain + the pinned temporaries of the sharded call


@pytest.mark.gpu
@pytest.mark.parametrize("field,n,P", [("secp256k1", 1 << 13, 2), ("secp256k1", 1 << 14, 4), ("m31", 1 << 18, 8)])
def test_split_exit_runs_its_pair_level_redundantly_with_one_exchange(field, n, P, monkeypatch, hooks_lib):
    """round 4: the lowest top level of a split EXIT — groups of two ranks, blocks of 2n/P — is one exchange
    (each rank gets its partner's share) and the single-GPU EXIT level of the block on BOTH ranks, instead of four split EXTENDs
    with eight exchanges plus the re-blocking one.  Exchanges per EXIT: 1 + 9 (log2 P - 1) + 1 instead of 1 + 9 log2 P — checked on
    a full context against its own split form (ECFFT_SPLIT_Q2_SPLIT=1) and on EXIT-shard contexts (which now carry T_2n/P for it),
    bit for bit against the single-GPU EXIT of arbitrary evaluations."""
    import torch
    import ecfft_amd
    F = ecfft_amd.FIELDS[field]
    c = n // P
    full_tree = F.build_fftree(n)
    rng = np.random.default_rng(8)
    if field == "m31":
        x = torch.from_numpy(rng.integers(0, 2**31 - 1, n, dtype=np.uint32).view(np.int32)).cuda()
    else:
        a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
        x = torch.from_numpy(a.view(np.int64)).cuda()
    want = full_tree.exit(x)
    torch.cuda.synchronize()
    logp = P.bit_length() - 1
    counts = {}
    bytes_held = {}
    for mode in ("full-gather", "full", "full-split-pairs", "shard", "shard-min-memory", "shard-one-rank-short"):
        # full-gather: the default of a FULL context at these sizes — one all-gather, then every top level redundantly
        if mode in ("full", "full-split-pairs"):
            monkeypatch.setenv("ECFFT_SPLIT_GATHER_MAX_LOG", "0")
        if mode == "full-split-pairs":
            monkeypatch.setenv("ECFFT_SPLIT_Q2_SPLIT", "1")
        if mode == "shard-one-rank-short":
            monkeypatch.setenv("ECFFT_TEST_PAIR_SPLIT_RANK", str(P - 1))
        got, nx = {}, {}

        def body(rank, make_comm):
            comm = make_comm()
            # the environment is read when a context is built.  shard-min-memory: ECFFT_EXIT_SHARD_MIN_MEMORY on every rank;
            # shard-one-rank-short: the LAST rank reports that T_2c does not fit (test switch) — the ranks agree, every rank splits
            ctx = F.build_exit_shard(n, comm, min_memory=(mode == "shard-min-memory")) if mode.startswith("shard") else F.build_fftree(n)
            bytes_held[(mode, rank)] = ctx.device_bytes
            mine = x[rank * c:(rank + 1) * c].clone()
            comm.stats(True)
            got[rank] = ctx.exit_sharded(comm, mine, n)
            nx[rank] = comm.stats()["exchanges"]

        _thread_ranks(P, body)
        monkeypatch.delenv("ECFFT_SPLIT_Q2_SPLIT", raising=False); monkeypatch.delenv("ECFFT_SPLIT_GATHER_MAX_LOG", raising=False)
        monkeypatch.delenv("ECFFT_TEST_PAIR_SPLIT_RANK", raising=False)
        for rank in range(P):
            assert torch.equal(got[rank], want[rank * c:(rank + 1) * c]), (mode, rank)
        counts[mode] = nx[0]
    assert counts["full-gather"] == 1
    assert counts["full-split-pairs"] == 1 + 9 * logp
    assert counts["full"] == counts["shard"] == 1 + 9 * (logp - 1) + 1
    # the redundant pair level is optional in a shard context.  Without T_2c the level is split again (9 exchanges) and the
    # context holds no tree above T_n/P: at P = 2 that is where sharding saves memory at all
    assert counts["shard-min-memory"] == counts["shard-one-rank-short"] == 1 + 9 * logp
    for r in range(P):
        assert bytes_held[("shard-min-memory", r)] < 0.9 * bytes_held[("shard", r)], (r, bytes_held)     # small n: fixed-size tables (low16, blk16) weigh in
        assert bytes_held[("shard-one-rank-short", r)] <= 1.02 * bytes_held[("shard-min-memory", r)]
    print("bytes held (rank 0):", {m: bytes_held[(m, 0)] for m in ("full", "shard", "shard-min-memory")})
    assert counts["full"] <= 0.8 * counts["full-split-pairs"] or logp > 2          # >= 20 % fewer exchange latencies (P = 2: 10 -> 2, P = 4: 19 -> 11, P = 8: 28 -> 20)


@pytest.mark.gpu
@pytest.mark.parametrize("field,n,P", [("secp256k1", 1 << 14, 4), ("secp256k1", 1 << 15, 8), ("m31", 1 << 18, 8)])
def test_link_striping_of_the_pairwise_exchanges_is_bit_exact(field, n, P, monkeypatch, hooks_lib):
    """round 5: the big PAIRWISE exchanges of a split ENTER / EXIT (the level's re-distribution, the pair level, small-group
    all-to-alls) travel striped over every link of the mesh — slice k of a message via rank k, two grouped exchanges
    (Transport::exchange_striped) — when that takes at least the communicator's threshold off the most loaded link (opt-in since round 6:
    ecfft_comm_set_link_striping; the projection suggests 4 MiB).  Here the threshold is 0 (test switch), so
    every eligible exchange of these small transforms is striped: results must equal the single-GPU transforms bit for bit, on
    shard contexts and on a full context, and more exchanges must have been issued than without striping."""
    import torch
    import ecfft_amd
    F = ecfft_amd.FIELDS[field]
    c = n // P
    full_tree = F.build_fftree(n)
    rng = np.random.default_rng(77)
    if field == "m31":
        x = torch.from_numpy(rng.integers(0, 2**31 - 1, n, dtype=np.uint32).view(np.int32)).cuda()
    else:
        a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
        x = torch.from_numpy(a.view(np.int64)).cuda()
    want_ev, want_co = full_tree.enter(x), full_tree.exit(x)
    torch.cuda.synchronize()
    monkeypatch.setenv("ECFFT_SPLIT_GATHER_MAX_LOG", "0")              # full contexts: the split levels, not the all-gather form
    counts = {}
    for mode in ("plain", "striped"):
        if mode == "striped":
            monkeypatch.setenv("ECFFT_STRIPE_MIN_GAIN", "0")
        else:
            monkeypatch.setenv("ECFFT_NO_STRIPE", "1")
        got = {}

        def body(rank, make_comm):
            comm = make_comm()
            esh, xsh, fc = F.build_enter_shard(n, P, rank), F.build_exit_shard(n, comm), F.build_fftree(n)
            mine = x[rank * c:(rank + 1) * c].clone()
            comm.stats(True)
            got[("enter", rank)] = esh.enter_sharded(comm, mine, n)
            got[("exit", rank)] = xsh.exit_sharded(comm, mine, n)
            got[("enter-full", rank)] = fc.enter_sharded(comm, mine, n)
            got[("exit-full", rank)] = fc.exit_sharded(comm, mine, n)
            got[("nx", rank)] = comm.stats()["exchanges"]

        _thread_ranks(P, body)
        monkeypatch.delenv("ECFFT_STRIPE_MIN_GAIN", raising=False); monkeypatch.delenv("ECFFT_NO_STRIPE", raising=False)
        for r in range(P):
            sl = slice(r * c, (r + 1) * c)
            assert torch.equal(got[("enter", r)], want_ev[sl]) and torch.equal(got[("enter-full", r)], want_ev[sl]), (mode, r)
            assert torch.equal(got[("exit", r)], want_co[sl]) and torch.equal(got[("exit-full", r)], want_co[sl]), (mode, r)
        counts[mode] = got[("nx", 0)]
    assert counts["striped"] > counts["plain"], counts


@pytest.mark.gpu
def test_link_striping_threshold_through_the_abi():
    """ecfft_comm_set_link_striping on the SHIPPED library (no environment switch): 0 stripes every exchange that striping makes
    lighter, SIZE_MAX none, the default (off since round 6) none — same bits in all three, more grouped exchanges only with 0.  Also: the
    threshold is frozen once the communicator has carried an exchange (ECFFT_ERR_BAD_ARG)."""
    import torch
    import ecfft_amd
    F = ecfft_amd.FIELDS["secp256k1"]
    n, P = 1 << 15, 8
    c = n // P
    a = np.random.default_rng(5).integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
    x = torch.from_numpy(a.view(np.int64)).cuda()
    want = F.build_fftree(n).enter(x)
    torch.cuda.synchronize()
    counts = {}
    for name, thr in (("default", None), ("always", 0), ("never", (1 << 64) - 1)):
        got = {}

        def body(rank, make_comm):
            comm = make_comm()
            if thr is not None:
                comm.set_link_striping(thr)
            esh = F.build_enter_shard(n, P, rank)
            comm.stats(True)
            got[rank] = esh.enter_sharded(comm, x[rank * c:(rank + 1) * c].clone(), n)
            got[("nx", rank)] = comm.stats()["exchanges"]
            try:
                comm.set_link_striping(0)
                got[("late", rank)] = "accepted"
            except Exception as ex:
                got[("late", rank)] = str(ex)

        _thread_ranks(P, body)
        for r in range(P):
            assert torch.equal(got[r], want[r * c:(r + 1) * c]), (name, r)
            assert got[("late", r)] != "accepted", "the striping threshold must be frozen after the first exchange"
        counts[name] = got[("nx", 0)]
    assert counts["default"] == counts["never"] == 8 and counts["always"] > 8, counts


@pytest.mark.gpu
def test_link_striping_threshold_mismatch_fails_every_rank():
    """'the same threshold on every rank' was an unchecked contract, and exchange_striped decides locally.  The ranks now
    compare it in the agreement that precedes the first call of a sharded shape: one rank with another value makes that call fail
    on EVERY rank (no hang, no differently striped exchanges), and a communicator whose ranks agree works."""
    import torch
    import ecfft_amd
    F = ecfft_amd.FIELDS["secp256k1"]
    n, P = 1 << 13, 4
    c = n // P
    a = np.random.default_rng(6).integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
    x = torch.from_numpy(a.view(np.int64)).cuda()
    want = F.build_fftree(n).enter(x)
    torch.cuda.synchronize()
    got = {}

    def body(rank, make_comm):
        comm = make_comm()
        if rank == 1:
            comm.set_link_striping(0)
        esh = F.build_enter_shard(n, P, rank)
        try:
            esh.enter_sharded(comm, x[rank * c:(rank + 1) * c].clone(), n)
            got[rank] = "ok"
        except ecfft_amd.fftree.EcfftError as ex:
            got[rank] = f"error: {ex}"
        comm2 = make_comm()
        comm2.set_link_striping(0)
        got[("second", rank)] = esh.enter_sharded(comm2, x[rank * c:(rank + 1) * c].clone(), n)

    _thread_ranks(P, body)
    for r in range(P):
        assert got[r].startswith("error"), (r, got[r])
        assert torch.equal(got[("second", r)], want[r * c:(r + 1) * c]), r


@pytest.mark.gpu
@pytest.mark.parametrize("op", ["extend", "enter", "exit"])
def test_local_failure_on_one_rank_fails_every_rank_instead_of_hanging(op, hooks_lib):
    """a sharded call whose local preparation fails on ONE rank (allocation failure; injected here with
    ecfft_test_fail_next_collective) must return an error on EVERY rank — the ranks vote before the first exchange — and must not
    leave the peers blocked in a receive.  The next call (nothing injected) succeeds on all ranks and is bit-exact."""
    import torch
    import ecfft_amd
    from ecfft_amd import fftree as FT
    F = ecfft_amd.FIELDS["secp256k1"]
    P, n = 2, 1 << 12
    c = n // P
    full = F.build_fftree(2 * n)
    rng = np.random.default_rng(11)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
    x = torch.from_numpy(a.view(np.int64)).cuda()
    want = {"extend": lambda: full.extend(x, ecfft_amd.Moiety.S1), "enter": lambda: full.enter(x), "exit": lambda: full.exit(x)}[op]()
    torch.cuda.synchronize()
    L, first, second = FT.lib(), {}, {}

    def call(ctx, comm, mine):
        if op == "extend":
            return ctx.extend_sharded(comm, mine, n, ecfft_amd.Moiety.S1)
        return ctx.enter_sharded(comm, mine, n) if op == "enter" else ctx.exit_sharded(comm, mine, n)

    def body(rank, make_comm):
        comm = make_comm()
        ctx = F.build_fftree(2 * n)                      # full contexts accept every sharded call
        mine = x[rank * c:(rank + 1) * c].clone()
        if rank == 1:
            assert L.ecfft_test_fail_next_collective(ctx._h) == 0
        try:
            call(ctx, comm, mine); first[rank] = "ok"
        except FT.EcfftError:
            first[rank] = "error"
        second[rank] = call(ctx, comm, mine)

    _thread_ranks(P, body)
    torch.cuda.synchronize()
    assert first == {0: "error", 1: "error"}, first       # rank 0 did nothing wrong and still backs out, together with rank 1
    for r in range(P):
        assert torch.equal(second[r], want[r * c:(r + 1) * c]), (op, r)


@pytest.mark.gpu
def test_failure_injected_after_a_shape_is_agreed_waits_for_the_next_new_shape(hooks_lib):
    """an agreed call shape never votes again (a vote is a symmetric all-rank exchange; a rank voting alone would pair its
    4-byte messages with its peers' data messages).  The failure hook armed on rank 1 AFTER the first successful call therefore
    leaves the second call of that shape untouched — bit-exact on both ranks — and hits the next NEW shape, where every rank votes:
    both ranks report the error there, and the call after it succeeds."""
    import torch
    import ecfft_amd
    from ecfft_amd import fftree as FT
    F = ecfft_amd.FIELDS["m31"]
    P, n = 2, 1 << 12
    c = n // P
    full = F.build_fftree(2 * n)
    x = torch.from_numpy(np.random.default_rng(12).integers(0, 2**31 - 1, n, dtype=np.uint32).view(np.int32)).cuda()
    want_ext, want_ent = full.extend(x, ecfft_amd.Moiety.S1), full.enter(x)
    torch.cuda.synchronize()
    L, log = FT.lib(), {0: [], 1: []}

    def body(rank, make_comm):
        comm = make_comm()
        ctx = F.build_fftree(2 * n)
        mine = x[rank * c:(rank + 1) * c].clone()
        log[rank].append(torch.equal(ctx.extend_sharded(comm, mine, n, ecfft_amd.Moiety.S1), want_ext[rank * c:(rank + 1) * c]))   # agrees the shape
        if rank == 1:
            assert L.ecfft_test_fail_next_collective(ctx._h) == 0
        log[rank].append(torch.equal(ctx.extend_sharded(comm, mine, n, ecfft_amd.Moiety.S1), want_ext[rank * c:(rank + 1) * c]))   # agreed: no vote, no failure
        try:
            ctx.enter_sharded(comm, mine, n); log[rank].append("ok")                                                              # new shape: votes, fails everywhere
        except FT.EcfftError:
            log[rank].append("error")
        log[rank].append(torch.equal(ctx.enter_sharded(comm, mine, n), want_ent[rank * c:(rank + 1) * c]))

    _thread_ranks(P, body)
    torch.cuda.synchronize()
    assert log == {0: [True, True, "error", True], 1: [True, True, "error", True]}, log


@pytest.mark.gpu
def test_collective_exit_shard_build_fails_on_every_rank_when_one_rank_fails(monkeypatch, hooks_lib):
    """the collective ecfft_build_exit_shard votes after its local part: with rank 1's local part failing, rank 0's build returns an
    error too instead of waiting for exchanges that never come"""
    import ecfft_amd
    from ecfft_amd import fftree as FT
    F = ecfft_amd.FIELDS["m31"]
    assert FT.lib().ecfft_test_fail_build_rank(1) == 0        # set through the ABI: no environment variable reaches the hook
    res = {}

    def body(rank, make_comm):
        comm = make_comm()
        try:
            res[rank] = "built" if F.build_exit_shard(1 << 12, comm) is not None else "none"
        except FT.EcfftError:
            res[rank] = "error"

    _thread_ranks(2, body)
    assert res == {0: "error", 1: "error"}, res
    assert FT.lib().ecfft_test_fail_build_rank(-1) == 0
    _thread_ranks(2, lambda rank, make_comm: res.__setitem__(rank, F.build_exit_shard(1 << 12, make_comm()) is not None))
    assert res == {0: True, 1: True}


@pytest.mark.gpu
def test_projection_transport_bills_the_modelled_time_per_remote_exchange(hooks_lib):
    """ecfft_comm_init_projection (measurement only, tools/split_project.py): rank 0 of a 4-rank job on its own.  The exchange count and
    the bytes are those of the real split ENTER; an injected delay shows up on the stream once per exchange with a remote peer; the
    self pieces move (the output is this rank's data pushed through the launches, of the right size — its values mean nothing)."""
    import time
    import torch
    import ecfft_amd
    from ecfft_amd import distributed as D
    F = ecfft_amd.FIELDS["secp256k1"]
    n, P = 1 << 14, 4
    c = n // P
    shard = F.build_enter_shard(n, P, 0)
    a = np.random.default_rng(8).integers(0, 2**64, size=(c, 4), dtype=np.uint64); a[:, 3] >>= np.uint64(1)
    mine = torch.from_numpy(a.view(np.int64)).cuda()

    def timed(delay_us):
        comm = D.Comm.projection(P, 0, 0, delay_us, 0.0)
        out = shard.enter_sharded(comm, mine, n)
        assert out.shape == mine.shape
        comm.stats(True)
        shard.enter_sharded(comm, mine, n)
        torch.cuda.synchronize()
        st = comm.stats()
        comm.stats(False)
        ts = []
        for _ in range(7):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            shard.enter_sharded(comm, mine, n)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        return sorted(ts)[len(ts) // 2], st

    t_zero, st0 = timed(0.0)
    t_slow, st1 = timed(400.0)
    assert st0["exchanges"] == st1["exchanges"] == 5 and st0["bytes_sent"] == st1["bytes_sent"] > 0      # 2 levels x 2 + the final re-blocking
    exposed = t_slow - t_zero
    assert 0.6 * 5 * 400e-6 < exposed < 1.6 * 5 * 400e-6, (t_zero, t_slow)
