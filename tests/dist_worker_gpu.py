"""Worker for the GPU leg of tests/test_distributed.py: torch.distributed.run, gloo backend, every rank on cuda:0.
Runs the C++ multi-GPU path of the C ABI (ecfft_extend_sharded / ecfft_enter_sharded / ecfft_exit_sharded: index maps,
pack / unpack operators, cyclic-shard stage kernels, block-local fused passes) with a CALLBACK transport — the exchanges are
torch.distributed point-to-point calls staged through host memory, because RCCL refuses several ranks on one device — and
compares every rank's shard with the single-GPU transforms of the same context, bit for bit.  The Python model of the same
algorithm (tests/split_model.py extend_sharded with HipOps) is run next to it for the split EXTEND."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ecfft_amd  # noqa: E402
from ecfft_amd import distributed as D  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import split_model as M  # noqa: E402  (the Python model of the split: test infrastructure)


def synth(field, n, seed):
    rng = np.random.default_rng(seed)
    if field == "m31":
        return rng.integers(0, 2**31 - 1, n, dtype=np.uint32)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] >>= np.uint64(1)
    return a


def main():
    # ECFFT_WORKER_RCCL=1 (multi-GPU hosts): one rank per GPU, exchanges over the real RCCL transport (grouped ncclSend / ncclRecv)
    # ECFFT_WORKER_RCCL=stub (one-GPU hosts): every rank on cuda:0, gloo for the process group, and the RcclTransport bound to the
    # test-only stand-in library (ECFFT_WORKER_RCCL_LIB = tests/stub_rccl/librccl_stub.so, handed to the library through
    # ecfft_comm_set_rccl_library — the library itself reads no environment variable): the RCCL code path with world > 1
    stub = os.environ.get("ECFFT_WORKER_RCCL") == "stub"
    rccl = os.environ.get("ECFFT_WORKER_RCCL") == "1"
    if stub:
        assert os.environ.get("ECFFT_WORKER_RCCL_LIB"), "stub mode needs ECFFT_WORKER_RCCL_LIB"
        D.Comm.set_rccl_library(os.environ["ECFFT_WORKER_RCCL_LIB"])
        dev = 0
        dist.init_process_group("gloo")
        torch.cuda.set_device(0)
    elif rccl:
        dev = int(os.environ.get("LOCAL_RANK", "0"))
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
    else:
        dev = 0
        dist.init_process_group("gloo")
        torch.cuda.set_device(0)
    rank, world = dist.get_rank(), dist.get_world_size()
    ok = True
    comm = D.Comm.rccl(device=dev) if (rccl or stub) else D.Comm.callback()
    assert comm.rank == rank and comm.world == world
    for field, n in (("secp256k1", 1 << 13), ("m31", 1 << 16)):
        F = ecfft_amd.FIELDS[field]
        tree = F.build_fftree(2 * n, device=dev)
        x = synth(field, n, 11)                                            # same on every rank
        view = np.int64 if field == "secp256k1" else np.int32
        c = n // world
        mine = torch.from_numpy(x[rank * c:(rank + 1) * c].view(view).reshape(c, -1).copy()).cuda()
        full = torch.from_numpy(x.view(view).reshape(n, -1).copy()).cuda()

        def check(what, got, want):
            good = torch.equal(got, want[rank * c:(rank + 1) * c])
            if not good:
                print(f"rank {rank}: {field} {what} MISMATCH", flush=True)
            return good
        # split EXTEND == single-GPU EXTEND (C++ path and the Python model)
        for moiety in (ecfft_amd.Moiety.S1, ecfft_amd.Moiety.S0):
            want = tree.extend(full, moiety)
            ok = check(f"extend_sharded {moiety}", tree.extend_sharded(comm, mine.clone(), n, moiety), want) and ok
            ok = check(f"model extend {moiety}", M.extend_sharded(M.HipOps(tree), mine.clone(), n, moiety), want) and ok
        # cyclic-in / cyclic-out variants on the full context (one exchange fewer each)
        cyc = full[rank::world].contiguous()
        for moiety in (ecfft_amd.Moiety.S1, ecfft_amd.Moiety.S0):
            want = tree.extend(full, moiety)
            ok = check(f"extend cyclic-in {moiety}", tree.extend_sharded(comm, cyc, n, moiety, cyclic_in=True), want) and ok
            got = tree.extend_sharded(comm, mine.clone(), n, moiety, cyclic_out=True)
            if not torch.equal(got, want[rank::world]):
                print(f"rank {rank}: {field} extend cyclic-out {moiety} MISMATCH", flush=True); ok = False
            got = tree.extend_sharded(comm, cyc, n, moiety, cyclic_in=True, cyclic_out=True)
            if not torch.equal(got, want[rank::world]):
                print(f"rank {rank}: {field} extend cyclic-in-out {moiety} MISMATCH", flush=True); ok = False
        # sharded EXTEND-only context: this rank's share of the tables only (ecfft_build_extend_shard), same results
        shard = F.build_extend_shard(n, world, rank, device=dev)
        for moiety in (ecfft_amd.Moiety.S1, ecfft_amd.Moiety.S0):
            ok = check(f"shard-context extend {moiety}", shard.extend_sharded(comm, mine.clone(), n, moiety), tree.extend(full, moiety)) and ok
        rc = ecfft_amd.lib().ecfft_enter(shard._h, mine.data_ptr(), mine.data_ptr(), c, 1, None)      # anything else is refused
        ok = (rc == ecfft_amd.fftree.ERR_BAD_ARG) and ok
        del shard
        # sharded ENTER-only / EXIT-only contexts (the EXIT one is a collective build over the communicator)
        if world > 1:
            esh = F.build_enter_shard(n, world, rank, device=dev)
            ok = check("shard-context enter", esh.enter_sharded(comm, mine.clone(), n), tree.enter(full)) and ok
            del esh
            xsh = F.build_exit_shard(n, comm, device=dev)
            ok = check("shard-context exit", xsh.exit_sharded(comm, mine.clone(), n), tree.exit(full)) and ok
            del xsh
        # in place (in == out is allowed by the ABI): run through the raw call
        buf = mine.clone()
        ecfft_amd.fftree._check(ecfft_amd.lib().ecfft_extend_sharded(tree._h, comm._h, buf.data_ptr(), buf.data_ptr(), n, 1,
                                                                      torch.cuda.current_stream().cuda_stream))
        ok = check("extend_sharded in place", buf, tree.extend(full, ecfft_amd.Moiety.S1)) and ok
        # sharded ENTER / EXIT == single-GPU ENTER / EXIT (EXIT of arbitrary evaluations, not only of ENTER outputs)
        ok = check("enter_sharded", tree.enter_sharded(comm, mine.clone(), n), tree.enter(full)) and ok
        ok = check("exit_sharded", tree.exit_sharded(comm, mine.clone(), n), tree.exit(full)) and ok
        torch.cuda.synchronize()
    st = comm.stats()
    ok = ok and st["exchanges"] > 0
    flag = torch.tensor([1 if ok else 0], device="cuda" if rccl else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST_GPU_OK" if int(flag.item()) == 1 else "DIST_GPU_FAIL")
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
