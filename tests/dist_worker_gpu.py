"""Worker for the GPU leg of tests/test_distributed.py: torch.distributed.run, gloo backend, every rank on cuda:0.
Runs the multi-GPU code path with the REAL local ops (HIP kernels through the C-ABI on device tensors) and compares with
the single-GPU transforms of the same context: split EXTEND, sharded ENTER, sharded EXIT."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import ecfft_amd  # noqa: E402
from ecfft_amd import distributed as D  # noqa: E402


def synth(field, n, seed):
    rng = np.random.default_rng(seed)
    if field == "m31":
        return rng.integers(0, 2**31 - 1, n, dtype=np.uint32)
    a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64)
    a[:, 3] >>= np.uint64(1)
    return a


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    torch.cuda.set_device(0)
    ok = True
    groups = D.make_groups()
    for field, n in (("secp256k1", 1 << 13), ("m31", 1 << 16)):
        F = ecfft_amd.FIELDS[field]
        tree = F.build_fftree(2 * n)
        ops = D.HipOps(tree)
        x = synth(field, n, 11)                                            # same on every rank
        view = np.int64 if field == "secp256k1" else np.int32
        c = n // world
        mine = torch.from_numpy(x[rank * c:(rank + 1) * c].view(view).reshape(c, -1).copy()).cuda()
        full = torch.from_numpy(x.view(view).reshape(n, -1).copy()).cuda()
        # split EXTEND == single-GPU EXTEND
        for moiety in (ecfft_amd.Moiety.S1, ecfft_amd.Moiety.S0):
            want = tree.extend(full, moiety)
            got = D.extend_sharded(ops, mine.clone(), n, moiety)
            ok = ok and torch.equal(got, want[rank * c:(rank + 1) * c])
        # sharded ENTER / EXIT == single-GPU ENTER / EXIT
        want = tree.enter(full)
        got = D.enter_sharded(ops, mine.clone(), n, groups)
        ok = ok and torch.equal(got, want[rank * c:(rank + 1) * c])
        want = tree.exit(full)
        got = D.exit_sharded(ops, mine.clone(), n, groups)
        ok = ok and torch.equal(got, want[rank * c:(rank + 1) * c])
        torch.cuda.synchronize()
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST_GPU_OK" if int(flag.item()) == 1 else "DIST_GPU_FAIL")
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
