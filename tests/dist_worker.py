"""Worker for tests/test_distributed.py: run with torch.distributed.run, gloo backend, CPU only.
Checks tests/split_model.py extend_sharded (block<->cyclic all-to-all orchestration) against the
oracle's FFTree::extend with a numpy local-stage backend built from the reference's own matrices
(decompose_matrices / recombine_matrices, src/fftree.rs:26-27, 83-97, 104-118)."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import oracle  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tests"))
import split_model as D  # noqa: E402  (the Python model of the split: test infrastructure)


class OracleOps:
    """local stages on numpy data with the reference's (un-normalised) 2x2 matrices; no W scaling needed"""

    def __init__(self, F, otree):
        self.F, self.t = F, otree

    def _mats(self, e, k, which):
        m = 2 * e
        tbl = self.t.table(which, m)                      # 4*m field elements, heap order, Mat2x2 row-major
        tbl = tbl.reshape((m, 4) + tbl.shape[1:])
        lo = (m // 2) >> k                                # BinaryTree::get_layer(k) of a len-m tree
        return tbl[lo:2 * lo]

    def _apply(self, M, a, b):
        F = self.F
        o0 = F.add(F.mul(M[:, 0], a), F.mul(M[:, 1], b))
        o1 = F.add(F.mul(M[:, 2], a), F.mul(M[:, 3], b))
        return o0, o1

    def _stage(self, x, e, k, moiety, recombine, h_local, idx_of):
        """x: local numpy shard; pairs (j, j+h_local) inside blocks of 2*h_local; idx_of(i_local) -> global pair index i"""
        n = x.shape[0]
        src_par = 1 - moiety
        par = moiety if recombine else src_par            # :87-90 / :108-111
        mats = self._mats(e, k, oracle.T_RECOMBINE if recombine else oracle.T_DECOMPOSE)
        j = np.arange(n // 2)
        i_loc = j % h_local
        lo = (j // h_local) * 2 * h_local + i_loc
        M = mats[2 * idx_of(i_loc) + par]
        o0, o1 = self._apply(M, x[lo], x[lo + h_local])
        x[lo] = o0
        x[lo + h_local] = o1

    # --- whole local transforms and pointwise table steps (multi-GPU ENTER / EXIT) ---
    def _np(self, t):
        return t.contiguous().numpy().view(self.F.dtype).reshape(self.F.shape(t.shape[0]))

    def _t(self, a):
        v = np.ascontiguousarray(a).view(np.int64 if self.F.limbs > 1 else np.int32).reshape(a.shape[0], -1)
        return torch.from_numpy(v.copy())

    def enter_local(self, x):
        return self._t(self.t.enter(self._np(x)))

    def exit_local(self, x):
        return self._t(self.t.exit(self._np(x)))

    def extend_local(self, x, moiety):
        return self._t(self.t.extend(self._np(x), moiety))

    def table_fma(self, x, y, m, which, t_off, t_stride, mode):
        F = self.F
        xn = self._np(x)
        T = self.t.table(which, m)[t_off + np.arange(xn.shape[0]) * t_stride]
        if mode == 0:
            r = F.mul(xn, T)
        elif mode == 1:
            r = F.add(F.mul(xn, T), self._np(y))
        elif mode == 2:
            r = F.sub(self._np(y), F.mul(xn, T))
        else:
            r = F.mul(F.sub(self._np(y), xn), T)
        return self._t(r)

    def top_cyclic(self, shard, e, moiety, log_p, rank, recombine):
        x = shard.numpy().view(self.F.dtype).reshape(self.F.shape(shard.shape[0]))
        P = 1 << log_p
        ks = range(log_p) if not recombine else reversed(range(log_p))
        for k in ks:
            h = e >> (k + 1)
            self._stage(x, e, k, moiety, recombine, h // P, lambda il: il * P + rank)

    def local_block(self, shard, e, moiety, log_p):
        x = shard.numpy().view(self.F.dtype).reshape(self.F.shape(shard.shape[0]))
        le = e.bit_length() - 1
        for k in range(log_p, le):
            self._stage(x, e, k, moiety, False, e >> (k + 1), lambda il: il)
        for k in reversed(range(log_p, le)):
            self._stage(x, e, k, moiety, True, e >> (k + 1), lambda il: il)


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    ok = True
    for field, e in (("m31", 64), ("secp256k1", 32), ("m31", 1024)):
        F = oracle.field(field)
        ot = F.build_fftree(2 * e)
        rng = np.random.default_rng(1234)                 # same data on every rank
        if field == "m31":
            x = rng.integers(0, 2**31 - 1, e, dtype=np.uint32)
        else:
            p = 2**256 - 2**32 - 977
            x = F.from_ints([int.from_bytes(rng.bytes(32), "little") % p for _ in range(e)])
        c = e // world
        for moiety in (oracle.S1, oracle.S0):
            expect = ot.extend(x, moiety)
            mine = x[rank * c:(rank + 1) * c].copy()
            t = torch.from_numpy(mine.view(np.int64 if field == "secp256k1" else np.int32).reshape(c, -1).copy())
            out = D.extend_sharded(OracleOps(F, ot), t, e, moiety)
            got = out.numpy().view(F.dtype).reshape(F.shape(c))
            good = np.array_equal(got, expect[rank * c:(rank + 1) * c])
            ok = ok and good
            # the transposes alone are inverse permutations
            back = D.cyclic_to_block(D.block_to_cyclic(t.clone(), world), world)
            ok = ok and torch.equal(back, t)
            cyc = D.block_to_cyclic(t.clone(), world).numpy().view(F.dtype).reshape(F.shape(c))
            ok = ok and np.array_equal(cyc, x[rank::world])
    # ---- one ENTER / EXIT of n coefficients split over the ranks ----
    groups = D.make_groups()
    for field, n in (("m31", 256), ("secp256k1", 128), ("m31", 2048)):
        F = oracle.field(field)
        ot = F.build_fftree(n)
        rng = np.random.default_rng(99)
        if field == "m31":
            x = rng.integers(0, 2**31 - 1, n, dtype=np.uint32)
        else:
            p = 2**256 - 2**32 - 977
            x = F.from_ints([int.from_bytes(rng.bytes(32), "little") % p for _ in range(n)])
        c = n // world
        if c < 4 * world:
            continue
        ops = OracleOps(F, ot)
        want = ot.enter(x)
        mine = ops._t(x[rank * c:(rank + 1) * c])
        ev = D.enter_sharded(ops, mine, n, groups)
        ok = ok and np.array_equal(ops._np(ev), want[rank * c:(rank + 1) * c])
        back = D.exit_sharded(ops, ev, n, groups)
        ok = ok and np.array_equal(ops._np(back), x[rank * c:(rank + 1) * c])
        # EXIT of arbitrary evaluations against the oracle
        r = ot.exit(x)
        got = D.exit_sharded(ops, mine, n, groups)
        ok = ok and np.array_equal(ops._np(got), r[rank * c:(rank + 1) * c])
        # round 4: the same with every level split (the form before), and the gather form of full contexts (one all-gather, then
        # every top level redundantly on the block that contains the rank's chunk)
        got = D.exit_sharded(ops, mine, n, groups, pair_local=False)
        ok = ok and np.array_equal(ops._np(got), r[rank * c:(rank + 1) * c])
        got = D.exit_sharded_gather(ops, mine, n)
        ok = ok and np.array_equal(ops._np(got), r[rank * c:(rank + 1) * c])
    flag = torch.tensor([1 if ok else 0])
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    if rank == 0:
        print("DIST_OK" if int(flag.item()) == 1 else "DIST_FAIL")
    dist.destroy_process_group()
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
