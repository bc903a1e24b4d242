"""TEST INFRASTRUCTURE: a pure-Python MODEL of ONE transform split across the ranks of a process group (the algorithm the product
runs below the C ABI: ecfft_amd/csrc/device_tree.h extend_split / api_enter_split / api_exit_split), with pluggable local ops.  The
CPU test-suite runs it over gloo with numpy / oracle local stages (tests/dist_worker.py): the index maps of the split are tested
without a GPU; on the GPU box the C++ path is tested against the single-GPU transforms (tests/dist_worker_gpu.py) and `HipOps`
lets the same model run on the C-ABI building blocks.  Moved out of the product package in round 6 (ecfft_amd/distributed.py keeps
the communicator set-up only).

Loops being split: /root/reference/src/fftree.rs:83-118 inside :143-161, 200-259.

Index maps.  Block-distributed I/O: rank r holds global positions [r*e/P, (r+1)*e/P) of the length-e vector.
Butterfly stage k pairs (i, i + e >> (k+1)):
  * in the BLOCK distribution every stage k >= log2 P is local,
  * in the CYCLIC distribution (position j on rank j mod P) every stage k < log2(e) - log2(P) is local,
so the transform is
    block --all-to-all--> cyclic : 1/W scaling + decompose stages 0..logP-1      (table stride P, offset rank)
          --all-to-all--> block  : all remaining stages, decompose then recombine  (the fused single-GPU kernels)
          --all-to-all--> cyclic : recombine stages logP-1..0 + W scaling
          --all-to-all--> block.
Each all-to-all sends one equal message per peer — one per point-to-point xGMI link.

The model delegates local compute to `ops`, an object with
    ops.enter_local(x) / ops.exit_local(x) / ops.extend_local(x, moiety)
    ops.table_fma(x, y, m, which, t_off, t_stride, mode)
    ops.top_cyclic(shard, e, moiety, log_p, rank, recombine)   (in place)
    ops.local_block(shard, e, moiety, log_p)                   (in place)
(`HipOps`: the C-ABI building blocks on device tensors; the CPU tests plug in a numpy implementation built from the
reference's own matrices).
"""
import torch
import torch.distributed as dist


TBL_XNN_S, TBL_XNN_S_INV, TBL_Z0_INV_S1, TBL_Z0Z0 = 3, 4, 7, 9      # ECFFT_TBL_* ids (include/ecfft_hip.h)
S0, S1 = 0, 1


class HipOps:
    """local work through the C-ABI (device tensors, current stream)"""

    def __init__(self, tree):
        self.tree = tree

    def enter_local(self, x):
        return self.tree.enter(x)

    def exit_local(self, x):
        return self.tree.exit(x)

    def extend_local(self, x, moiety):
        return self.tree.extend(x, moiety)

    def table_fma(self, x, y, m, which, t_off, t_stride, mode):
        return self.tree.table_fma(x, y, m, which, t_off, t_stride, mode)

    def top_cyclic(self, shard, e, moiety, log_p, rank, recombine):
        self.tree.extend_top_cyclic(shard, e, moiety, log_p, rank, recombine)

    def local_block(self, shard, e, moiety, log_p):
        self.tree.extend_local_block(shard, e, moiety, log_p)


def _a2a(recv, send, group=None, **kw):
    """all_to_all_single; with the gloo backend (functional tests: several ranks sharing one GPU) device tensors are
    staged through host memory, with nccl (= RCCL over xGMI) they go GPU to GPU."""
    if send.is_cuda and dist.get_backend(group) == "gloo":
        r = torch.empty(recv.shape, dtype=recv.dtype)
        dist.all_to_all_single(r, send.cpu(), group=group, **kw)
        recv.copy_(r)
    else:
        dist.all_to_all_single(recv, send, group=group, **kw)


def _rows(t, n_elems):
    """view a shard as [n_elems, limbs] whatever the limb count"""
    return t.reshape(n_elems, -1)


def block_to_cyclic(x, world, group=None):
    """x: block shard [c, limbs] -> cyclic shard [c, limbs] (local j' <-> global j'*P + rank)."""
    c = x.shape[0]
    send = _rows(x, c).reshape(c // world, world, -1).transpose(0, 1).contiguous()   # [P, c/P, limbs]: row q = x[q::P]
    recv = torch.empty_like(send)
    _a2a(recv.view(-1), send.view(-1), group)
    return recv.reshape(c, -1)                                                       # source-rank major = ascending j'


def cyclic_to_block(y, world, group=None):
    """inverse of block_to_cyclic"""
    c = y.shape[0]
    send = _rows(y, c).contiguous()                                                  # chunk r = y[r*c/P:(r+1)*c/P] -> rank r
    recv = torch.empty_like(send)
    _a2a(recv.view(-1), send.view(-1), group)
    return recv.reshape(world, c // world, -1).transpose(0, 1).contiguous().reshape(c, -1)


def extend_sharded(ops, x_block, e, moiety, group=None, cyclic_in=False, cyclic_out=False):
    """FFTree::extend (src/fftree.rs:123-126) of ONE length-e vector held block-distributed over the
    process group.  x_block: this rank's e/P elements as a [e/P, limbs] (or [e/P]) tensor.  Returns the
    rank's block shard of the result.  cyclic_in / cyclic_out: the shard on that side is the cyclic one (local j' = global
    j'*P + rank) and that side's all-to-all disappears (ecfft_extend_sharded_layout)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert world & (world - 1) == 0, "power-of-two number of ranks"
    log_p = world.bit_length() - 1
    shape = x_block.shape
    c = shape[0]
    assert c * world == e and c >= 2 * world, "need e/P elements per rank and at least 2P of them"
    x = x_block.reshape(c, -1)
    y = x.clone() if cyclic_in else block_to_cyclic(x, world, group)   # world == 1: a self all-to-all, log_p = 0 (no cyclic stage)
    ops.top_cyclic(y, e, moiety, log_p, rank, False)
    z = cyclic_to_block(y, world, group)
    ops.local_block(z, e, moiety, log_p)
    y = block_to_cyclic(z, world, group)
    ops.top_cyclic(y, e, moiety, log_p, rank, True)
    out = y if cyclic_out else cyclic_to_block(y, world, group)
    return out.reshape(shape)


# ------------------------------------------------------------------------------------------------------------------
# ONE ENTER / EXIT of n coefficients split over the P ranks.
# Block distribution: rank r holds positions [r*c, (r+1)*c), c = n/P, of the coefficient / evaluation vector.
#   * levels with block size m <= c touch one rank only: they are the local ENTER / EXIT of the rank's chunk;
#   * level m = c*2^j (j = 1..log2 P) works inside groups of Q = 2^j consecutive ranks with every vector CYCLIC over the
#     (half-)group (entry j of rank a = position j*Q + a): its EXTENDs are cyclic-in / cyclic-out split-EXTENDs, its pointwise
#     steps are `table_fma` calls at those strided positions, and one all_to_all_single with split sizes per level moves the
#     data to the next level's order; one more all-to-all at the user's block boundary.  The C++ path (device_tree.h
#     api_enter_split / api_exit_split) is this algorithm; the CPU tests run it on the oracle's local operators.
# ------------------------------------------------------------------------------------------------------------------
def make_groups():
    """every group of 2^j consecutive ranks, j >= 1 (collective: all ranks call it once, in the same order)"""
    world, rank = dist.get_world_size(), dist.get_rank()
    groups = {}
    size = 2
    while size <= world:
        for idx in range(world // size):
            g = dist.new_group(list(range(idx * size, (idx + 1) * size)))
            if idx == rank // size:
                groups[size] = g
        size *= 2
    return groups


def _extend(ops, x, e, moiety, group, cyclic=False):
    if group is None or dist.get_world_size(group) == 1:
        return ops.extend_local(x, moiety)                     # one rank: cyclic order == natural order
    return extend_sharded(ops, x, e, moiety, group, cyclic_in=cyclic, cyclic_out=cyclic)


def _a2a_split(pieces, dests, srcs, piece_len, Q, like, group):
    """send pieces[k] (1-D, equal length piece_len words) to group rank dests[k] (ascending); receive one piece from each of srcs"""
    in_split = [0] * Q
    for d in dests:
        in_split[d] += piece_len
    out_split = [0] * Q
    for s_ in srcs:
        out_split[s_] += piece_len
    send = torch.cat(pieces).contiguous()
    recv = torch.empty(piece_len * len(srcs), dtype=like.dtype, device=like.device)
    _a2a(recv, send, group, output_split_sizes=out_split, input_split_sizes=in_split)
    return [recv[k * piece_len:(k + 1) * piece_len] for k in range(len(srcs))]


def enter_sharded(ops, x_block, n, groups):
    """FFTree::enter (src/fftree.rs:164-167) of n coefficients held block-distributed; returns this rank's block of
    the evaluations (leaf order).  The MODEL of DeviceChain::api_enter_split: every vector of a top level stays CYCLIC over
    its (half-)group — position i = i'*Q + a of the level's result on rank a — so the split EXTEND runs cyclic-in / cyclic-out
    and one exchange per level hands rank a the whole `cur` (a even: its outputs are the even positions, u0 + xnn*v0) or `ext`
    (a odd: u1 + xnn*v1) share of sub-rank a/2 of both half-groups."""
    world, rank = dist.get_world_size(), dist.get_rank()
    c = x_block.shape[0]
    assert c * world == n
    shape = x_block.shape
    cur = ops.enter_local(x_block.reshape(c, -1))
    limbs = cur.shape[1]
    Q = 2
    while Q <= world:
        half = Q // 2
        base = (rank // Q) * Q
        a = rank - base
        m, e = c * Q, c * Q // 2
        ext = _extend(ops, cur, e, S1, groups.get(half) if half > 1 else None, cyclic=True)   # u1 (lower half-group) or v1 (upper)
        ap = a % half
        got = _a2a_split([cur.reshape(-1), ext.reshape(-1)], [2 * ap, 2 * ap + 1], [a // 2, half + a // 2], c * limbs, Q, cur, groups[Q])
        U, V = got[0].reshape(c, limbs), got[1].reshape(c, limbs)
        cur = ops.table_fma(V.contiguous(), U.contiguous(), m, TBL_XNN_S, a, Q, 1).reshape(c, limbs)   # out[i'*Q + a] = U + xnn_s * V  (:157-158)
        Q *= 2
    out = cyclic_to_block(cur, world) if world > 1 else cur                                    # cyclic over all ranks -> the user's block
    return out.reshape(shape)


def exit_level_block(ops, blk, m):
    """ONE level of FFTree::exit (src/fftree.rs:206-224) on a whole block of m evaluations in natural order held by ONE rank:
    returns [u0 | v0] (m/2 each).  The redundant top levels of the split EXIT (DeviceChain::api_exit_split, round 4) run this —
    there as the single-GPU fused passes, here with the same table steps as the split level at Q = 1."""
    limbs = blk.shape[1]
    e = m // 2
    pairs = blk.reshape(e, 2 * limbs)
    e0, e1 = pairs[:, :limbs].contiguous(), pairs[:, limbs:].contiguous()

    def redc(x0, x1):
        t0 = ops.table_fma(x0, None, m, TBL_XNN_S_INV, 0, 2, 0)
        g1 = ops.extend_local(t0, S1)
        h1 = ops.table_fma(ops.table_fma(g1, x1, m, TBL_XNN_S, 1, 2, 2), None, m, TBL_Z0_INV_S1, 0, 1, 0)
        return ops.extend_local(h1, S0), h1
    h0, h1 = redc(e0, e1)
    u0, _ = redc(ops.table_fma(h0, None, m, TBL_Z0Z0, 0, 2, 0), ops.table_fma(h1, None, m, TBL_Z0Z0, 1, 2, 0))
    v0 = ops.table_fma(u0, e0, m, TBL_XNN_S_INV, 0, 2, 3)
    return torch.cat([u0, v0]).contiguous()


def exit_sharded_gather(ops, y_block, n):
    """FFTree::exit of n evaluations held block-distributed, the form a FULL context uses for n <= 2^21 (api_exit_split, round 4):
    ONE all-gather, then every rank walks its own path down the tree — level Q on the block of Q c evaluations that contains its
    chunk, keep the half that contains the chunk — and finishes with the local EXIT of its chunk.  No exchange after the first."""
    world, rank = dist.get_world_size(), dist.get_rank()
    c = y_block.shape[0]
    assert c * world == n
    shape = y_block.shape
    mine = y_block.reshape(c, -1).contiguous()
    parts = [torch.empty_like(mine) for _ in range(world)]
    if world > 1:
        dist.all_gather(parts, mine)
    else:
        parts = [mine]
    blk = torch.cat(parts).contiguous()                        # the block of Q c evaluations that contains the chunk, Q = world
    Q = world
    while Q >= 2:
        m = c * Q
        out = exit_level_block(ops, blk, m)                    # [u0 | v0]
        half = (rank // (Q // 2)) & 1
        blk = out[half * (m // 2):(half + 1) * (m // 2)].contiguous()
        Q //= 2
    return ops.exit_local(blk).reshape(shape)


def exit_sharded(ops, y_block, n, groups, pair_local=True):
    """FFTree::exit (src/fftree.rs:227-230) of n evaluations held block-distributed; returns this rank's block of the
    coefficients.  The MODEL of DeviceChain::api_exit_split (pair_local: the pair level redundantly, as the C++ does since round 4): one all-to-all turns the block into (e0, e1) cyclic over all
    ranks; inside a level every length-m/2 vector is cyclic over the group of Q ranks (entry j = position j*Q + a), the
    tables are read at those positions, the four EXTENDs run cyclic-in / cyclic-out, and one exchange re-distributes
    (u0 | v0): rank a's whole u0 share is the even (a even) or odd (a odd) half of what sub-rank a/2 of the lower half-group
    needs next, its v0 share the same for the upper half-group."""
    world, rank = dist.get_world_size(), dist.get_rank()
    c = y_block.shape[0]
    assert c * world == n
    shape = y_block.shape
    cur = y_block.reshape(c, -1).contiguous()
    limbs = cur.shape[1]
    hc = c // 2
    pairs = cur.reshape(hc, 2 * limbs)                         # row t = (even, odd) entry of pair t of the chunk
    if world > 1:
        pairs = block_to_cyclic(pairs, world)                 # row j' = pair j'*P + rank of the whole vector
    e0, e1 = pairs[:, :limbs].contiguous(), pairs[:, limbs:].contiguous()
    Q = world
    while Q >= 2:
        half = Q // 2
        base = (rank // Q) * Q
        a = rank - base
        m, e = c * Q, c * Q // 2
        G = groups[Q]
        if Q == 2 and pair_local:
            # round 4: the level of the PAIRS runs redundantly on both ranks: one exchange hands each rank its partner's (e0, e1)
            # share (positions i = 2j + partner), the level is the whole-block level, each rank keeps its half of [u0 | v0]
            mine2 = torch.cat([e0, e1], dim=1).contiguous()                       # row j = (e0, e1) at i = 2j + a
            both = [torch.empty_like(mine2) for _ in range(2)]
            dist.all_gather(both, mine2, group=G)
            blk = torch.stack([both[0], both[1]], dim=1).reshape(2 * c, limbs).contiguous()   # row 2i + par, i = 2j + a'
            out = exit_level_block(ops, blk, m)
            cur = out[a * c:(a + 1) * c].contiguous()
            return ops.exit_local(cur).reshape(shape)

        def redc(x0, x1):                                      # redc_impl with a = xnn_s, moiety S0 (:232-259), at positions j*Q + a
            t0 = ops.table_fma(x0, None, m, TBL_XNN_S_INV, 2 * a, 2 * Q, 0)
            g1 = _extend(ops, t0, e, S1, G, cyclic=True)
            h1 = ops.table_fma(ops.table_fma(g1, x1, m, TBL_XNN_S, 2 * a + 1, 2 * Q, 2), None, m, TBL_Z0_INV_S1, a, Q, 0)
            h0 = _extend(ops, h1, e, S0, G, cyclic=True)
            return h0, h1
        h0, h1 = redc(e0, e1)                                  # modular_reduce_impl (:277-281)
        hc0 = ops.table_fma(h0, None, m, TBL_Z0Z0, 2 * a, 2 * Q, 0)
        hc1 = ops.table_fma(h1, None, m, TBL_Z0Z0, 2 * a + 1, 2 * Q, 0)
        u0, _ = redc(hc0, hc1)
        v0 = ops.table_fma(u0, e0, m, TBL_XNN_S_INV, 2 * a, 2 * Q, 3)                 # (e0 - u0) * xinv  (:217-219)
        ap = a % half
        got = _a2a_split([u0.reshape(-1), v0.reshape(-1)], [a // 2, half + a // 2], [2 * ap, 2 * ap + 1], hc * limbs, Q, cur, G)
        e0, e1 = got[0].reshape(hc, limbs).contiguous(), got[1].reshape(hc, limbs).contiguous()
        Q //= 2
    cur = torch.stack([e0, e1], dim=1).reshape(c, limbs).contiguous()
    out = ops.exit_local(cur)
    return out.reshape(shape)
