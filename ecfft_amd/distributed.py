"""One EXTEND split across the P GPUs of a node (BASELINE.json configs[3]).

Block-distributed I/O: rank r holds global positions [r*e/P, (r+1)*e/P) of the length-e vector.
Butterfly stage k pairs (i, i + e >> (k+1)):
  * in the BLOCK distribution every stage k >= log2 P is local,
  * in the CYCLIC distribution (position j on rank j mod P) every stage k < log2(e) - log2(P) is local,
so the transform is
    block --all-to-all--> cyclic : 1/W scaling + decompose stages 0..logP-1      (table stride P, offset rank)
          --all-to-all--> block  : all remaining stages, decompose then recombine  (the fused single-GPU kernels)
          --all-to-all--> cyclic : recombine stages logP-1..0 + W scaling
          --all-to-all--> block.
Each all-to-all (`torch.distributed.all_to_all_single`; backend "nccl" = RCCL over xGMI on the GPU box, "gloo"
in the CPU tests) sends one equal message per peer — one per point-to-point xGMI link.

The local compute is delegated to `ops`, an object with
    ops.top_cyclic(shard, e, moiety, log_p, rank, recombine)   (in place)
    ops.local_block(shard, e, moiety, log_p)                   (in place)
In production `ops` is an `ecfft_amd.FFTree` (HIP kernels through the C ABI); the CPU tests plug in a numpy
implementation built from the reference's own matrices so that the data movement is tested without a GPU.
"""
import torch
import torch.distributed as dist


class HipOps:
    """local stages through the C-ABI (device tensors, current stream)"""

    def __init__(self, tree):
        self.tree = tree

    def top_cyclic(self, shard, e, moiety, log_p, rank, recombine):
        self.tree.extend_top_cyclic(shard, e, moiety, log_p, rank, recombine)

    def local_block(self, shard, e, moiety, log_p):
        self.tree.extend_local_block(shard, e, moiety, log_p)


def _rows(t, n_elems):
    """view a shard as [n_elems, limbs] whatever the limb count"""
    return t.reshape(n_elems, -1)


def block_to_cyclic(x, world, group=None):
    """x: block shard [c, limbs] -> cyclic shard [c, limbs] (local j' <-> global j'*P + rank)."""
    c = x.shape[0]
    send = _rows(x, c).reshape(c // world, world, -1).transpose(0, 1).contiguous()   # [P, c/P, limbs]: row q = x[q::P]
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
    return recv.reshape(c, -1)                                                       # source-rank major = ascending j'


def cyclic_to_block(y, world, group=None):
    """inverse of block_to_cyclic"""
    c = y.shape[0]
    send = _rows(y, c).contiguous()                                                  # chunk r = y[r*c/P:(r+1)*c/P] -> rank r
    recv = torch.empty_like(send)
    dist.all_to_all_single(recv.view(-1), send.view(-1), group=group)
    return recv.reshape(world, c // world, -1).transpose(0, 1).contiguous().reshape(c, -1)


def extend_sharded(ops, x_block, e, moiety, group=None):
    """FFTree::extend (src/fftree.rs:123-126) of ONE length-e vector held block-distributed over the
    process group.  x_block: this rank's e/P elements as a [e/P, limbs] (or [e/P]) tensor.  Returns the
    rank's block shard of the result."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    assert world & (world - 1) == 0, "power-of-two number of ranks"
    log_p = world.bit_length() - 1
    shape = x_block.shape
    c = shape[0]
    assert c * world == e and c >= 2 * world, "need e/P elements per rank and at least 2P of them"
    x = x_block.reshape(c, -1)
    if world == 1:
        raise ValueError("use FFTree.extend on a single rank")
    y = block_to_cyclic(x, world, group)
    ops.top_cyclic(y, e, moiety, log_p, rank, False)
    z = cyclic_to_block(y, world, group)
    ops.local_block(z, e, moiety, log_p)
    y = block_to_cyclic(z, world, group)
    ops.top_cyclic(y, e, moiety, log_p, rank, True)
    out = cyclic_to_block(y, world, group)
    return out.reshape(shape)
