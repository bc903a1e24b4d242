"""Communicator set-up for ONE transform split across the GPUs of a node (BASELINE.json configs[3]).

The orchestration of a split transform lives BELOW the C ABI (ecfft_extend_sharded / ecfft_enter_sharded / ecfft_exit_sharded,
ecfft_amd/csrc/device_tree.h: extend_split, api_enter_split, api_exit_split) and moves data with grouped ncclSend / ncclRecv on
librccl directly.  This module only creates the `ecfft_comm` those calls take: `Comm.rccl()` (rank 0 makes the RCCL unique id,
torch.distributed hands it round) or `Comm.callback()` (the exchanges are done by torch.distributed point-to-point calls staged
through host memory: lets several ranks share one GPU in the tests, where RCCL refuses duplicate devices).

(The pure-Python MODEL of the same split that the CPU test-suite runs over gloo is test infrastructure: tests/split_model.py.)
"""
import ctypes

import torch
import torch.distributed as dist



class Comm:
    """`ecfft_comm` of the C ABI: the inter-GPU transport of the sharded transforms (one process per GPU)."""

    def __init__(self, handle, keep=None):
        from . import fftree
        self._h, self._keep = handle, keep
        self._L = fftree.lib()      # the library that made the handle serves every later call on it, __del__ included

    def __del__(self):
        try:
            if self._h:
                self._L.ecfft_comm_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def rank(self):
        return self._L.ecfft_comm_rank(self._h)

    @property
    def world(self):
        return self._L.ecfft_comm_world(self._h)

    @staticmethod
    def rccl(device=None, world=None, rank=None):
        """RCCL communicator over all ranks of the default process group (or a single-rank one when torch.distributed is
        not initialised): rank 0 creates the unique id, torch.distributed broadcasts it, every rank calls ncclCommInitRank."""
        from . import fftree
        L = fftree.lib()
        if world is None:
            world = dist.get_world_size() if dist.is_initialized() else 1
            rank = dist.get_rank() if dist.is_initialized() else 0
        device = torch.cuda.current_device() if device is None else device
        buf = ctypes.create_string_buffer(128)
        if rank == 0:
            fftree._check(L.ecfft_comm_get_unique_id(buf))
        if world > 1:
            box = [bytes(buf.raw)]
            dist.broadcast_object_list(box, src=0)
            buf = ctypes.create_string_buffer(box[0], 128)
        h = ctypes.c_void_p()
        fftree._check(L.ecfft_comm_init_rank(buf, world, rank, device, ctypes.byref(h)))
        return Comm(h)

    @staticmethod
    def callback(device=None, world=None, rank=None, exchange=None):
        """communicator whose exchanges run over torch.distributed point-to-point calls staged through host memory (any
        backend; the tests use gloo with several ranks on one GPU).  `exchange` (with `world` and `rank`) replaces that
        transport by a caller-supplied function of the ecfft_exchange_fn signature (include/ecfft_hip.h)."""
        from . import fftree
        import numpy as np
        L = fftree.lib()
        device = torch.cuda.current_device() if device is None else device
        if exchange is not None:
            cb = fftree.EXCHANGE_FN(exchange)
            h = ctypes.c_void_p()
            fftree._check(L.ecfft_comm_init_callback(world, rank, device, cb, None, ctypes.byref(h)))
            return Comm(h, keep=cb)
        world, rank = dist.get_world_size(), dist.get_rank()

        def exchange(user, ns, sp, sptr, sb, nr, rp, rptr, rb, stream):
            try:
                torch.cuda.synchronize()
                sends, recvs = {}, {}                      # peer -> list of (ptr, bytes), in the order given
                for i in range(ns):
                    sends.setdefault(sp[i], []).append((sptr[i], sb[i]))
                for i in range(nr):
                    recvs.setdefault(rp[i], []).append((rptr[i], rb[i]))
                ops, rbufs = [], {}
                for peer, items in recvs.items():
                    if peer == rank:
                        continue
                    rbufs[peer] = torch.empty(sum(b for _, b in items), dtype=torch.uint8)
                    ops.append(dist.P2POp(dist.irecv, rbufs[peer], peer))
                for peer, items in sends.items():
                    host = np.empty(sum(b for _, b in items), dtype=np.uint8)
                    off = 0
                    for ptr, b in items:
                        if L.ecfft_device_copy(host.ctypes.data + off, ptr, b, 0) != 0:
                            return 1
                        off += b
                    if peer == rank:
                        rbufs[peer] = torch.from_numpy(host)
                    else:
                        ops.append(dist.P2POp(dist.isend, torch.from_numpy(host), peer))
                if ops:
                    for w in dist.batch_isend_irecv(ops):
                        w.wait()
                for peer, items in recvs.items():
                    host = rbufs[peer].numpy()
                    off = 0
                    for ptr, b in items:
                        if L.ecfft_device_copy(ptr, host.ctypes.data + off, b, 1) != 0:
                            return 1
                        off += b
                return 0
            except Exception as ex:  # pragma: no cover - surfaced as ECFFT_ERR_HIP by the library
                print("ecfft exchange callback failed:", ex)
                return 1
        cb = fftree.EXCHANGE_FN(exchange)
        h = ctypes.c_void_p()
        fftree._check(L.ecfft_comm_init_callback(world, rank, device, cb, None, ctypes.byref(h)))
        return Comm(h, keep=cb)

    @staticmethod
    def set_rccl_library(path):
        """ecfft_comm_set_rccl_library: the RCCL library the next (first) communicator of this process binds; None = default"""
        from . import fftree
        fftree._check(fftree.lib().ecfft_comm_set_rccl_library(path.encode() if path else None))

    @staticmethod
    def projection(world, rank, device=0, delay_us=25.0, link_gbps=0.0):
        """MEASUREMENT ONLY (ecfft_comm_init_projection): one rank of a `world`-rank job timed on its own — exchanges cost the
        modelled time on the stream, results are meaningless"""
        from . import fftree
        L = fftree.lib()
        if not L.has_hooks:
            raise fftree.EcfftError("ecfft_comm_init_projection is a measurement hook: load the hooks build (fftree.use_hooks_library())")
        h = ctypes.c_void_p()
        fftree._check(L.ecfft_comm_init_projection(world, rank, device, float(delay_us), float(link_gbps), ctypes.byref(h)))
        return Comm(h)

    def set_link_striping(self, min_gain_bytes):
        """ecfft_comm_set_link_striping: threshold (bytes off the most loaded link) above which a pairwise exchange of a split ENTER / EXIT
        is striped over all links of the mesh; 0 = whenever it helps, 2**64 - 1 = never (the default).  The same value on every rank
        (checked in the ranks' first agreement), and only before the communicator's first exchange."""
        from . import fftree
        fftree._check(self._L.ecfft_comm_set_link_striping(self._h, int(min_gain_bytes)))

    def abort(self):
        """ncclCommAbort (RCCL transports): unblocks exchanges in flight; later sharded calls on this communicator fail.  Returns
        False for a callback transport.  May be called from another thread than the blocked one."""
        L = self._L
        L.ecfft_comm_abort.restype, L.ecfft_comm_abort.argtypes = ctypes.c_int, [ctypes.c_void_p]
        return L.ecfft_comm_abort(self._h) == 0

    def stats(self, enable=None):
        """enable=True/False switches per-exchange timing on / off; enable=None reads {comm_ms, exchanges, bytes_sent} and resets"""
        from . import fftree
        L = self._L
        if enable is not None:
            fftree._check(L.ecfft_comm_stats_enable(self._h, int(enable)))
            return None
        ms, n, b = ctypes.c_double(), ctypes.c_double(), ctypes.c_double()
        fftree._check(L.ecfft_comm_stats_read(self._h, ctypes.byref(ms), ctypes.byref(n), ctypes.byref(b)))
        return {"comm_ms": ms.value, "exchanges": n.value, "bytes_sent": b.value}
