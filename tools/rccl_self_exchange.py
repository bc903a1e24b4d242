#!/usr/bin/env python3
"""Round 6: the floor under ONE exchange of the split transforms on the REAL RCCL of this stack, measured with the one GPU a lease
has.  A communicator of world = 1 (ncclCommInitRank) and, exactly as RcclTransport::do_exchange (ecfft_amd/csrc/transport.h) issues
them, ncclGroupStart / k x ncclSend(self) / k x ncclRecv(self) / ncclGroupEnd on a compute stream, bracketed by HIP events on that
stream: the enqueue + kernel cost of a grouped send / receive set with no wire underneath — a LOWER bound of what an exchange costs on
xGMI, and the constant tools/split_project.py's `delay_us` stands for.  Also times the host side of the call (GroupStart..GroupEnd
returning) and a same-size device-to-device hipMemcpyAsync for scale.
usage: rccl_self_exchange.py [OUT.txt]"""
import ctypes
import statistics
import sys
import time

import torch
import torch.distributed  # noqa: F401  (maps torch's bundled librccl)

SIZES = [4, 64 << 10, 4 << 20, 16 << 20]
MSGS = [1, 7]
REPS = 50


def load_rccl():
    # the copy torch already mapped (same rule as RcclApi::load): RTLD_NOLOAD first
    for name in ("librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"):
        try:
            return ctypes.CDLL(name, mode=ctypes.RTLD_GLOBAL), name
        except OSError:
            continue
    raise SystemExit("librccl not found")


class UniqueId(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]


def main():
    out = open(sys.argv[1], "w") if len(sys.argv) > 1 else None

    def emit(*a):
        line = " ".join(str(x) for x in a)
        print(line)
        if out:
            out.write(line + "\n")

    torch.cuda.init()
    _ = torch.zeros(1, device="cuda")          # librccl gets mapped with torch's HIP runtime
    R, name = load_rccl()
    R.ncclGetUniqueId.argtypes = [ctypes.POINTER(UniqueId)]
    R.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, UniqueId, ctypes.c_int]
    R.ncclSend.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    R.ncclRecv.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
    R.ncclCommDestroy.argtypes = [ctypes.c_void_p]
    ver = ctypes.c_int(0)
    R.ncclGetVersion(ctypes.byref(ver))
    uid = UniqueId()
    assert R.ncclGetUniqueId(ctypes.byref(uid)) == 0
    comm = ctypes.c_void_p()
    rc = R.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0)
    assert rc == 0, f"ncclCommInitRank -> {rc}"
    mapped = sorted({l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l})
    emit(f"# real RCCL ({name}; mapped: {', '.join(mapped)}; version code {ver.value}), world = 1, device: {torch.cuda.get_device_name(0)}")
    emit("# grouped ncclSend/ncclRecv to SELF on a compute stream, HIP events on that stream, median / min / p90 of", REPS, "(after 5 warm-ups)")
    emit("# host_us = wall time for ncclGroupStart..ncclGroupEnd to RETURN (enqueue cost on the caller's thread); memcpy_us = hipMemcpyAsync D2D of the same bytes")
    emit(f"{'bytes/msg':>10} {'msgs':>4} {'event_us med':>12} {'min':>8} {'p90':>8} {'host_us med':>11} {'memcpy_us med':>13} {'GB/s (event)':>12}")
    stream = torch.cuda.Stream()
    s = ctypes.c_void_p(stream.cuda_stream)
    rows = []
    for size in SIZES:
        for k in MSGS:
            src = torch.empty(size * k, dtype=torch.uint8, device="cuda").random_()
            dst = torch.zeros(size * k, dtype=torch.uint8, device="cuda")
            ev, host, mc = [], [], []
            try:
              with torch.cuda.stream(stream):
                  for it in range(REPS + 5):
                      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                      a.record(stream)
                      t0 = time.perf_counter()
                      assert R.ncclGroupStart() == 0
                      for i in range(k):
                          assert R.ncclSend(src.data_ptr() + i * size, size, 0, 0, comm, s) == 0
                      for i in range(k):
                          assert R.ncclRecv(dst.data_ptr() + i * size, size, 0, 0, comm, s) == 0
                      assert R.ncclGroupEnd() == 0
                      t1 = time.perf_counter()
                      b.record(stream)
                      stream.synchronize()
                      if it >= 5:
                          ev.append(a.elapsed_time(b) * 1e3); host.append((t1 - t0) * 1e6)
                  assert torch.equal(src, dst), "self exchange moved the wrong bytes"
                  for it in range(REPS + 5):
                      a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                      a.record(stream)
                      dst.copy_(src, non_blocking=True)
                      b.record(stream)
                      stream.synchronize()
                      if it >= 5:
                          mc.append(a.elapsed_time(b) * 1e3)
            except AssertionError as ex:
                emit(f"{size:>10} {k:>4}  FAILED: {ex}")
                continue
            ev.sort()
            med = statistics.median(ev)
            emit(f"{size:>10} {k:>4} {med:>12.1f} {ev[0]:>8.1f} {ev[int(0.9 * len(ev))]:>8.1f} {statistics.median(host):>11.1f} {statistics.median(mc):>13.1f} {size * k / med / 1e3:>12.1f}")
            rows.append((size, k, med))
    # back-to-back exchanges inside one timed region (what a split transform does: exchange, kernels, exchange ...): per-exchange cost
    for size, k in ((4, 1), (4 << 20, 1)):
        src = torch.empty(size * k, dtype=torch.uint8, device="cuda").random_()
        dst = torch.zeros(size * k, dtype=torch.uint8, device="cuda")
        with torch.cuda.stream(stream):
            res = []
            for it in range(12):
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(stream)
                for _ in range(20):
                    R.ncclGroupStart(); R.ncclSend(src.data_ptr(), size, 0, 0, comm, s); R.ncclRecv(dst.data_ptr(), size, 0, 0, comm, s); R.ncclGroupEnd()
                b.record(stream)
                stream.synchronize()
                if it >= 2:
                    res.append(a.elapsed_time(b) * 1e3 / 20)
        emit(f"# 20 back-to-back exchanges of {size} B x {k}: {statistics.median(res):.1f} us each (median of 10)")
    R.ncclCommDestroy(comm)
    if out:
        out.close()


if __name__ == "__main__":
    main()
