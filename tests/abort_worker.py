"""Worker of tests/test_distributed.py::test_comm_abort_unblocks_a_rank_whose_peer_never_arrives.
`main`: rank 0 of a two-rank communicator; starts the peer process, makes a split EXTEND call the peer never joins, and a second
host thread calls Comm.abort() (ecfft_comm_abort -> ncclCommAbort) after two seconds.  `peer <id hex>`: rank 1, attaches and sleeps."""
import ctypes
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ecfft_amd  # noqa: E402
from ecfft_amd import fftree, distributed as D  # noqa: E402

L = fftree.lib()
D.Comm.set_rccl_library(os.environ["ECFFT_WORKER_RCCL_LIB"])          # the stand-in for librccl (tests/stub_rccl)
if sys.argv[1] == "peer":
    h = ctypes.c_void_p()
    assert L.ecfft_comm_init_rank(ctypes.create_string_buffer(bytes.fromhex(sys.argv[2]), 128), 2, 1, 0, ctypes.byref(h)) == 0
    time.sleep(8)                       # never makes the sharded call
    os._exit(0)
buf = ctypes.create_string_buffer(128)
assert L.ecfft_comm_get_unique_id(buf) == 0
p = subprocess.Popen([sys.executable, os.path.abspath(__file__), "peer", bytes(buf.raw).hex()])
h = ctypes.c_void_p()
assert L.ecfft_comm_init_rank(buf, 2, 0, 0, ctypes.byref(h)) == 0
comm = D.Comm(h)
n = 1 << 10
tree = ecfft_amd.FIELDS["m31"].build_fftree(2 * n)
x = torch.zeros(n // 2, dtype=torch.int32, device="cuda")
threading.Timer(2.0, lambda: print("abort ->", comm.abort(), flush=True)).start()
t0 = time.time()
try:
    tree.extend_sharded(comm, x, n, ecfft_amd.Moiety.S1)
    torch.cuda.synchronize()
    print("RETURNED_OK", flush=True)
except Exception as ex:
    print("RETURNED_ERROR after %.1f s: %s" % (time.time() - t0, type(ex).__name__), flush=True)
p.wait()
