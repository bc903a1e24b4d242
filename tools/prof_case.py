#!/usr/bin/env python3
"""One hot-path case, nothing else, for rocprofv3: builds the tree, then runs `reps` calls of one operation on
device-resident data.  tools/prof_case.sh wraps it with kernel-trace and PMC passes.
usage: prof_case.py FIELD LOG_N OP [REPS] [--count FILE]     OP = enter | exit | extend | both
ECFFT_PROF_BATCH=B in the environment: every call transforms B polynomials laid end to end (ecfft_enter_many / _exit_many) — the
batched mode of bench.py's `batched` object; ECFFT_LIB=path: another build of the library (ecfft_amd/variants/*.so).
--count FILE: instead of the plain reps, run ONE rep with the library's per-launch profiler on and write the number of
launches per kernel class of that rep to FILE (tools/counters_json.py uses it to cut the timed launches out of a trace)."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import ecfft_amd
from bench import synth

field, log_n, op = sys.argv[1], int(sys.argv[2]), sys.argv[3]
reps = int(sys.argv[4]) if len(sys.argv) > 4 and not sys.argv[4].startswith("--") else 3
count_file = sys.argv[sys.argv.index("--count") + 1] if "--count" in sys.argv else None
n = 1 << log_n
F = ecfft_amd.FIELDS[field]
tree = F.build_fftree(2 * n if op == "extend" else n)
B = int(os.environ.get("ECFFT_PROF_BATCH", "1"))
h = synth(field, n, 7) if B == 1 else np.concatenate([synth(field, n, 7 + i) for i in range(B)])
x = torch.from_numpy(h.view(np.int64) if field == "secp256k1" else h.view(np.int32)).cuda()
torch.cuda.synchronize()


def one():
    if op == "enter":
        return tree.enter(x, count=B)
    if op == "exit":
        return tree.exit(x, count=B)
    if op == "extend":
        return tree.extend(x, ecfft_amd.Moiety.S1, count=B)
    return tree.exit(tree.enter(x, count=B), count=B)


if count_file:
    import json
    one(); torch.cuda.synchronize()
    tree.profile(True)
    one(); torch.cuda.synchronize()
    classes = tree.profile_read()
    tree.profile(False)
    with open(count_file, "w") as f:
        json.dump({c["name"]: c["launches"] for c in classes if c["launches"]}, f)
    print("counted", classes)
    sys.exit(0)
for _ in range(reps):
    y = one()
    torch.cuda.synchronize()
print("done", field, log_n, op, reps)
