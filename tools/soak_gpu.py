#!/usr/bin/env python3
"""Randomised GPU-vs-oracle soak (run on the GPU box): every size 2^1..2^max, both fields, ENTER / EXIT / EXTEND (both
moieties, batched), MEXTEND / REDC / MOD / VANISH / DEGREE.  Not part of pytest: a wider net after kernel refactors."""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ecfft_amd
from oracle import oracle

max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 15
bad = 0
for field in ("secp256k1", "m31"):
    F = oracle.field(field)
    G = ecfft_amd.FIELDS[field]
    N = 1 << max_log
    ot = F.build_fftree(N)
    gt = G.build_fftree(N)
    rng = np.random.default_rng(99)

    def rand(n):
        if field == "m31":
            return rng.integers(0, 2**31 - 1, n, dtype=np.uint32)
        ints = [int.from_bytes(rng.bytes(32), "little") % (2**256 - 2**32 - 977) for _ in range(n)] if n <= 64 else None
        if ints is not None:
            return F.from_ints(ints)
        a = rng.integers(0, 2**64, size=(n, 4), dtype=np.uint64); a[:, 3] &= 0x7FFFFFFFFFFFFFFF
        return a

    def chk(name, got, want):
        global bad
        if not np.array_equal(np.asarray(got), np.asarray(want)):
            bad += 1; print("MISMATCH", field, name)

    t0 = time.time()
    for ln in range(0, max_log + 1):
        n = 1 << ln
        c = rand(n)
        ev = ot.enter(c)
        chk(f"enter 2^{ln}", gt.enter(c), ev)
        r = rand(n)
        chk(f"exit 2^{ln}", gt.exit(r), ot.exit(r))
        if ln < max_log:
            for mo, omo in ((ecfft_amd.Moiety.S0, oracle.S0), (ecfft_amd.Moiety.S1, oracle.S1)):
                chk(f"extend 2^{ln} {omo}", gt.extend(r, mo), ot.extend(r, omo))
                if ln <= 12:
                    chk(f"mextend 2^{ln} {omo}", gt.mextend(r, mo), ot.mextend(r, omo))
        if 1 <= ln <= 12:
            for cnt in (2, 3):
                big = rand(n * cnt)
                want = np.concatenate([ot.enter(big[i * n:(i + 1) * n]) for i in range(cnt)])
                chk(f"enter_many 2^{ln} x{cnt}", gt.enter(big, cnt), want)
                want = np.concatenate([ot.exit(big[i * n:(i + 1) * n]) for i in range(cnt)])
                chk(f"exit_many 2^{ln} x{cnt}", gt.exit(big, cnt), want)
        if 1 <= ln <= 11:
            a = ot.table(oracle.T_XNN_S, n)
            chk(f"redc0 2^{ln}", gt.redc_z0(r, a), ot.redc(r, a, oracle.S0))
            chk(f"redc1 2^{ln}", gt.redc_z1(r, a), ot.redc(r, a, oracle.S1))
            cc = ot.table(oracle.T_Z0Z0, n)
            chk(f"mod 2^{ln}", gt.modular_reduce(r, a, cc), ot.modular_reduce(r, a, cc))
            dom = rand(n // 2) if n >= 2 else None
            if dom is not None and ln >= 1:
                chk(f"vanish 2^{ln}", gt.vanish(dom), ot.vanish(dom))
            lo = c.copy(); k = int(rng.integers(0, n)); lo[k + 1:] = 0
            el = ot.enter(lo)
            if gt.degree(el) != ot.degree(el):
                bad += 1; print("MISMATCH degree", field, ln)
    print(field, "sizes 2^0..2^%d done in %.1fs" % (max_log, time.time() - t0), "mismatches so far:", bad)
print("SOAK", "FAILED" if bad else "OK")
sys.exit(1 if bad else 0)
