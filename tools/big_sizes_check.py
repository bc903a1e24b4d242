import sys, time, numpy as np, torch
sys.path.insert(0, '.')
import ecfft_amd
from bench import synth
for field, ln in [(a.split(":")[0], int(a.split(":")[1])) for a in (sys.argv[1:] or ["secp256k1:24", "m31:27"])]:
    F = ecfft_amd.FIELDS[field]; n = 1 << ln
    t0 = time.time(); t = F.build_fftree(n); torch.cuda.synchronize(); print(field, ln, "build", round(time.time() - t0, 2), "s", flush=True)
    x = synth(field, n, 5)
    d = torch.from_numpy(x.view(np.int64) if field == "secp256k1" else x.view(np.int32)).cuda()
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.time(); ev = t.enter(d); torch.cuda.synchronize(); t1 = time.time(); back = t.exit(ev); torch.cuda.synchronize(); t2 = time.time()
    print("  enter", round((t1 - t0) * 1e3, 2), "ms  exit", round((t2 - t1) * 1e3, 2), "ms  round trip", bool(torch.equal(back, d)), flush=True)
    lo = d.clone(); lo[n // 2:] = 0
    el = t.enter(lo)
    print("  extend S0->S1 of a degree<n/2 polynomial:", bool(torch.equal(t.extend(el[0::2].contiguous(), ecfft_amd.Moiety.S1), el[1::2].contiguous())), flush=True)
    # Horner spot check of ENTER at 3 leaves through a small subtree transform: ENTER of the first 2^10 coefficients on T_n's subtree
    print("  mem GiB", round(torch.cuda.mem_get_info()[0] / 2**30, 1), "free", flush=True)
    del t, d, ev, back, el, lo
    torch.cuda.empty_cache()
