#!/usr/bin/env python3
"""Builds the shipped library, the hooks build and any number of -D variants of the same sources in parallel (one hipcc each).
usage: build_variants.py [NAME=DEFINE[,DEFINE...] ...]      e.g.  build_variants.py wl0=ECFFT_WAVE_LOCAL=0 bs0=ECFFT_BATCH_SPLIT=0
-> ecfft_amd/libecfft_hip.so, tests/hooks/libecfft_hip_hooks.so, ecfft_amd/variants/NAME.so"""
import os
import sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ecfft_amd import build as b  # noqa: E402
d = os.path.join(ROOT, "ecfft_amd", "variants")
os.makedirs(d, exist_ok=True)
jobs = [(None, ()), (os.path.join(ROOT, "tests", "hooks", "libecfft_hip_hooks.so"), ("ECFFT_TEST_HOOKS",))]
for a in sys.argv[1:]:
    name, defs = a.split("=", 1)
    jobs.append((os.path.join(d, name + ".so"), tuple(defs.split(","))))
with ThreadPoolExecutor(len(jobs)) as ex:
    for r in ex.map(lambda j: b.build(force=True, out=j[0], defines=j[1]), jobs):
        print("built", r)
