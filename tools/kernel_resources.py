#!/usr/bin/env python3
"""VGPR / AGPR / SGPR / scratch of every kernel of the library's code object: compiles ecfft_capi.hip to gfx950 assembly (device
only, ~35 s) and reads the amdhsa metadata.  usage: kernel_resources.py [-DFLAG ...] > profiles/rNN/kernel_resources.txt"""
import os, re, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "ecfft_amd", "csrc", "ecfft_capi.hip")
with tempfile.TemporaryDirectory() as d:
    asm = os.path.join(d, "capi.s")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-Wno-pass-failed", "-S", "--cuda-device-only", "-o", asm, src] + sys.argv[1:],
                   check=True, stderr=subprocess.DEVNULL)
    s = open(asm).read()
items = re.findall(r"- \.agpr_count:\s+(\d+).*?\.name:\s+(\S+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_count:\s+(\d+).*?\.vgpr_count:\s+(\d+)", s, re.S)
names = subprocess.run(["c++filt"], input="\n".join(i[1] for i in items), capture_output=True, text=True).stdout.splitlines()
rows = []
for (ag, _, priv, sg, vg), dn in zip(items, names):
    dn = re.sub(r"^void ", "", dn)
    dn = dn[:dn.index("(")] if "(" in dn and not dn.startswith("ecfft::k_foreach") else dn[:100]
    rows.append((dn, int(vg), int(ag), int(sg), int(priv)))
print(f"{len(rows)} kernels; flags: {' '.join(sys.argv[1:]) or '(default build)'}")
print(f"{'scratch B':>9} {'vgpr':>5} {'agpr':>5} {'sgpr':>5}  kernel")
for dn, vg, ag, sg, priv in sorted(rows, key=lambda r: (-r[4], r[0])):
    print(f"{priv:>9} {vg:>5} {ag:>5} {sg:>5}  {dn}")
