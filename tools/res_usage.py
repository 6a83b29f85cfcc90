#!/usr/bin/env python3
"""Summarises hipcc -Rpass-analysis=kernel-resource-usage output (dev tool). usage: res_usage.py <stderr file> [name filter]"""
import re, subprocess, sys
t = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
blocks = re.split(r"remark: [^\n]*Function Name: ", t)[1:]
names = [b.split("\n")[0].strip().split()[0] for b in blocks]
dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
def g(b, k):
    m = re.search(k + r": (\d+)", b)
    return m.group(1) if m else "?"
for b, d in zip(blocks, dem):
    if flt not in d:
        continue
    d = d.replace("spv::gemm_kernel<spv::GemmCfg<", "gemm<").replace("> >(spv::GemmParams)", ">").replace("void ", "").replace("(unsigned short)", "")
    print(f"{d[:78]:78s} VGPR {g(b, 'VGPRs'):>4s} AGPR {g(b, 'AGPRs'):>4s} spill {g(b, 'VGPRs Spill'):>4s} LDS {g(b, 'LDS Size .bytes/block.'):>7s} occ {g(b, 'Occupancy .waves/SIMD.')}")
