#!/usr/bin/env python3
"""Times each C-ABI entry point at benchmark shapes (dev tool).  usage: bench_kernels.py [B] [G] [precision]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spvipes_amd import _abi, ops
from tests import _hip_harness as H
from spvipes_amd.data import make_synthetic_group

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
prec = sys.argv[3] if len(sys.argv) > 3 else "bf16"
nsplit = 3 if prec == "fp32" else 1
dev = torch.device("cuda:0")
grp = make_synthetic_group(0, max(B, 8192), G, dev)
rows = torch.randperm(grp.counts.n_cells, device=dev)[:B].to(torch.int32)
H, n_p, n_s = 128, 10, 25
g = torch.Generator(device=dev).manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g, device=dev) * sc).requires_grad_(True)
ws = ops.Workspace(dev)
enc = [r(H, G, sc=0.01), r(H, sc=0.01), r(H, G, sc=0.01), r(H, sc=0.01)]
dec = dict(zp=r(B, n_p), zs=r(B, n_s), m=r(B, 256), Wp=r(G, n_p, sc=0.5), cp=r(G, sc=0.3), Ws=r(G, n_s, sc=0.3), cs=r(G, sc=0.3),
           Wm=r(G, 256 + n_p + n_s, sc=0.05), bm=r(G, sc=0.2), px_r=r(G))
w = torch.full((B,), 1.0 / B, device=dev)
names = list(_abi._SIGNATURES)
def step():
    h1, lib = ops.EncoderFC1.apply(grp.counts, rows, B, *enc, nsplit, ws)
    loss, rec = H.DecoderNBLoss.apply(grp.counts, rows, B, *dec.values(), lib, w, nsplit, True, ws)
    (loss + h1.sum() * 1e-6).backward()
for _ in range(3): step()
torch.cuda.synchronize()
_abi.profile_start([n for n in names if n.startswith("spv_") and n not in ("spv_version", "spv_last_error")])
N = 10
t0 = time.perf_counter()
for _ in range(N): step()
torch.cuda.synchronize()
el = (time.perf_counter() - t0) / N
prof = _abi.profile_stop()
print(f"B={B} G={G} {prec}: {el*1e3:.3f} ms per fwd+bwd (one group)")
tot = 0
for k, v in prof.items():
    if v:
        per = sum(v) / N
        tot += per
        print(f"  {k:22s} calls/step {len(v)/N:5.1f}  avg {np.mean(v)*1e3:8.1f} us  per-step {per*1e3:8.1f} us")
print(f"  sum of ABI calls {tot*1e3:.1f} us")
