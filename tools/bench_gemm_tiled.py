#!/usr/bin/env python3
"""Times the decoder-backward GEMMs on [B,G] tiled operands for several split counts (dev tool)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spvipes_amd import _abi, ops
from spvipes_amd.ops import _gemm_slabs
dev = torch.device("cuda:0")
Bp, Gp, G, B = 4096, 10240, 10000, 4096
KMP = 320
ws = ops.Workspace(dev)
dL = torch.randint(-2000, 2000, (Bp, Gp), dtype=torch.int16, device=dev)
Am = torch.randint(-2000, 2000, (Bp, KMP), dtype=torch.int16, device=dev)
Wm = torch.randint(-2000, 2000, (Gp, KMP), dtype=torch.int16, device=dev)
Aps = torch.randint(-2000, 2000, (Bp, 48), dtype=torch.int16, device=dev)
T = Gp // 32
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for sp in (1, 2, 4, 8, 16):
    t1 = timeit(lambda: _gemm_slabs(True, dL, None, Gp, Am, None, KMP, G, KMP, Bp, 1, sp, ws, f"a{sp}", a_tiles=T))
    t2 = timeit(lambda: _gemm_slabs(False, dL, None, Gp, Wm, None, KMP, B, KMP, G, 1, sp, ws, f"d{sp}", a_tiles=T))
    t3 = timeit(lambda: _gemm_slabs(True, dL, None, Gp, Aps, None, 48, G, 16, Bp, 1, sp, ws, f"b{sp}", a_tiles=T))
    t4 = timeit(lambda: _gemm_slabs(False, dL, None, Gp, Wm, None, KMP, B, 16, G, 1, sp, ws, f"e{sp}", a_tiles=T))
    print(f"splits {sp:2d}: wgrad N=320 {t1:7.1f} us | dgrad N=320 {t2:7.1f} us | wgrad N=16 {t3:7.1f} us | dgrad N=16 {t4:7.1f} us")
