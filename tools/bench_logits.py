#!/usr/bin/env python3
"""spv_dec_logits at several K: separates the epilogue / launch cost from the main loop (dev tool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spvipes_amd import _abi
from spvipes_amd._abi import ptr, stream_ptr
dev = torch.device("cuda:0")
Bp, Gp = 4096, 10240
out = torch.empty(Bp, Gp, dtype=torch.float16, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for K in (32, 64, 128, 320, 640):
    Am = torch.randint(-2000, 2000, (Bp, K), dtype=torch.int16, device=dev)
    Wm = torch.randint(-2000, 2000, (Gp, K), dtype=torch.int16, device=dev)
    t = timeit(lambda: _abi.call("spv_dec_logits", ptr(Am), None, ptr(Wm), None, K, Bp, Gp, 1, ptr(out), 0, stream_ptr()))
    print(f"K={K:4d}: {t:7.1f} us   ({2.0*Bp*Gp*K/t/1e6:7.1f} TFLOP/s)")
