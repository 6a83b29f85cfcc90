#!/usr/bin/env python3
"""Lists the torch (non-spv) operators launched by one eager training step (dev tool)."""
import os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spvipes_amd.data import make_synthetic_group
from spvipes_amd.module import spVIPESmodule
from spvipes_amd.train import Trainer

dev = torch.device("cuda:0")
B, G, N = 4096, 10000, 8192
groups = [make_synthetic_group(g, N, G, dev) for g in (0, 1)]
torch.manual_seed(0)
mod = spVIPESmodule({0: G, 1: G}, use_labels=True, n_hidden=128, n_dimensions_shared=25, n_dimensions_private=10, precision="bf16").to(dev)
mod.train()
tr = Trainer(mod, [g.counts for g in groups], labels=[g.labels for g in groups])
rows = [torch.randperm(N, device=dev)[:B].to(torch.int32) for _ in (0, 1)]
for _ in range(3):
    tr.step(rows)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=False) as prof:
    tr.step(rows)
    torch.cuda.synchronize()
ev = prof.key_averages()
rows_ = [(e.key, e.count, e.device_time_total if hasattr(e, "device_time_total") else e.cuda_time_total) for e in ev]
print("== aten ops / kernels by count")
for k, c, t in sorted(rows_, key=lambda r: -r[1])[:70]:
    print(f"{c:5d} {t:10.1f} us  {k[:110]}")
