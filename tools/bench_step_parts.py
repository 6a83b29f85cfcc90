#!/usr/bin/env python3
"""Times the C-ABI entry points inside real training steps at benchmark shape (dev tool)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spvipes_amd import _abi
from spvipes_amd.data import MinibatchSampler, make_synthetic_group
from spvipes_amd.module import spVIPESmodule
from spvipes_amd.train import Trainer
B = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
G = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device("cuda:0")
torch.manual_seed(0)
groups = [make_synthetic_group(g, 16384, G, dev) for g in range(2)]
module = spVIPESmodule({0: G, 1: G}, use_labels=True).to(dev)
tr = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups])
sm = MinibatchSampler([16384, 16384], B, dev, seed=0)
module.train()
it = iter(sm.epoch())
for _ in range(2): tr.step(next(it), kl_weight=1.0)
names = [n for n in _abi._SIGNATURES if n not in ("spv_version", "spv_last_error")]
torch.cuda.synchronize()
_abi.profile_start(names)
import time
t0 = time.perf_counter(); N = 2
for _ in range(N): tr.step(next(it), kl_weight=1.0)
torch.cuda.synchronize(); el = (time.perf_counter() - t0) / N
prof = _abi.profile_stop()
tot = 0
for k, v in sorted(prof.items(), key=lambda kv: -sum(kv[1])):
    if v:
        tot += sum(v) / N
        print(f"  {k:22s} calls/step {len(v)/N:5.1f}  avg {np.mean(v)*1e3:8.1f} us  per-step {sum(v)/N*1e3:8.1f} us")
print(f"step {el*1e3:.3f} ms (with event overhead); sum of ABI calls {tot*1e3:.1f} us")
