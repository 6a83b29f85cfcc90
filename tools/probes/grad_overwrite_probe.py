"""Probe (dev tool, GPU): is every element of the flat gradient buffer that belongs to a parameter OVERWRITTEN by a training step
(so that zeroing it at the start of the step is unnecessary)?  Fills the buffer with NaN before forward + backward and reports the
parameters that still hold a NaN afterwards."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from spvipes_amd.data import MinibatchSampler, make_synthetic_group
from spvipes_amd.module import spVIPESmodule
from spvipes_amd.train import Trainer
from spvipes_amd import plan as plan_mod

dev = torch.device("cuda:0")
groups = [make_synthetic_group(g, 2048, 600, dev) for g in range(2)]
for mode in ("label", "cluster", "paired"):
    torch.manual_seed(0)
    kw = dict(n_hidden=128, n_dimensions_shared=10, n_dimensions_private=5)
    if mode == "label":
        module = spVIPESmodule({0: 600, 1: 600}, use_labels=True, **kw).to(dev)
        trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups])
    else:
        import numpy as np
        rng = np.random.default_rng(0)
        n = 2048
        i = np.repeat(np.arange(n), 3); j = rng.integers(0, n, size=3 * n); v = rng.random(3 * n).astype(np.float32) + 0.1
        import scipy.sparse as sp
        P = sp.coo_matrix((v, (i, j)), shape=(n, n)).tocsr()
        module = spVIPESmodule({0: 600, 1: 600}, use_labels=False, transport_plan=P, pair_data=(mode == "paired"), **kw).to(dev)
        trainer = Trainer(module, [g.counts for g in groups], components=[g.labels for g in groups] if mode == "cluster" else None)
    sampler = MinibatchSampler([2048, 2048], 256, dev, seed=0)
    module.train()
    rows = next(iter(sampler.epoch()))
    trainer.step(rows, kl_weight=1.0)   # warm
    orig_zero = trainer.fp.grad.zero_
    trainer.fp.grad.zero_ = lambda: trainer.fp.grad.fill_(float("nan"))
    trainer.step(rows, kl_weight=1.0, optimizer_step=False)
    bad = [(n_, int(torch.isnan(p.grad).sum()), p.numel()) for n_, p in module.named_parameters() if p.grad is not None and bool(torch.isnan(p.grad).any())]
    print(mode, "parameters with un-overwritten gradient elements:", bad if bad else "none")
