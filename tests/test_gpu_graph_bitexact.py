"""GPU: what bench.py TIMES -- the hipGraph replay of the step (two streams inside the graph, grouped LDS-DMA launches, weight
images written by the Adam kernel, device-side noise / dropout generator) -- against the same steps launched eagerly, at the
benchmark's own shape (BASELINE configs[1]: B 4096 x G 10 000, H 128, 25 / 10, bf16 mode, resident uint16 counts).

Every kernel on the path is deterministic (fixed-order reductions, no data-path atomics but the PoE backward's <= 2-operand scatter),
so eager and replayed steps started from identical parameters and the same device-RNG counter must agree BIT FOR BIT: the flat
gradient after every step, and after the last step the parameters, both Adam moments, the packed weight images and the BatchNorm
running statistics.  The eager oracle comparisons (tests/test_gpu_fullsize_parity.py) cannot see a bug that only exists under
replay -- a missing edge in the captured graph, a workspace block reused while a branch still reads it (reference path:
module/spVIPESmodule.py:425-899; training loop: model/base/training_mixin.py:89-123)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from spvipes_amd import _abi
    _abi.load()
    return torch.device("cuda:0")


@pytest.fixture(scope="module")
def c2_groups(dev):
    from spvipes_amd.data import make_synthetic_group
    return [make_synthetic_group(g, 12_288, 10_000, dev) for g in range(2)]


def _trainer(dev, groups, overlap):
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    torch.manual_seed(0)   # same initial parameters AND the same generator key (Trainer derives it from torch's seed)
    module = spVIPESmodule({0: 10_000, 1: 10_000}, use_labels=True, n_hidden=128, n_dimensions_shared=25, n_dimensions_private=10,
                           precision="bf16").to(dev)   # dropout at the reference's default 0.1: the masks are part of what must agree
    trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups], overlap_allreduce=overlap)
    module.train()
    return module, trainer


def _state(module, trainer):
    torch.cuda.synchronize()
    return {"flat": trainer.fp.flat.clone(), "m": trainer.opt.m.clone(), "v": trainer.opt.v.clone(),
            "images": [img.clone() for _ws, _k, img, _p, _t in trainer._img_specs],
            "buffers": {k: b.clone() for k, b in module.named_buffers()}, "counter": int(module._rng_counter)}


@pytest.mark.parametrize("overlap", [False, True])
def test_graph_replay_is_bit_identical_to_eager_steps_at_the_bench_shape(dev, c2_groups, overlap):
    """``overlap``: the data-parallel form of the step (backward pass split at the encoder / decoder cut, TWO graphs sharing a pool)"""
    B, steps = 4096, 3
    rng = np.random.default_rng(7)
    batches = [[torch.tensor(rng.permutation(12_288)[:B].astype(np.int32), device=dev) for _ in range(2)] for _ in range(steps + 1)]
    runs = []
    for use_graph in (False, True):
        module, trainer = _trainer(dev, c2_groups, overlap)
        trainer._ensure_images()
        if use_graph:
            keep = {k: b.clone() for k, b in module.named_buffers()}
            trainer.capture(batches[steps])   # (warm-up + capture run forward passes: they move the BatchNorm running statistics only)
            with torch.no_grad():
                for k, b in module.named_buffers():
                    b.copy_(keep[k])
            assert trainer.graph is not None and (trainer.graph2 is not None) == overlap
        assert int(module._rng_counter) == 0
        grads, losses = [], []
        for i in range(steps):
            lo = trainer.step(batches[i], kl_weight=0.5 + 0.25 * i)
            torch.cuda.synchronize()
            grads.append(trainer.fp.grad.clone())
            losses.append(lo.loss.detach().clone())
        runs.append((grads, losses, _state(module, trainer)))
    (g_e, l_e, s_e), (g_r, l_r, s_r) = runs
    for i in range(steps):
        assert torch.equal(l_e[i], l_r[i]), (i, float(l_e[i]), float(l_r[i]))
        if not torch.equal(g_e[i], g_r[i]):
            module, trainer = _trainer(dev, c2_groups, overlap)
            base, bad = trainer.fp.flat.data_ptr(), []
            for name, p in module.named_parameters():
                off = (p.data_ptr() - base) // 4
                d = (g_e[i][off:off + p.numel()] != g_r[i][off:off + p.numel()])
                if bool(d.any()):
                    bad.append(f"{name}: {int(d.sum())} of {p.numel()}")
            raise AssertionError(f"step {i}: gradient differs between eager and replay in " + "; ".join(bad))
        assert bool(torch.isfinite(g_e[i]).all())
    assert s_e["counter"] == s_r["counter"] == steps
    for k in ("flat", "m", "v"):
        assert torch.equal(s_e[k], s_r[k]), k
    assert len(s_e["images"]) == 4
    for a, b in zip(s_e["images"], s_r["images"]):
        assert torch.equal(a, b)
    for k in s_e["buffers"]:
        assert torch.equal(s_e["buffers"][k], s_r["buffers"][k]), k
