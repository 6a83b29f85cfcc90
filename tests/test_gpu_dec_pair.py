"""GPU: the decoder's per-group launches as ONE grid per kernel for both groups (ops.DEC_PAIR: spv_dec_tables / logits / lse / nb_fwd /
heads_bwd _grouped, spv_gemm_bf16_grouped; group = blockIdx.z) against the per-group launches on two streams.  A pair grid runs the
single-group kernel bodies unchanged, so the step must agree BIT FOR BIT: loss, flat gradient after every step, parameters, Adam moments
and BatchNorm buffers after the last -- eagerly and under hipGraph replay, at the reference's default minibatch (128 cells: where the pair
form is the default), at a ragged one, and at the bench's shape (module/spVIPESmodule.py:425-899)."""
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


def _run(dev, groups, genes, B, mode, use_graph, monkeypatch, steps=3, serial=False):
    from spvipes_amd import ops
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    monkeypatch.setattr(ops, "DEC_PAIR", mode)
    monkeypatch.setattr(ops, "SERIAL_STREAMS", serial)
    torch.manual_seed(0)
    module = spVIPESmodule({0: genes[0], 1: genes[1]}, use_labels=True, n_hidden=128, n_dimensions_shared=25, n_dimensions_private=10, precision="bf16").to(dev)
    trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups])
    module.train()
    trainer._ensure_images()
    rng = np.random.default_rng(11)
    n = min(g.counts.X.shape[0] for g in groups)
    batches = [[torch.tensor(rng.permutation(n)[:B].astype(np.int32), device=dev) for _ in range(2)] for _ in range(steps + 1)]
    if use_graph:
        keep = {k: b.clone() for k, b in module.named_buffers()}
        trainer.capture(batches[steps])
        with torch.no_grad():
            for k, b in module.named_buffers():
                b.copy_(keep[k])
    out = []
    for i in range(steps):
        lo = trainer.step(batches[i], kl_weight=0.7)
        torch.cuda.synchronize()
        out.append((lo.loss.detach().clone(), trainer.fp.grad.clone()))
    state = {"flat": trainer.fp.flat.clone(), "m": trainer.opt.m.clone(), "v": trainer.opt.v.clone(), **{k: b.clone() for k, b in module.named_buffers()}}
    return out, state


@pytest.mark.parametrize("genes,B,use_graph,serial", [((2000, 1500), 128, False, False), ((2000, 1500), 128, True, True), ((3001, 2000), 200, False, True),
                                                        ((10_000, 10_000), 4096, True, False)])
def test_pair_grids_give_the_bits_of_the_per_group_launches(dev, genes, B, use_graph, serial, monkeypatch):
    from spvipes_amd.data import make_synthetic_group
    groups = [make_synthetic_group(g, max(3 * B, 1000), genes[g], dev) for g in range(2)]
    (ref, s_ref), (got, s_got) = (_run(dev, groups, genes, B, m, use_graph, monkeypatch, serial=serial) for m in ("0", "1"))
    for i, ((l0, g0), (l1, g1)) in enumerate(zip(ref, got)):
        assert torch.isfinite(g1).all() or torch.equal(torch.isnan(g0), torch.isnan(g1))
        assert torch.equal(l0, l1), (i, float(l0), float(l1))
        assert torch.equal(torch.nan_to_num(g0), torch.nan_to_num(g1)), (i, float((torch.nan_to_num(g0) - torch.nan_to_num(g1)).abs().max()))
    for k in s_ref:
        assert torch.equal(torch.nan_to_num(s_ref[k]), torch.nan_to_num(s_got[k])), k


def test_auto_mode_pairs_small_steps_only(monkeypatch):
    from spvipes_amd import ops
    monkeypatch.setattr(ops, "DEC_PAIR", "auto")
    monkeypatch.setattr(ops, "SERIAL_AUTO", True)
    monkeypatch.setattr(ops, "SERIAL_STREAMS", False)
    for B, G, n, small in ((128, 2000, 2, True), (1024, 2000, 2, True), (128, 10_000, 2, True), (512, 10_000, 2, False), (4096, 10_000, 2, False), (128, 2000, 3, False)):
        ops.set_step_shape(B, G, n)
        assert ops.dec_pair_for(B, n) == small and ops.serial_streams() == small, (B, G, n)
    ops.set_step_shape(4096, 10_000, 2)
