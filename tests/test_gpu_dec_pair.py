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


@pytest.mark.parametrize("B,G,G2,H", [(128, 2000, 2000, 64), (100, 333, 500, 32), (300, 1000, 777, 64)])
def test_fc1_plain_pair_launches_give_the_bits_of_the_per_group_calls(dev, B, G, G2, H):
    """n_hidden <= 64 on resident images: the LDS-DMA fc1 kernels do not take the shape, and spv_enc_fc1_fwd_grouped / _bwd_grouped run the
    register-staged GEMM, its epilogue, the dh-image prep and the weight-gradient GEMM for BOTH groups in one grid each (gemm_pair_kernel):
    same kernel bodies, same bits as the per-group entry points (nn/networks.py:119 and its backward)"""
    from spvipes_amd import ops
    rng = np.random.default_rng(B + G)
    g = torch.Generator().manual_seed(5)
    mk = lambda *s: (torch.randn(*s, generator=g) * 0.05)
    counts, rows, params, dhs = [], [], [], []
    for b_, g_ in ((B, G), (B, G2)):
        Xh = (rng.poisson(2.0, size=(b_ + 9, g_)) * (rng.random((b_ + 9, g_)) < 0.3)).astype(np.float32)
        Xh[:, 0] += 1
        counts.append(ops.GroupCounts(torch.tensor(Xh.astype(np.uint16).view(np.int16)).to(dev), g_, 0, resident=True))
        rows.append(torch.tensor(rng.permutation(b_ + 9)[:b_], dtype=torch.int32, device=dev))
        params.append([mk(H, g_), mk(H), mk(H, g_), mk(H)])
        dhs.append(torch.randn(b_, 2 * H, generator=g).to(dev))
    pg = [[t.clone().to(dev).requires_grad_(True) for t in ps] for ps in params]
    outs = ops.EncoderFC1Grouped.apply(counts, rows, [B, B], 1, [ops.Workspace(dev), ops.Workspace(dev)], None, *pg[0], *pg[1])
    ((outs[0] * dhs[0]).sum() + (outs[2] * dhs[1]).sum()).backward()
    torch.cuda.synchronize()
    for k in range(2):
        ps = [t.clone().to(dev).requires_grad_(True) for t in params[k]]
        h1, lib = ops.EncoderFC1.apply(counts[k], rows[k], B, *ps, 1, ops.Workspace(dev))
        (h1 * dhs[k]).sum().backward()
        torch.cuda.synchronize()
        assert torch.equal(h1.detach(), outs[2 * k].detach()) and torch.equal(lib, outs[2 * k + 1])
        for a, b in zip(ps, pg[k]):
            assert torch.equal(a.grad, b.grad), (k, float((a.grad - b.grad).abs().max()))
        # and the values themselves against fp64 on the f16-rounded operands
        Xsel = torch.log1p(torch.tensor(np.asarray(counts[k].X.cpu().view(torch.uint16).numpy(), dtype=np.float32))[rows[k].cpu().long()])
        xr = Xsel.to(torch.float16).double()
        W = (torch.cat([params[k][0], params[k][2]]) * 256.0).to(torch.float16).double() / 256.0
        ref = torch.relu(xr @ W.t() + torch.cat([params[k][1], params[k][3]]).double())
        torch.testing.assert_close(h1.detach().cpu().double(), ref, rtol=1e-4, atol=1e-4)
