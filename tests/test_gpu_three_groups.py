"""GPU: BASELINE config 4's structure -- THREE groups with cluster-matched PoE -- which the reference cannot run
(data/prepare_adatas.py:94-95; spVIPESmodule.py:283-286, 723-726).  SURVEY.md 8d asks for a throughput-only path with the N-expert
generalisation of ``_product_of_experts`` (spVIPESmodule.py:573-581); there is no reference output to compare with, so these are
CONSISTENCY checks: the N-group PoE kernels (csrc/spv_poe_n.h) against a plain torch fp64 restatement of their definition
(values and gradients), and a three-group training step that runs, is bit-reproducible and learns."""
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


def _torch_poe_components(loc, logvar, comp, eps, n_comp):
    """the definition in csrc/spv_poe_n.h, fp64, autograd"""
    NG = len(loc)
    means = []
    for h in range(NG):
        oh = torch.nn.functional.one_hot(comp[h].long(), n_comp).double()           # [B, C]
        cnt = oh.sum(0)                                                            # [C]
        m_loc = (oh.t() @ loc[h]) / cnt.clamp(min=1).unsqueeze(1)
        m_lv = (oh.t() @ logvar[h]) / cnt.clamp(min=1).unsqueeze(1)
        means.append((m_loc, m_lv, cnt))
    out = []
    for g in range(NG):
        inv = torch.exp(-logvar[g])
        prec, num = 1.0 + inv, loc[g] * inv
        c = comp[g].long()
        for h in range(NG):
            if h == g:
                continue
            m_loc, m_lv, cnt = means[h]
            present = (cnt[c] > 0).double().unsqueeze(1)
            w = torch.exp(-m_lv[c]) * present
            prec = prec + w
            num = num + m_loc[c] * w
        jl, jv = num / prec, -torch.log(prec)
        sc = torch.exp(0.5 * jv)
        sq = sc.clamp(min=1e-6)
        z = jl + sq * eps[g]
        kl = (0.5 * (sq * sq + jl * jl - 1.0 - torch.log(sq * sq))).sum(1)
        out.append((jl, jv, sc, z, torch.softmax(z, -1), kl))
    return out


def test_n_group_poe_kernels_match_their_definition(dev):
    from spvipes_amd.nn_ops import PoEComponents
    from spvipes_amd.ops import Workspace
    g = torch.Generator().manual_seed(0)
    Bs, n, n_comp = (40, 33, 50), 6, 5
    loc = [torch.randn(B, n, generator=g, dtype=torch.float64).requires_grad_(True) for B in Bs]
    logvar = [(0.5 * torch.randn(B, n, generator=g, dtype=torch.float64)).requires_grad_(True) for B in Bs]
    comp = [torch.randint(0, 4, (Bs[0],), generator=g).float(), torch.randint(1, 5, (Bs[1],), generator=g).float(),
            torch.randint(0, 3, (Bs[2],), generator=g).float()]   # component 4 only in group 1, component 0 not in group 1, 3 not in group 2
    eps = [torch.randn(B, n, generator=g, dtype=torch.float64) for B in Bs]
    want = _torch_poe_components(loc, logvar, comp, eps, n_comp)
    coef = [[torch.randn(t.shape, generator=g, dtype=torch.float64) for t in w] for w in want]
    sum((t * c).sum() for w, cs in zip(want, coef) for k, (t, c) in enumerate(zip(w, cs)) if k != 4).backward()
    d_loc = [t.detach().float().to(dev).requires_grad_(True) for t in loc]
    d_lv = [t.detach().float().to(dev).requires_grad_(True) for t in logvar]
    flat = [t for pair in zip(d_loc, d_lv) for t in pair]
    o = PoEComponents.apply([c.to(dev) for c in comp], n_comp, [e.float().to(dev) for e in eps], Workspace(dev), *flat)
    sum((o[7 * i + k] * coef[i][k].float().to(dev)).sum() for i in range(3) for k in range(6) if k != 4).backward()
    torch.cuda.synchronize()
    for i in range(3):
        for k, name in enumerate(("loc", "logvar", "scale", "log_z", "theta", "kl")):
            torch.testing.assert_close(o[7 * i + k].detach().cpu().double(), want[i][k].detach(), rtol=2e-5, atol=2e-6, msg=lambda m: f"group {i} {name}: {m}")
        torch.testing.assert_close(d_loc[i].grad.cpu().double(), loc[i].grad, rtol=2e-4, atol=2e-5, msg=lambda m: f"group {i} d loc: {m}")
        torch.testing.assert_close(d_lv[i].grad.cpu().double(), logvar[i].grad, rtol=2e-4, atol=2e-5, msg=lambda m: f"group {i} d logvar: {m}")


def _three_group_trainer(dev, seed=0):
    from spvipes_amd.data import make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    Gs = (300, 260, 340)
    groups = [make_synthetic_group(g % 2, 1024, Gs[g], dev) for g in range(3)]
    torch.manual_seed(seed)
    module = spVIPESmodule({g: Gs[g] for g in range(3)}, transport_plan="components", pair_data=False, allow_more_groups=True, n_components=10,
                           n_hidden=256, n_dimensions_shared=10, n_dimensions_private=5, dropout_rate=0.0, precision="bf16").to(dev)
    trainer = Trainer(module, [g.counts for g in groups], components=[g.labels for g in groups], lr=5e-3)
    module.train()
    return module, trainer


def test_three_group_step_runs_learns_and_is_reproducible(dev):
    from spvipes_amd.data import MinibatchSampler
    from spvipes_amd.module import spVIPESmodule
    with pytest.raises(ValueError, match="only supported value is 2"):
        spVIPESmodule({0: 10, 1: 10, 2: 10})   # the reference's behaviour unless the extension is asked for
    runs = []
    for rep in range(2):
        module, trainer = _three_group_trainer(dev)
        sampler = MinibatchSampler([1024] * 3, 256, dev, seed=0)
        torch.manual_seed(5)
        losses = []
        for ep in range(5):
            for rows in sampler.epoch():
                lo = trainer.step(rows, kl_weight=1.0)
                losses.append(float(lo.loss.detach()))
        torch.cuda.synchronize()
        assert set(lo.reconstruction_loss) == {f"reconst_loss_groups_{g}_poe" for g in (1, 2, 3)} and len(lo.kl_local) == 6
        runs.append((losses, trainer.fp.flat.clone()))
    losses = runs[0][0]
    assert np.isfinite(losses).all() and np.mean(losses[-4:]) < np.mean(losses[:4]) - 1.0, (losses[:4], losses[-4:])
    assert runs[0][0] == runs[1][0] and torch.equal(runs[0][1], runs[1][1]), "two identical runs must agree bit for bit (fixed-order reductions)"
