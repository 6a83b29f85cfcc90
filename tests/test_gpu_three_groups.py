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


def test_cells_without_a_valid_component_code_get_no_expert(dev):
    """codes outside [0, n_components) -- negative, too large, NaN -- must not index the component tables: such a cell
    joins no component mean and is fused with the prior expert alone"""
    from spvipes_amd.nn_ops import PoEComponents
    from spvipes_amd.ops import Workspace
    g = torch.Generator().manual_seed(3)
    Bs, n, n_comp = (40, 33, 25), 6, 4
    loc = [torch.randn(B, n, generator=g, dtype=torch.float64) for B in Bs]
    logvar = [torch.randn(B, n, generator=g, dtype=torch.float64) * 0.3 for B in Bs]
    comp = [torch.randint(0, n_comp, (B,), generator=g).float() for B in Bs]
    eps = [torch.randn(B, n, generator=g, dtype=torch.float64) for B in Bs]
    bad = [c.clone() for c in comp]
    bad[0][[0, 5, 9]] = torch.tensor([-1.0, float(n_comp), float("nan")])
    bad[2][[1]] = torch.tensor([1000.0])
    # reference: the same problem with the bad cells removed (they are in no component), their own posteriors fused with the prior only
    keep = [torch.ones(B, dtype=torch.bool) for B in Bs]
    keep[0][[0, 5, 9]] = False
    keep[2][[1]] = False
    want = _torch_poe_components([l[k] for l, k in zip(loc, keep)], [l[k] for l, k in zip(logvar, keep)], [c[k] for c, k in zip(comp, keep)],
                                 [e[k] for e, k in zip(eps, keep)], n_comp)
    flat = [t.float().to(dev) for pair in zip(loc, logvar) for t in pair]
    o = PoEComponents.apply([c.to(dev) for c in bad], n_comp, [e.float().to(dev) for e in eps], Workspace(dev), *flat)
    torch.cuda.synchronize()
    for i in range(3):
        got_loc, got_lv = o[7 * i].cpu().double(), o[7 * i + 1].cpu().double()
        torch.testing.assert_close(got_loc[keep[i]], want[i][0], rtol=2e-5, atol=2e-6)
        torch.testing.assert_close(got_lv[keep[i]], want[i][1], rtol=2e-5, atol=2e-6)
        out = ~keep[i]
        if bool(out.any()):   # prior expert only: prec = 1 + 1 / v, loc* = (mu / v) / prec
            inv = torch.exp(-logvar[i][out])
            torch.testing.assert_close(got_loc[out], loc[i][out] * inv / (1 + inv), rtol=2e-5, atol=2e-6)
            torch.testing.assert_close(got_lv[out], -torch.log(1 + inv), rtol=2e-5, atol=2e-6)
    assert all(bool(torch.isfinite(t).all()) for t in o if t is not None and t.is_floating_point())


# ---- BASELINE configs[3] at its size: 3 groups x 15 000 genes, n_hidden 256, B 4096, cluster-matched PoE ---------------------------------
# The reference cannot run three groups (data/prepare_adatas.py:94-95), so there is no oracle: the full-size step is held to
# properties -- finite, bit-reproducible, equivariant under a permutation of the minibatch's cells, and, when the third group
# shares no component with the first two, equal for groups 0 / 1 to the two-group N-expert step on the same parameters.
C4 = dict(G=15_000, H=256, B=4096, n_cells=6000, n_s=25, n_p=10, n_comp=12)


@pytest.fixture(scope="module")
def c4_groups(dev):
    from spvipes_amd.data import make_synthetic_group
    return [make_synthetic_group(g % 2 if g < 2 else 0, C4["n_cells"], C4["G"], dev) for g in range(3)]


def _c4_trainer(dev, groups, comps, seed=0, state=None):
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    torch.manual_seed(seed)
    NG = len(groups)
    module = spVIPESmodule({g: C4["G"] for g in range(NG)}, transport_plan="components", pair_data=False, allow_more_groups=True,
                           n_components=C4["n_comp"], n_hidden=C4["H"], n_dimensions_shared=C4["n_s"], n_dimensions_private=C4["n_p"],
                           dropout_rate=0.0, precision="bf16").to(dev)
    if state is not None:
        module.load_state_dict(state, strict=True)
    trainer = Trainer(module, [g.counts for g in groups], components=comps)
    module.train()
    return module, trainer


def _c4_noise(NG, B, seed=11):
    gen = torch.Generator().manual_seed(seed)
    noise = {}
    for g in range(NG):
        noise[f"enc_{g}_private"] = torch.randn(B, C4["n_p"], generator=gen)
        noise[f"enc_{g}_shared"] = torch.randn(B, C4["n_s"], generator=gen)
        noise[f"poe_{g}"] = torch.randn(B, C4["n_s"], generator=gen)
    return noise


def _c4_step(dev, trainer, rows, noise):
    lo = trainer.step(rows, kl_weight=1.0, noise={k: v.to(dev) for k, v in noise.items()}, optimizer_step=False)
    torch.cuda.synchronize()
    rec = [v.detach().clone() for v in lo.reconstruction_loss.values()]
    kl = [v.detach().clone() for v in lo.kl_local.values()]
    return float(lo.loss.detach()), rec, kl, trainer.fp.grad.clone()


def test_c4_full_size_step_is_finite_reproducible_and_cell_permutation_equivariant(dev, c4_groups):
    B = C4["B"]
    comps = [g.labels for g in c4_groups]
    rng = np.random.default_rng(2)
    rows = [torch.tensor(rng.permutation(C4["n_cells"])[:B].astype(np.int32), device=dev) for _ in range(3)]
    noise = _c4_noise(3, B)
    module, trainer = _c4_trainer(dev, c4_groups, comps)
    sd = {k: v.detach().clone() for k, v in module.state_dict().items()}
    loss, rec, kl, grad = _c4_step(dev, trainer, rows, noise)
    assert np.isfinite(loss) and all(bool(torch.isfinite(t).all()) for t in rec + kl) and bool(torch.isfinite(grad).all())
    assert len(rec) == 3 and len(kl) == 6 and all(t.shape == (B,) for t in rec + kl)
    assert float(grad.abs().max()) > 0
    # the same step again from the same state: bit for bit
    module2, trainer2 = _c4_trainer(dev, c4_groups, comps, state=sd)
    loss2, rec2, kl2, grad2 = _c4_step(dev, trainer2, rows, noise)
    assert loss2 == loss and all(torch.equal(a, b) for a, b in zip(rec + kl, rec2 + kl2)) and torch.equal(grad, grad2)
    # permute the cells of every group's minibatch (and their noise rows): per-cell terms follow the permutation, the loss and
    # the gradients stay (up to fp32 summation order inside the batch statistics / component means)
    perms = [torch.tensor(rng.permutation(B), device=dev) for _ in range(3)]
    rows_p = [r[p].contiguous() for r, p in zip(rows, perms)]
    noise_p = {}
    for g in range(3):
        for k in (f"enc_{g}_private", f"enc_{g}_shared", f"poe_{g}"):
            noise_p[k] = noise[k][perms[g].cpu()]
    module3, trainer3 = _c4_trainer(dev, c4_groups, comps, state=sd)
    loss3, rec3, kl3, grad3 = _c4_step(dev, trainer3, rows_p, noise_p)
    assert abs(loss3 - loss) <= 2e-6 * abs(loss)
    for g in range(3):
        torch.testing.assert_close(rec3[g], rec[g][perms[g]], rtol=2e-5, atol=2e-3)
    for i in range(6):
        torch.testing.assert_close(kl3[i], kl[i][perms[i // 2]], rtol=1e-4, atol=1e-4)
    scale = float(grad.abs().max())
    assert float((grad3 - grad).abs().max()) <= 2e-3 * scale   # bf16 gradient arrays: a rounding flip per element is ~4e-3 of ITS value


def test_c4_with_a_disjoint_third_group_equals_the_two_group_n_expert_step(dev, c4_groups):
    """group 2's component codes occur in no other group: it contributes no expert, so everything of groups 0 / 1 -- per-cell
    reconstruction and KL terms, parameter gradients -- must equal the TWO-group N-expert step on the same parameters, rows and noise"""
    B = C4["B"]
    comps3 = [c4_groups[0].labels, c4_groups[1].labels, 10.0 + (c4_groups[2].labels % 2)]   # {10, 11}: disjoint from 0..9
    rng = np.random.default_rng(4)
    rows = [torch.tensor(rng.permutation(C4["n_cells"])[:B].astype(np.int32), device=dev) for _ in range(3)]
    noise = _c4_noise(3, B)
    m3, t3 = _c4_trainer(dev, c4_groups, comps3)
    loss3, rec3, kl3, _ = _c4_step(dev, t3, rows, noise)
    g3 = {k: p.grad.detach().clone() for k, p in m3.named_parameters()}
    sd2 = {k: v.detach().clone() for k, v in m3.state_dict().items() if not any(s in k for s in ("encoder_2_", "decoder_2", "px_r.2"))}
    m2, t2 = _c4_trainer(dev, c4_groups[:2], comps3[:2], state=sd2)
    loss2, rec2, kl2, _ = _c4_step(dev, t2, rows[:2], {k: v for k, v in noise.items() if not k.endswith("_2") and "_2_" not in k})
    for g in range(2):
        assert torch.equal(rec3[g], rec2[g]), g
    for i in range(4):
        assert torch.equal(kl3[i], kl2[i]), i
    for k, p in m2.named_parameters():
        # (the loss is the batch mean over the same B cells in both runs, so the factors agree)
        torch.testing.assert_close(g3[k], p.grad, rtol=0, atol=0, msg=lambda m: f"{k}: {m}")
