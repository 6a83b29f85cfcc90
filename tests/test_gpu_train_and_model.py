"""GPU: optimiser kernel, training loop, user API, and size-independent properties at benchmark sizes."""
import numpy as np
import pytest
import torch

from tests import _hip_harness as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    from spvipes_amd import _abi
    _abi.load()
    return torch.device("cuda:0")


def test_hip_adam_matches_torch_adam(dev):
    from spvipes_amd.train import FlatParams, HipAdam
    torch.manual_seed(0)
    mk = lambda: torch.nn.Sequential(torch.nn.Linear(13, 17), torch.nn.Linear(17, 5)).to(dev)
    a, b = mk(), mk()
    b.load_state_dict(a.state_dict())
    fp = FlatParams(a)
    mine = HipAdam(fp, lr=1e-3, eps=0.01, weight_decay=1e-6)
    ref = torch.optim.Adam(b.parameters(), lr=1e-3, eps=0.01, weight_decay=1e-6)
    for step in range(5):
        x = torch.randn(9, 13, device=dev)
        fp.zero_grad()
        a(x).pow(2).sum().backward()
        mine.step()
        ref.zero_grad()
        b(x).pow(2).sum().backward()
        ref.step()
    for pa, pb in zip(a.parameters(), b.parameters()):
        torch.testing.assert_close(pa, pb, rtol=1e-5, atol=1e-7)


from tests._duck import DuckAnnData, make_duck as _duck  # noqa: E402


def test_user_api_train_and_latents(dev):
    from spvipes_amd.model import spVIPES
    ad = _duck()
    spVIPES.setup_anndata(ad, groups_key="groups", label_key="cell_type")
    torch.manual_seed(0)
    model = spVIPES(ad, n_hidden=32, n_dimensions_shared=10, n_dimensions_private=5, precision="fp32")
    gi = [list(ad.uns["groups_obs_indices"][0]), list(ad.uns["groups_obs_indices"][1])]
    model.train(gi, batch_size=64, max_epochs=8, train_size=1.0, n_epochs_kl_warmup=None, plan_kwargs={"lr": 5e-3})
    h = model.history["train_loss"]  # constant KL weight: the loss is the negative ELBO and must go down
    assert len(h) == 8 and all(np.isfinite(h)) and h[-1] < h[0], h
    lat = model.get_latent_representation(gi, batch_size=64)
    assert set(lat) == {"shared", "private", "shared_reordered", "private_reordered"}
    assert lat["shared"][0].shape == (300, 10) and lat["shared"][1].shape == (260, 10)
    assert lat["private"][0].shape == (300, 5) and lat["private_reordered"][1].shape == (260, 5)
    assert all(np.isfinite(v).all() for d in lat.values() for v in d.values())
    load = model.get_loadings()   # DataFrames with the reference's column names (model/spvipes.py:652-677)
    assert load[(0, "shared")].shape == (96, 10) and load[(1, "private")].shape == (80, 5)
    assert list(load[(0, "shared")].columns[:2]) == ["Z_shared_0", "Z_shared_1"] and list(load[(1, "private")].columns) == [f"Z_private_{n}" for n in range(5)]
    np.testing.assert_allclose(load[(1, "private")].to_numpy(), model.module.get_loadings(1, "private"))


def test_user_api_paired_transport_plan_cycling(dev):
    from spvipes_amd.model import spVIPES
    ad = _duck(n=(96, 64), with_plan=True)
    spVIPES.setup_anndata(ad, groups_key="groups", transport_plan_key="plan")
    model = spVIPES(ad, n_hidden=16, n_dimensions_shared=6, n_dimensions_private=3, precision="fp32")
    assert model.module.pair_data and model.module.use_transport_plan and not model.module.use_labels
    gi = [list(ad.uns["groups_obs_indices"][0]), list(ad.uns["groups_obs_indices"][1])]
    model.train(gi, batch_size=32, max_epochs=2, train_size=1.0)
    lat = model.get_latent_representation(gi, batch_size=32)  # unequal groups -> cycling path (:578-626)
    assert lat["shared"][0].shape == (96, 6) and lat["shared"][1].shape == (64, 6)


@pytest.mark.parametrize("mode", ["label", "paired"])
def test_get_latent_representation_values_follow_the_oracle(dev, mode):
    """SURVEY 8 f-2, value level: with the noise pinned to zero (sampled log_z = loc) the four dictionaries returned by
    get_latent_representation must equal what the reference's batch loop produces -- oracle encoders + PoE (eval mode)
    on the batch pairs of oracle/host_semantics.latent_steps (ConcatDataLoader order; cycling chunks for the paired PoE
    with unequal groups), assembled by its restatement of _format_results (model/spvipes.py:424-650)."""
    from oracle import host_semantics as HS
    from oracle import spvipes_oracle as O
    from spvipes_amd.model import spVIPES
    n = (75, 52)
    ad = _duck(n=n, seed=3, with_plan=(mode == "paired"))
    rng = np.random.default_rng(9)
    ad.obs["indices"] = np.concatenate([rng.permutation(n[0]), rng.permutation(n[1])]).astype(np.int32)  # not sorted: the reorder must do work
    if mode == "label":
        spVIPES.setup_anndata(ad, groups_key="groups", label_key="cell_type")
    else:
        spVIPES.setup_anndata(ad, groups_key="groups", transport_plan_key="plan")
    torch.manual_seed(5)
    n_s, n_p, B = 6, 3, 32
    model = spVIPES(ad, n_hidden=16, n_dimensions_shared=n_s, n_dimensions_private=n_p, precision="fp32")
    gi = [list(ad.uns["groups_obs_indices"][0]), list(ad.uns["groups_obs_indices"][1])]
    zeros = lambda step, b0, b1: {**{f"enc_{g}_{k}": torch.zeros(b, d, device=dev) for g, b in ((0, b0), (1, b1)) for k, d in (("private", n_p), ("shared", n_s))},
                                  **{f"poe_{g}": torch.zeros(b, n_s, device=dev) for g, b in ((0, b0), (1, b1))}}
    got = model.get_latent_representation(gi, batch_size=B, _noise=zeros)
    # ---- the reference's loop, restated ----------------------------------------------------------------------
    sd = {k: v.detach().cpu() for k, v in model.module.state_dict().items()}
    var_idx = ad.uns["groups_var_indices"]
    labels = np.unique(ad.obs["cell_type"], return_inverse=True)[1].astype(np.float32)
    res = {k: [] for k in ("groups_1_latent", "groups_2_latent", "groups_1_latent_shared", "groups_2_latent_shared", "groups_2_original_indices")}
    steps = HS.latent_steps(gi, B, False, use_cycling=(mode == "paired"))
    assert len(steps) >= 3
    for r0, r1 in steps:
        rows = [np.asarray(r0), np.asarray(r1)]
        x = [torch.log1p(torch.tensor(ad.X[rows[g]][:, var_idx[g]])) for g in range(2)]
        z = lambda g, d: torch.zeros(len(rows[g]), d)
        priv = [O.encoder_forward(sd, f"encoder_{g}_private", x[g], z(g, n_p), False) for g in range(2)]
        sh = [O.encoder_forward(sd, f"encoder_{g}_shared", x[g], z(g, n_s), False) for g in range(2)]
        if mode == "label":
            p0, p1 = O.poe_label(sh[0], sh[1], torch.tensor(labels[rows[0]]), torch.tensor(labels[rows[1]]))
        else:
            i0, i1 = ad.obs["indices"][rows[0]], ad.obs["indices"][rows[1]]
            p0, p1 = O.poe_paired(sh[0], sh[1], torch.tensor(ad.uns["plan"])[i0][:, i1])
        res["groups_1_latent"].append(priv[0]["log_z"].numpy())
        res["groups_2_latent"].append(priv[1]["log_z"].numpy())
        res["groups_1_latent_shared"].append(O.poe_sample(p0, z(0, n_s), mode == "paired")["logtheta_log_z"].numpy())
        res["groups_2_latent_shared"].append(O.poe_sample(p1, z(1, n_s), mode == "paired")["logtheta_log_z"].numpy())
        res["groups_2_original_indices"].append(ad.obs["indices"][rows[1]][:, None].astype(np.float32))
    want = HS.format_results(res, n[0], n[1])
    for k in ("shared", "private", "shared_reordered", "private_reordered"):
        for g in (0, 1):
            np.testing.assert_allclose(got[k][g], want[k][g], rtol=1e-3, atol=2e-4, err_msg=f"{mode} {k}[{g}]")


def test_validation_metrics_and_early_stopping(dev):
    """train_size < 1 holds cells out as MultiGroupDataSplitter does; validation metrics are logged per epoch and early stopping
    (scvi-tools defaults: monitor elbo_validation, mode min) ends training when they stop improving."""
    from spvipes_amd.model import spVIPES
    ad = _duck(n=(320, 288))
    spVIPES.setup_anndata(ad, groups_key="groups", label_key="cell_type")
    torch.manual_seed(0)
    model = spVIPES(ad, n_hidden=32, n_dimensions_shared=10, n_dimensions_private=5, precision="fp32")
    gi = [list(ad.uns["groups_obs_indices"][0]), list(ad.uns["groups_obs_indices"][1])]
    model.train(gi, batch_size=32, max_epochs=6, train_size=0.75, n_epochs_kl_warmup=None, plan_kwargs={"lr": 5e-3}, check_val_every_n_epoch=1)
    h = model.history
    assert len(h["elbo_validation"]) == 6 and np.isfinite(h["elbo_validation"]).all() and h["train_loss"][-1] < h["train_loss"][0]
    assert abs(h["elbo_validation"][-1] - h["elbo_train"][-1]) / h["elbo_train"][-1] < 0.2   # held-out cells of the same distribution
    assert len(model.sampler_.val_idx[0]) == 80 and len(model.sampler_.train_idx[0]) == 240   # ceil(0.75 * 320) training cells
    # with an unreachable min_delta only the first epoch counts as an improvement: patience 2 stops after epoch index 2
    model2 = spVIPES(ad, n_hidden=32, n_dimensions_shared=10, n_dimensions_private=5, precision="fp32")
    model2.train(gi, batch_size=32, max_epochs=10, train_size=0.75, early_stopping=True, early_stopping_patience=2,
                 early_stopping_min_delta=1e9, n_epochs_kl_warmup=None)
    assert len(model2.history["elbo_validation"]) == 3 and model2.trainer_.stopped_epoch == 2
    with pytest.raises(ValueError, match="early_stopping needs"):
        spVIPES(ad, n_hidden=32, n_dimensions_shared=10, n_dimensions_private=5).train(gi, batch_size=32, max_epochs=1, train_size=1.0, early_stopping=True)


def test_setup_errors_match_reference(dev):
    from spvipes_amd.model import spVIPES
    ad = _duck()
    with pytest.raises(ValueError):
        spVIPES.setup_anndata(ad, groups_key="groups", transport_plan_key="missing")
    with pytest.raises(ValueError):
        spVIPES(ad)  # setup_anndata not run


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_full_size_invariances(dev, precision):
    """BASELINE config-1 sizes (4096 cells x 10000 genes): properties that need no oracle run --
    (a) permuting the cells of the minibatch permutes the per-cell reconstruction terms,
    (b) permuting genes (counts columns and every per-gene parameter row together) leaves them unchanged,
    (c) the weighted loss is linear in the cell weights."""
    from spvipes_amd import ops
    from spvipes_amd.data import make_synthetic_group
    B, G, n_p, n_s = 4096, 10000, 10, 25
    grp = make_synthetic_group(0, B, G, dev)
    nsplit = 3 if precision == "fp32" else 1
    g = torch.Generator(device=dev).manual_seed(0)
    r = lambda *s, sc=1.0: torch.randn(*s, generator=g, device=dev) * sc
    t = dict(zp=r(B, n_p), zs=r(B, n_s), m=torch.relu(r(B, 256)), Wp=r(G, n_p, sc=0.5), cp=r(G, sc=0.3), Ws=r(G, n_s, sc=0.3),
             cs=r(G, sc=0.3), Wm=r(G, 256 + n_p + n_s, sc=0.05), bm=r(G, sc=0.2), px_r=r(G))
    Xf = torch.from_numpy(grp.counts.X.cpu().numpy().view(np.uint16).astype(np.float32)).to(dev)
    lib = torch.log(torch.log1p(Xf).sum(1))
    w = torch.rand(B, generator=g, device=dev) / B
    ws = ops.Workspace(dev)
    rows = torch.arange(B, dtype=torch.int32, device=dev)

    def run(rows, t, lib, w, counts=grp.counts):
        with torch.no_grad():
            loss, rec = H.DecoderNBLoss.apply(counts, rows, B, *t.values(), lib, w, nsplit, False, ws)
        return float(loss), rec.clone()

    base_loss, base_rec = run(rows, t, lib, w)
    assert torch.isfinite(base_rec).all()
    tol = 1e-5 if precision == "fp32" else 2e-4
    perm = torch.randperm(B, generator=g, device=dev)
    tp = dict(t, zp=t["zp"][perm], zs=t["zs"][perm], m=t["m"][perm])
    _, rec_p = run(perm.to(torch.int32), tp, lib[perm], w[perm])
    torch.testing.assert_close(rec_p, base_rec[perm], rtol=tol, atol=tol * 1e3)
    gp = torch.randperm(G, generator=g, device=dev)
    tg = dict(t, Wp=t["Wp"][gp], cp=t["cp"][gp], Ws=t["Ws"][gp], cs=t["cs"][gp], Wm=t["Wm"][gp], bm=t["bm"][gp], px_r=t["px_r"][gp])
    Xg = ops.GroupCounts(grp.counts.X[:, gp.cpu().to(dev)].contiguous(), G, 0)
    _, rec_g = run(rows, tg, lib, w, counts=Xg)
    torch.testing.assert_close(rec_g, base_rec, rtol=max(tol, 5e-5), atol=max(tol, 5e-5) * 1e3)
    l2, _ = run(rows, t, lib, 2.0 * w)
    assert abs(l2 - 2 * base_loss) <= 1e-5 * abs(base_loss)


def test_training_step_runs_at_benchmark_size_and_learns(dev):
    from spvipes_amd.data import MinibatchSampler, make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    torch.manual_seed(0)
    groups = [make_synthetic_group(g, 8192, 2000, dev) for g in range(2)]
    module = spVIPESmodule({0: 2000, 1: 2000}, use_labels=True, n_hidden=64, n_dimensions_shared=10, n_dimensions_private=5).to(dev)
    trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups], lr=5e-3)
    sampler = MinibatchSampler([8192, 8192], 1024, dev, seed=0)
    module.train()
    losses = []
    for ep in range(6):
        for rows in sampler.epoch():
            losses.append(float(trainer.step(rows, kl_weight=1.0).loss.detach()))
    assert np.isfinite(losses).all() and np.mean(losses[-8:]) < np.mean(losses[:8]) - 1.0, (losses[:8], losses[-8:])


def test_graph_replay_matches_eager_steps(dev):
    """The captured hipGraph step must produce what the eager step produces on the same minibatches (dropout off,
    injected-free noise is drawn from the same generator state), and must keep learning."""
    from spvipes_amd.data import MinibatchSampler, make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    groups = [make_synthetic_group(g, 4096, 1000, dev) for g in range(2)]

    def run(use_graph):
        torch.manual_seed(0)
        module = spVIPESmodule({0: 1000, 1: 1000}, use_labels=True, n_hidden=64, n_dimensions_shared=10, n_dimensions_private=5,
                               dropout_rate=0.0, precision="fp32").to(dev)
        trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups], lr=5e-3)
        sampler = MinibatchSampler([4096, 4096], 512, dev, seed=0)
        module.train()
        batches = [rows for _ in range(3) for rows in sampler.epoch()]
        if use_graph:
            trainer.capture(batches[0])
        torch.manual_seed(123)  # same noise stream from here on in both runs
        return [float(trainer.step(rows, kl_weight=1.0).loss.detach()) for rows in batches]

    eager, graph = run(False), run(True)
    assert np.isfinite(graph).all() and np.mean(graph[-4:]) < np.mean(graph[:4])
    # capture warm-up steps move the parameters of the graph run a little before step 0: compare trends, not bits
    assert abs(np.mean(graph[-4:]) - np.mean(eager[-4:])) / abs(np.mean(eager[-4:])) < 5e-2


def test_trainer_rejects_label_codes_the_pairing_kernel_cannot_index(dev):
    from spvipes_amd.data import make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    groups = [make_synthetic_group(g, 300, 64, dev) for g in range(2)]
    module = spVIPESmodule({0: 64, 1: 64}, use_labels=True, n_hidden=16, n_dimensions_shared=4, n_dimensions_private=2).to(dev)
    for bad in (torch.full((300,), 1024.0, device=dev), torch.full((300,), 2.5, device=dev), torch.full((300,), -1.0, device=dev)):
        with pytest.raises(ValueError, match="integral codes"):
            Trainer(module, [g.counts for g in groups], labels=[groups[0].labels, bad])


def test_five_optimiser_steps_follow_the_oracle_trajectory(dev):
    """End to end over several steps: HIP forward/backward + spv_adam_step against the CPU oracle + torch.optim.Adam
    (scvi TrainingPlan's settings: lr 1e-3, eps 0.01, weight_decay 1e-6) on the same minibatches and noise -- the loss of
    every step and the parameters after the last one."""
    from oracle import spvipes_oracle as O
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import FlatParams, HipAdam
    B, Gs, H, n_s, n_p, steps = 48, (70, 55), 16, 6, 3, 5
    rng = np.random.default_rng(11)
    torch.manual_seed(3)
    module = spVIPESmodule({0: Gs[0], 1: Gs[1]}, use_labels=True, n_hidden=H, n_dimensions_shared=n_s, n_dimensions_private=n_p,
                           dropout_rate=0.0, precision="fp32").to(dev)
    module.train()
    sd = {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
    pnames = [k for k, _ in module.named_parameters()]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in pnames}
    ref_opt = torch.optim.Adam([leaves[k] for k in pnames], lr=1e-3, eps=0.01, weight_decay=1e-6)
    fp = FlatParams(module)
    opt = HipAdam(fp, lr=1e-3, eps=0.01, weight_decay=1e-6)
    gen = torch.Generator().manual_seed(8)
    for step in range(steps):
        counts = [(rng.poisson(2.5, size=(B, G)) * (rng.random((B, G)) < 0.4)).astype(np.float32) for G in Gs]
        for c in counts:
            c[:, 0] += 1
        labels = [rng.integers(0, 3, size=B).astype(np.float32), rng.integers(1, 4, size=B).astype(np.float32)]
        noise = {f"enc_{g}_{k}": torch.randn(B, n, generator=gen) for g in range(2) for k, n in (("private", n_p), ("shared", n_s))}
        noise.update({f"poe_{g}": torch.randn(B, n_s, generator=gen) for g in range(2)})
        tensors = []
        for g in range(2):
            X = np.zeros((B, sum(Gs)), np.float32)
            X[:, (0 if g == 0 else Gs[0]):(Gs[0] if g == 0 else sum(Gs))] = counts[g]
            tensors.append({"X": torch.tensor(X).to(dev), "batch": torch.zeros(B, 1, device=dev), "groups": torch.full((B, 1), float(g), device=dev),
                            "indices": torch.arange(B, dtype=torch.float32, device=dev).unsqueeze(1), "labels": torch.tensor(labels[g], device=dev).unsqueeze(1)})
        fp.zero_grad()
        _, _, lo = module(tuple(tensors), inference_kwargs={"noise": {k: v.to(dev) for k, v in noise.items()}}, loss_kwargs={"kl_weight": 0.5})
        lo.loss.backward()
        opt.step()
        # oracle step (its BatchNorm running statistics are not needed in training mode)
        sd_ref = dict(sd)
        sd_ref.update(leaves)
        ref_opt.zero_grad()
        out = O.forward_loss(sd_ref, [torch.tensor(c) for c in counts], n_dimensions_shared=n_s, n_dimensions_private=n_p, noise=noise, mode="label",
                             labels=[torch.tensor(l) for l in labels], training=True, kl_weight=0.5)
        out["loss"].backward()
        ref_opt.step()
        got, want = float(lo.loss.detach()), float(out["loss"].detach())
        assert abs(got - want) / abs(want) < 2e-4, (step, got, want)
    torch.cuda.synchronize()
    for k, p in module.named_parameters():
        torch.testing.assert_close(p.detach().cpu(), leaves[k].detach(), rtol=2e-3, atol=2e-5, msg=lambda m: f"{k} after {steps} steps: {m}")


def test_adam_written_weight_images_are_bit_identical_to_repacking(dev):
    """bf16 mode: the optimiser rewrites the packed bf16 weight images itself (spv_adam_step_images) and the next forward pass
    skips its spv_pack_bf16 launches.  Same minibatches and noise, feature on / off: every loss, every parameter and the images
    themselves must agree BITWISE (same rounding), eagerly and under hipGraph replay; a torch-side write to a parameter must be
    noticed (version counter) and repacked."""
    from spvipes_amd.data import MinibatchSampler, make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    groups = [make_synthetic_group(g, 2048, 777, dev) for g in range(2)]

    def run(by_adam, use_graph, poke):
        torch.manual_seed(0)
        module = spVIPESmodule({0: 777, 1: 777}, use_labels=True, n_hidden=128, n_dimensions_shared=10, n_dimensions_private=5,
                               dropout_rate=0.0, precision="bf16").to(dev)
        Trainer.IMAGES_BY_ADAM = by_adam
        try:
            trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups], lr=5e-3)
            sampler = MinibatchSampler([2048, 2048], 256, dev, seed=0)
            module.train()
            batches = [rows for _ in range(2) for rows in sampler.epoch()]
            if use_graph:
                trainer.capture(batches[0], warmup=1)
            torch.manual_seed(123)
            losses = []
            for i, rows in enumerate(batches):
                if poke and i == 5:   # somebody writes to parameters between two steps
                    with torch.no_grad():
                        module.encoders[0]["shared"].fc1.weight.mul_(1.25)
                        module.decoders[1].mixture.linear.bias.add_(0.5)
                losses.append(float(trainer.step(rows, kl_weight=1.0).loss.detach()))
            imgs = [img.clone() for _ws, _k, img, _p, _t in trainer._image_specs()]
            assert (trainer._img_plan is not None) == bool(by_adam)
            if by_adam:
                assert trainer._images_are_fresh()
            return losses, trainer.fp.flat.clone(), imgs
        finally:
            Trainer.IMAGES_BY_ADAM = True

    for use_graph in (False, True):
        for poke in (False, True):
            a, b = run(True, use_graph, poke), run(False, use_graph, poke)
            assert a[0] == b[0], (use_graph, poke, a[0][:8], b[0][:8])
            assert torch.equal(a[1], b[1])
    # images == pack(parameters) after an Adam-with-images step
    torch.manual_seed(0)
    module = spVIPESmodule({0: 777, 1: 777}, use_labels=True, n_hidden=128, n_dimensions_shared=10, n_dimensions_private=5, dropout_rate=0.0).to(dev)
    trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups], lr=5e-3)
    sampler = MinibatchSampler([2048, 2048], 256, dev, seed=0)
    module.train()
    for rows in list(sampler.epoch())[:3]:
        trainer.step(rows, kl_weight=1.0)
    by_adam = [img.clone() for _ws, _k, img, _p, _t in trainer._img_specs]
    trainer.parameters_changed()
    assert not trainer._images_are_fresh()
    trainer._ensure_images()    # explicit repack from the same parameters
    for x, (_ws, key, img, _p, _t) in zip(by_adam, trainer._img_specs):
        assert torch.equal(x, img), key


@pytest.mark.parametrize("mode,precision,H,overlap,graph", [
    ("label", "bf16", 128, False, False), ("label", "fp32", 128, False, False), ("cluster", "bf16", 128, False, False), ("paired", "fp32", 128, False, False),
    ("label", "bf16", 256, False, False), ("label", "bf16", 128, True, True), ("label", "bf16", 256, True, True), ("paired", "fp32", 128, True, False),
    ("three", "bf16", 256, False, False), ("three", "bf16", 256, True, True), ("three", "fp32", 128, True, True)])
def test_every_gradient_element_is_overwritten_by_a_step(dev, mode, precision, H, overlap, graph):
    """train.Trainer does not zero the flat gradient buffer at the start of a step (ZERO_GRADS_EACH_STEP off): that is only right
    while every gradient kernel overwrites.  Fill the buffer with NaN, run forward + backward, and no parameter may hold a NaN --
    for every PoE mode, both precisions, n_hidden 128 / 256 (one / two 256-unit halves in the fc1 kernels), the three-group path
    (PoEComponents, decoder chunks of <= 2 groups), the split backward pass of a data-parallel job (``overlap``) and its two
    captured graphs (``graph``): an accumulate-or-skip kernel anywhere would feed stale gradients to Adam."""
    import scipy.sparse as sp
    from spvipes_amd.data import MinibatchSampler, make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    NG = 3 if mode == "three" else 2
    Gs = (300, 260, 340)[:NG] if NG == 3 else (300, 300)
    groups = [make_synthetic_group(g % 2, 1024, Gs[g], dev) for g in range(NG)]
    torch.manual_seed(0)
    kw = dict(n_hidden=H, n_dimensions_shared=10, n_dimensions_private=5, precision=precision)
    if mode == "label":
        module = spVIPESmodule({0: 300, 1: 300}, use_labels=True, **kw).to(dev)
        trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups], overlap_allreduce=overlap)
    elif mode == "three":
        module = spVIPESmodule({g: Gs[g] for g in range(3)}, transport_plan="components", pair_data=False, allow_more_groups=True, n_components=10, **kw).to(dev)
        trainer = Trainer(module, [g.counts for g in groups], components=[g.labels for g in groups], overlap_allreduce=overlap)
    else:
        rng = np.random.default_rng(0)
        n = 1024
        i, j, v = np.repeat(np.arange(n), 3), rng.integers(0, n, size=3 * n), rng.random(3 * n).astype(np.float32) + 0.1
        P = sp.coo_matrix((v, (i, j)), shape=(n, n)).tocsr()
        module = spVIPESmodule({0: 300, 1: 300}, use_labels=False, transport_plan=P, pair_data=(mode == "paired"), **kw).to(dev)
        trainer = Trainer(module, [g.counts for g in groups], components=[g.labels for g in groups] if mode == "cluster" else None, overlap_allreduce=overlap)
    assert not trainer.ZERO_GRADS_EACH_STEP
    sampler = MinibatchSampler([1024] * NG, 256, dev, seed=0)
    module.train()
    rows = next(iter(sampler.epoch()))
    if graph:
        trainer.capture(rows, warmup=1)
    trainer.step(rows, kl_weight=1.0)
    trainer.fp.grad.fill_(float("nan"))
    trainer.step(rows, kl_weight=1.0, optimizer_step=False)
    torch.cuda.synchronize()
    bad = [(n_, int(torch.isnan(p.grad).sum())) for n_, p in module.named_parameters() if p.grad is not None and bool(torch.isnan(p.grad).any())]
    assert not bad, bad


# ---- the DEVICE sampler against the oracle that the reference-run loader fixtures pin (tests/golden/host_*.npz) ----
@pytest.mark.parametrize("sizes,B,train_size,val_size,seed", [((40, 33), 4, 0.9, None, 0), ((250, 31), 16, 0.5, 0.25, 11), ((5000, 1800), 128, 0.9, None, 3)])
def test_device_sampler_split_and_epoch_order_follow_the_oracle(dev, sizes, B, train_size, val_size, seed):
    from oracle import host_semantics as HS
    from spvipes_amd.data import MinibatchSampler
    gi = [np.arange(100, 100 + sizes[0]), np.arange(10_000, 10_000 + sizes[1])]
    s = MinibatchSampler(list(sizes), B, dev, seed=seed, train_size=train_size, validation_size=val_size, group_indices_list=gi)
    want = HS.split_groups(gi, train_size, val_size, seed)          # data/_multi_datasplitter.py:65-79
    for g in range(2):
        assert s.train_idx[g].tolist() == want["train"][g].tolist() and s.val_idx[g].tolist() == want["val"][g].tolist()
        assert s._train_dev[g].device.type == "cuda" and s._train_dev[g].cpu().tolist() == want["train"][g].tolist()
    firsts = []
    for ep in range(2):
        perms = s.draw_permutations()                                # one device randperm per group and epoch
        assert all(p.device.type == "cuda" and p.dtype == torch.int32 for p in perms)
        orders = []
        for g in range(2):
            pos = {int(c): i for i, c in enumerate(want["train"][g])}
            orders.append(np.asarray([pos[int(c)] for c in perms[g].cpu().tolist()]))
            assert sorted(orders[g].tolist()) == list(range(len(want["train"][g])))      # a permutation of the training rows
        ref_steps = HS.concat_loader_steps(want["train"], B, drop_last=True, orders=orders)   # dataloaders/_concat_dataloader.py:101-110
        got = list(s.epoch_from_permutations(perms))
        assert len(got) == len(ref_steps) == s.steps_per_epoch == max(len(t) // B for t in want["train"])
        for a, b in zip(got, ref_steps):
            for g in range(2):
                assert a[g].device.type == "cuda" and a[g].is_contiguous() and a[g].cpu().tolist() == b[g].tolist()
        firsts.append(got[0][0].cpu().tolist())
    assert firsts[0] != firsts[1]                                    # reshuffled every epoch
    ep = list(s.epoch())                                             # the generator the trainer consumes: same shape of epoch
    assert len(ep) == s.steps_per_epoch and all(r.numel() == B for st in ep for r in st)


# ---- save(dir) / load(dir, adata=...) as the tutorial calls them (Tutorial.ipynb:487,538) ---------------------------
@pytest.mark.parametrize("mode", ["label", "paired"])
def test_save_and_load_round_trip(dev, tmp_path, mode):
    from spvipes_amd.model import spVIPES
    ad = _duck(with_plan=(mode == "paired"), n=(300, 300) if mode == "paired" else (300, 260))
    kw = {"label_key": "cell_type"} if mode == "label" else {"transport_plan_key": "plan"}
    spVIPES.setup_anndata(ad, groups_key="groups", **kw)
    torch.manual_seed(0)
    model = spVIPES(ad, n_hidden=32, n_dimensions_shared=10, n_dimensions_private=5, dropout_rate=0.05, precision="fp32")
    gi = [list(ad.uns["groups_obs_indices"][0]), list(ad.uns["groups_obs_indices"][1])]
    model.train(gi, batch_size=64, max_epochs=3, train_size=1.0)     # moves parameters AND BatchNorm running statistics
    d = str(tmp_path / "spvipes_model")
    model.save(d)
    with pytest.raises(ValueError, match="already exists"):
        model.save(d)
    model.save(d, overwrite=True)
    blob = torch.load(d + "/model.pt", map_location="cpu", weights_only=False)
    assert set(blob) == {"model_state_dict", "var_names", "attr_dict"}
    assert any(k.endswith("running_var") for k in blob["model_state_dict"]) and "px_r.0" in blob["model_state_dict"]
    # a fresh AnnData object that was never set up: load() registers it with the saved setup arguments
    ad2 = _duck(with_plan=(mode == "paired"), n=(300, 300) if mode == "paired" else (300, 260))
    back = spVIPES.load(d, adata=ad2)
    assert back.is_trained_ and not back.module.training and back.history["train_loss"] == model.history["train_loss"]
    assert back.init_params_ == model.init_params_ and back.module.dropout_rate == 0.05
    sd_a, sd_b = model.module.state_dict(), back.module.state_dict()
    assert list(sd_a) == list(sd_b)
    for k in sd_a:
        assert torch.equal(sd_a[k], sd_b[k]), k                       # parameters, BN running statistics, num_batches_tracked
    la, lb = model.get_loadings(), back.get_loadings()
    for k in la:
        np.testing.assert_array_equal(la[k].to_numpy(), lb[k].to_numpy())
    # the same latents from both models given the same injected draws
    def noise(step, b0, b1):
        g = torch.Generator(device="cpu").manual_seed(step)
        mk = lambda b, n: torch.randn(b, n, generator=g).to(dev)
        return {"enc_0_private": mk(b0, 5), "enc_0_shared": mk(b0, 10), "enc_1_private": mk(b1, 5), "enc_1_shared": mk(b1, 10), "poe_0": mk(b0, 10), "poe_1": mk(b1, 10)}
    za = model.get_latent_representation(gi, batch_size=128, _noise=noise)
    zb = back.get_latent_representation(gi, batch_size=128, _noise=noise)
    for k in za:
        for g in (0, 1):
            np.testing.assert_array_equal(za[k][g], zb[k][g])
    with pytest.raises(ValueError, match="no adata was passed"):
        spVIPES.load(d)
    with pytest.raises(ValueError, match="Failed to load model file"):
        spVIPES.load(str(tmp_path / "nowhere"), adata=ad2)
    with pytest.raises(NotImplementedError):
        model.save(str(tmp_path / "with_adata"), save_anndata=True)
