"""GPU: batch covariates (a module built with n_batch > 1: the reference appends one_hot(batch) to the input of both encoders'
fc1 and of all four decoder layers -- nn/networks.py:60-68,105-119,314-325, module/spVIPESmodule.py:132-133,440-445,751-757).
The reference-run goldens of this feature (tests/golden/*_batch*.npz) are covered by test_gpu_parity.py's golden tests; here the
training path proper: resident count matrices, the trainer's minibatches, LDS-DMA fc1 shapes, hipGraph replay, the gradient sink."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    return torch.device("cuda:0")


def _setup(dev, precision, B, Gs, H, n_s, n_p, nb, cells=1024, seed=0, overlap=None):
    from spvipes_amd.data import make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    groups = [make_synthetic_group(g, cells, Gs[g], dev) for g in range(2)]
    codes = [torch.tensor(rng.integers(0, nb, size=cells), dtype=torch.int32, device=dev) for _ in range(2)]
    module = spVIPESmodule({0: Gs[0], 1: Gs[1]}, use_labels=True, n_batch=nb, n_hidden=H, n_dimensions_shared=n_s, n_dimensions_private=n_p,
                           dropout_rate=0.0, precision=precision).to(dev)
    with torch.no_grad():   # covariate weights large enough to matter
        for g in range(2):
            for kind in ("private", "shared"):
                module.encoders[g][kind].fc1.weight[:, Gs[g]:].normal_(0.0, 0.3)
    module.train()
    trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups], batch_codes=codes, lr=1e-3, overlap_allreduce=overlap)
    rows = [torch.tensor(rng.permutation(cells)[:B], dtype=torch.int32, device=dev) for _ in range(2)]
    gen = torch.Generator().manual_seed(seed + 1)
    noise = {f"enc_{g}_{k}": torch.randn(B, n, generator=gen) for g in range(2) for k, n in (("private", n_p), ("shared", n_s))}
    noise.update({f"poe_{g}": torch.randn(B, n_s, generator=gen) for g in range(2)})
    return module, trainer, groups, codes, rows, noise


def _oracle(module, groups, codes, rows, noise, n_s, n_p, nb, kl_weight, dtype=torch.float64):
    from oracle import spvipes_oracle as O
    sd = {k: (v.detach().cpu().to(dtype) if v.is_floating_point() else v.detach().cpu()) for k, v in module.state_dict().items()}
    for k in O.param_names(sd):
        sd[k].requires_grad_(True)
    counts = [torch.from_numpy(groups[g].counts.X[rows[g].long()].cpu().numpy().view(np.uint16).astype(np.float64)).to(dtype) for g in range(2)]
    out = O.forward_loss(sd, counts, n_dimensions_shared=n_s, n_dimensions_private=n_p, noise={k: v.to(dtype) for k, v in noise.items()},
                         mode="label", labels=[groups[g].labels.cpu()[rows[g].cpu().long()] for g in range(2)], training=True, kl_weight=kl_weight,
                         n_batch=nb, batch_index=[codes[g].cpu()[rows[g].cpu().long()] for g in range(2)])
    out["loss"].backward()
    return sd, out


@pytest.mark.parametrize("precision,B,Gs,H", [("fp32", 96, (130, 111), 16), ("fp32", 256, (700, 650), 128), ("bf16", 256, (700, 650), 128)])
def test_training_step_with_batch_covariates_matches_the_oracle(dev, precision, B, Gs, H):
    n_s, n_p, nb = 25, 10, 3
    module, trainer, groups, codes, rows, noise = _setup(dev, precision, B, Gs, H, n_s, n_p, nb)
    trainer.fp.grad.fill_(float("nan"))   # every gradient element, covariate columns included, must be overwritten by the step
    lo = trainer._forward_backward(rows, torch.full((), 0.7, device=dev), noise={k: v.to(dev) for k, v in noise.items()})
    inf, _gen = trainer.last_outputs
    torch.cuda.synchronize()
    sd, want = _oracle(module, groups, codes, rows, noise, n_s, n_p, nb, 0.7)
    for k, p in module.named_parameters():   # (the flat buffer's alignment padding between parameters is nobody's to write)
        assert bool(torch.isfinite(p.grad).all()), f"gradient of {k} was not (fully) written"
    ltol, lat_tol, gtol = (2e-5, 2e-4, 5e-3) if precision == "fp32" else (1e-3, 5e-3, 6e-2)
    assert abs(float(lo.loss.detach()) - float(want["loss"].detach())) / abs(float(want["loss"].detach())) < ltol
    for g in range(2):
        for kind, key in (("private", "private_stats"), ("shared", "shared_stats")):
            a, b = inf[key][g]["logtheta_loc"].detach().cpu().double(), want[key][g]["logtheta_loc"].detach()
            assert float((a - b).abs().max()) < lat_tol * max(1.0, float(b.abs().max())), (kind, g)
    gmax = max(float(sd[k].grad.abs().max()) for k, _ in module.named_parameters() if sd[k].grad is not None)
    for k, p in module.named_parameters():
        ref = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        mine = p.grad.detach().cpu().double()
        if precision == "fp32":
            err = float((mine - ref).abs().max())
            assert err < gtol * float(ref.abs().max()) + gtol * 2e-2 * gmax, f"{precision} grad {k}: {err:.3e} (max {float(ref.abs().max()):.3e})"
        else:   # 16-bit operand mode: relative L2 per parameter, the measure and the bound of test_gpu_fullsize_parity.py (biases of a
            #      layer in front of a train-mode BatchNorm have an analytically zero gradient: absolute floor)
            err, nrm = float((mine - ref).norm()), float(ref.norm())
            assert err < 8e-2 * nrm + 1e-3 * gmax, f"{precision} grad {k}: L2 err {err:.3e} of {nrm:.3e}"
    # the covariate columns themselves carry signal
    for g in range(2):
        w = module.encoders[g]["shared"].fc1.weight
        assert float(w.grad[:, Gs[g]:].abs().max()) > 0
        assert float(module.decoders[g].mixture.linear.weight.grad[:, -nb:].abs().max()) > 0


def test_loss_value_with_batch_covariates_matches_the_oracle(dev):
    n_s, n_p, nb = 25, 10, 4
    B, Gs, H = 128, (300, 280), 32
    module, trainer, groups, codes, rows, noise = _setup(dev, "fp32", B, Gs, H, n_s, n_p, nb, seed=3)
    _inf, _gen, lo = module(trainer.minibatch(rows), inference_kwargs={"noise": {k: v.to(dev) for k, v in noise.items()}}, loss_kwargs={"kl_weight": 1.0})
    _sd, want = _oracle(module, groups, codes, rows, noise, n_s, n_p, nb, 1.0)
    assert abs(float(lo.loss) - float(want["loss"])) / abs(float(want["loss"])) < 2e-5
    torch.testing.assert_close(lo.reconstruction_loss["reconst_loss_groups_1_poe"].cpu().double(), want["reconstruction_loss"][0].detach(), rtol=5e-5, atol=5e-3)
    # the batch code changes the result: with every code set to 0 the loss moves
    zero = [torch.zeros_like(c) for c in codes]
    trainer.batch_codes = zero
    _i, _g, lo0 = module(trainer.minibatch(rows), inference_kwargs={"noise": {k: v.to(dev) for k, v in noise.items()}}, loss_kwargs={"kl_weight": 1.0})
    assert abs(float(lo0.loss) - float(lo.loss)) > 1e-3 * abs(float(lo.loss))


def test_graph_replay_equals_eager_steps_with_batch_covariates(dev):
    """the captured step (torch-side concatenations, the covariate gradient copies, per-step weight packs) replays bit for bit"""
    from spvipes_amd.data import MinibatchSampler
    n_s, n_p, nb = 25, 10, 2
    B, Gs, H = 256, (640, 640), 128

    def run(use_graph):
        module, trainer, groups, codes, rows, noise = _setup(dev, "bf16", B, Gs, H, n_s, n_p, nb, cells=1024, seed=5)
        sampler = MinibatchSampler([1024, 1024], B, dev, seed=1)
        batches = [r for _ in range(2) for r in sampler.epoch()]
        if use_graph:
            trainer.capture(batches[0], warmup=1)
        losses = [float(trainer.step(r, kl_weight=1.0).loss.detach()) for r in batches]
        return losses, trainer.fp.flat.clone()

    (la, fa), (lb, fb) = run(False), run(True)
    assert la == lb
    assert bool((fa == fb).all())
    assert la[-1] < la[0]


@pytest.mark.parametrize("graph", [False, True])
def test_split_backward_with_batch_covariates_overwrites_and_equals_the_single_pass(dev, graph):
    """the split backward pass of a data-parallel job takes gradients w.r.t. the cut tensors only (autograd.grad) and
    relies on every PARAMETER gradient being written by a gradient-sink kernel.  Covariates add torch-side pieces between the cut and
    the fused decoder op (the concatenations of module._decoder_operands, CovariateColumns): a gradient that came back through
    autograd instead of the sink would be dropped silently.  NaN-filled buffer, split pass (eager and as two captured graphs), every
    element overwritten and equal to the one-pass step's gradient."""
    from spvipes_amd.data import MinibatchSampler
    n_s, n_p, nb = 25, 10, 3
    B, Gs, H = 256, (640, 600), 128

    def run(overlap, use_graph):
        module, trainer, groups, codes, rows, noise = _setup(dev, "bf16", B, Gs, H, n_s, n_p, nb, cells=1024, seed=7, overlap=overlap)
        assert trainer.overlap == overlap
        sampler = MinibatchSampler([1024, 1024], B, dev, seed=2)
        rows = next(iter(sampler.epoch()))
        if use_graph:
            trainer.capture(rows, warmup=1)
            assert (trainer.graph2 is not None) == overlap
        trainer.fp.grad.fill_(float("nan"))
        trainer.step(rows, kl_weight=0.7, optimizer_step=False)
        torch.cuda.synchronize()
        bad = [(n_, int(torch.isnan(p.grad).sum())) for n_, p in module.named_parameters() if p.grad is not None and bool(torch.isnan(p.grad).any())]
        assert not bad, bad
        for g in range(2):   # the covariate columns themselves carry a gradient
            assert float(module.encoders[g]["private"].fc1.weight.grad[:, Gs[g]:].abs().max()) > 0
            assert float(module.decoders[g].mixture.linear.weight.grad.abs().max()) > 0
        return trainer.fp.grad.clone(), {n_: p.grad.clone() for n_, p in module.named_parameters()}

    flat_one, by_name_one = run(False, False)
    flat_split, by_name_split = run(True, graph)
    for n_ in by_name_one:
        assert torch.equal(by_name_one[n_], by_name_split[n_]), n_
    # (the flat buffer's alignment padding keeps the NaN fill: compare it with the padding mapped to zero)
    assert torch.equal(torch.nan_to_num(flat_one), torch.nan_to_num(flat_split))


def test_covariate_errors(dev):
    from spvipes_amd.data import make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    with pytest.raises(NotImplementedError):
        spVIPESmodule({0: 50, 1: 50}, use_labels=True, n_batch=9)   # 10 + 9 + 1 > 16 operand slots of the private regressor
    groups = [make_synthetic_group(g, 128, 60, dev) for g in range(2)]
    module = spVIPESmodule({0: 60, 1: 60}, use_labels=True, n_batch=2, n_hidden=16).to(dev)
    with pytest.raises(ValueError):
        Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups])   # no codes
    bad = [torch.zeros(128, dtype=torch.int32, device=dev), torch.full((128,), 2, dtype=torch.int32, device=dev)]
    with pytest.raises(ValueError):
        Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups], batch_codes=bad)
    # n_batch = 1 is "no covariates" (nn/networks.py:62): the parameter shapes are the plain ones
    m1 = spVIPESmodule({0: 60, 1: 60}, use_labels=True, n_batch=1, n_hidden=16)
    assert m1.encoders[0]["shared"].fc1.weight.shape == (16, 60) and m1.decoders[0].mixture.linear.weight.shape[1] == 256 + 35
    assert module.encoders[0]["shared"].fc1.weight.shape == (16, 62) and module.decoders[0].mixture.linear.weight.shape[1] == 256 + 35 + 2
    assert module.get_loadings(0, "shared").shape == (60, 25)


def test_user_api_with_batch_key(dev):
    """setup_anndata(batch_key=...) (model/spvipes.py:294,358): categorical codes over the whole AnnData, n_batch = their count; the
    covariates reach training, validation-free latents and the loadings (whose covariate columns the reference drops, :804-805)."""
    from spvipes_amd.model import spVIPES
    from tests._duck import make_duck
    ad = make_duck()
    rng = np.random.default_rng(4)
    ad.obs["donor"] = rng.choice(np.array(["d1", "d2", "d3"]), size=ad.n_obs)
    with pytest.raises(KeyError):
        spVIPES.setup_anndata(ad, groups_key="groups", label_key="cell_type", batch_key="nope")
    spVIPES.setup_anndata(ad, groups_key="groups", label_key="cell_type", batch_key="donor")
    torch.manual_seed(0)
    model = spVIPES(ad, n_hidden=32, n_dimensions_shared=10, n_dimensions_private=5, precision="fp32")
    assert model.module.n_batch == 3 and model.module.encoders[0]["shared"].fc1.weight.shape == (32, 96 + 3)
    gi = [list(ad.uns["groups_obs_indices"][0]), list(ad.uns["groups_obs_indices"][1])]
    model.train(gi, batch_size=64, max_epochs=6, train_size=1.0, n_epochs_kl_warmup=None, plan_kwargs={"lr": 5e-3})
    h = model.history["train_loss"]
    assert len(h) == 6 and all(np.isfinite(h)) and h[-1] < h[0], h
    w0 = model.module.encoders[1]["private"].fc1.weight[:, 80:].detach().clone()
    assert float(w0.abs().max()) > 0 and bool(torch.isfinite(w0).all())

    def latents():
        noise = lambda step, b0, b1: {**{f"enc_{g}_{k}": torch.randn(b, n, generator=torch.Generator().manual_seed(7 * step + g)).to(dev)
                                         for g, b in ((0, b0), (1, b1)) for k, n in (("private", 5), ("shared", 10))},
                                      **{f"poe_{g}": torch.randn(b, 10, generator=torch.Generator().manual_seed(99 + step + g)).to(dev) for g, b in ((0, b0), (1, b1))}}
        return model.get_latent_representation(gi, batch_size=64, _noise=noise)

    a = latents()
    assert a["shared"][0].shape == (300, 10) and a["private"][1].shape == (260, 5)
    assert all(np.isfinite(v).all() for d in a.values() for v in d.values())
    b = latents()
    np.testing.assert_array_equal(a["private"][0], b["private"][0])   # same noise, same codes: same latents
    for t in model._batch:   # ... and other codes: other latents (the encoders read the covariates)
        t.copy_((t + 1) % 3)
    c = latents()
    assert np.abs(c["private"][0] - a["private"][0]).max() > 1e-4
    load = model.get_loadings()
    assert load[(0, "shared")].shape == (96, 10) and load[(1, "private")].shape == (80, 5)
    # a single category is "no covariates" (scvi registers one dummy batch without a batch_key)
    ad1 = make_duck()
    ad1.obs["donor"] = np.array(["only"] * ad1.n_obs)
    spVIPES.setup_anndata(ad1, groups_key="groups", label_key="cell_type", batch_key="donor")
    m1 = spVIPES(ad1, n_hidden=16, n_dimensions_shared=6, n_dimensions_private=3, precision="fp32")
    assert m1.module.n_batch == 1 and m1._batch is None and m1.module.encoders[0]["shared"].fc1.weight.shape == (16, 96)
