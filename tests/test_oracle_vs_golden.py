"""Pins the CPU oracle (oracle/spvipes_oracle.py) against outputs of the reference's own
spVIPESmodule (tests/golden/*.npz; generator: tests/golden/make_goldens.py)."""
import pytest
import torch

from oracle import spvipes_oracle as O
from tests._golden import ALL_CASES, LOSS_CASES, Golden

# fp32 restatement vs fp32 reference: differences are summation-order only.
RTOL, ATOL = 2e-5, 2e-6


def _run(g: Golden, dtype=torch.float32, requires_grad=False):
    sd = g.state_dict(dtype)
    if requires_grad:
        for k in O.param_names(sd):
            sd[k].requires_grad_(True)
    out = O.forward_loss(
        sd, g.counts(dtype), n_dimensions_shared=g.n_s, n_dimensions_private=g.n_p, noise=g.noise(dtype),
        mode=g.mode, kl_weight=g.kl_weight, training=g.training, dropout_rate=g.dropout,
        dropout_masks=g.dropout_masks(dtype), update_running_stats=True, **g.poe_inputs(dtype), **g.batch_kwargs(),
    ) if g.has_loss else None
    return sd, out


@pytest.mark.parametrize("case", ALL_CASES)
def test_inference_stats_match_reference(case):
    g = Golden(case)
    sd = g.state_dict()
    x = [torch.log(1 + c) for c in g.counts()]
    noise = g.noise()
    shared, private = [], []
    for grp in range(2):
        lib = torch.log(x[grp].sum(1)).unsqueeze(1)
        torch.testing.assert_close(lib, g.t(f"out/library_{grp}"), rtol=RTOL, atol=ATOL)
        for kind, store in (("private", private), ("shared", shared)):
            dm = (g.dropout_masks() or {}).get(f"enc_{grp}_{kind}")
            st = O.encoder_forward(sd, f"encoder_{grp}_{kind}", x[grp], noise[f"enc_{grp}_{kind}"], g.training,
                                   dropout_rate=g.dropout, dropout_mask=dm, one_hot=g.one_hot(grp))
            store.append(st)
            for k in ("logtheta_loc", "logtheta_logvar", "logtheta_scale", "log_z", "theta"):
                torch.testing.assert_close(st[k], g.t(f"out/{kind}_{grp}/{k}"), rtol=2e-4, atol=2e-5, msg=lambda m: f"{kind}_{grp}/{k}: {m}")
    kw = g.poe_inputs()
    if g.mode == "label":
        p = O.poe_label(shared[0], shared[1], *kw["labels"])
    elif g.mode == "paired":
        p = O.poe_paired(shared[0], shared[1], kw["plan_block"])
    else:
        p = O.poe_cluster(shared[0], shared[1], kw["plan_block"], *kw["components"])
    for grp in range(2):
        s = O.poe_sample(p[grp], noise[f"poe_{grp}"], clamp_scale=(g.mode != "label"))
        for k in ("logtheta_loc", "logtheta_logvar", "logtheta_scale", "logtheta_log_z", "logtheta_theta"):
            torch.testing.assert_close(s[k], g.t(f"out/poe_{grp}/{k}"), rtol=2e-4, atol=2e-5, msg=lambda m: f"poe_{grp}/{k}: {m}")


@pytest.mark.parametrize("case", LOSS_CASES)
def test_loss_terms_match_reference(case):
    g = Golden(case)
    _, out = _run(g)
    for grp in range(2):
        rp, rs, lg = out["decoder"][grp]
        torch.testing.assert_close(rp, g.t(f"out/dec_{grp}/rate_private"), rtol=2e-4, atol=1e-7)
        torch.testing.assert_close(rs, g.t(f"out/dec_{grp}/rate_shared"), rtol=2e-4, atol=1e-7)
        torch.testing.assert_close(lg, g.t(f"out/dec_{grp}/mix_logits"), rtol=2e-4, atol=2e-5)
        torch.testing.assert_close(out["reconstruction_loss"][grp], g.t(f"out/rec_{grp}"), rtol=RTOL, atol=1e-4)
        torch.testing.assert_close(out["kl_local"][f"private_{grp}"], g.t(f"out/kl_private_{grp}"), rtol=1e-4, atol=1e-5)
        torch.testing.assert_close(out["kl_local"][f"poe_{grp}"], g.t(f"out/kl_poe_{grp}"), rtol=1e-4, atol=1e-5)
    torch.testing.assert_close(out["loss"], g.t("out/loss"), rtol=RTOL, atol=1e-5)


@pytest.mark.parametrize("case", LOSS_CASES)
def test_gradients_match_reference(case):
    g = Golden(case)
    sd, out = _run(g, requires_grad=True)
    out["loss"].backward()
    ref = g.grads()
    assert set(ref) == set(O.param_names(sd))
    gmax = max(float(v.abs().max()) for v in ref.values())
    for k, gr in ref.items():
        mine = sd[k].grad if sd[k].grad is not None else torch.zeros_like(sd[k])
        # biases in front of a train-mode BatchNorm have an analytically zero gradient: both sides
        # hold ~1e-8 rounding noise there, hence the absolute floor.
        tol = 5e-4 * float(gr.abs().max()) + 5e-5 * gmax
        err = float((mine - gr).abs().max())
        assert err < tol, f"{case} grad {k}: abs err {err:.3e} > {tol:.3e}"


@pytest.mark.parametrize("case", [c for c in LOSS_CASES if "train" in c])
def test_running_stats_match_reference(case):
    g = Golden(case)
    _, out = _run(g)
    new = out["new_running_stats"]
    checked = 0
    for k, v in g.raw.items():
        if k.startswith("bn/"):
            torch.testing.assert_close(new[k[3:]], torch.tensor(v), rtol=1e-4, atol=1e-6, msg=lambda m: f"{k}: {m}")
            checked += 1
    assert checked == len(new) and checked > 0


def test_float64_oracle_agrees_with_float32_goldens():
    """The oracle in float64 is the high-precision yardstick for the GPU tests; it must sit
    within fp32 rounding of the reference's fp32 run."""
    g = Golden("label_train")
    _, out = _run(g, dtype=torch.float64)
    assert abs(float(out["loss"]) - float(g.raw["out/loss"])) / abs(float(g.raw["out/loss"])) < 1e-5


@pytest.mark.parametrize("case", LOSS_CASES)
def test_loadings_match_reference(case):
    """get_loadings (spVIPESmodule.py:773-807) on the state the reference module holds AFTER the recorded forward pass: the
    golden's state_dict with, in training mode, the running statistics the oracle's own forward pass produces."""
    g = Golden(case)
    sd, out = _run(g)
    sd = {k: v.detach() for k, v in sd.items()}
    if g.training:
        sd.update(out["new_running_stats"])
    for grp in range(2):
        for t in ("private", "shared"):
            torch.testing.assert_close(O.get_loadings(sd, grp, t, n_batch=g.n_batch), g.t(f"out/loadings_{grp}_{t}"), rtol=1e-5, atol=1e-7)
    with pytest.raises(ValueError):
        O.get_loadings(sd, 0, "both")
