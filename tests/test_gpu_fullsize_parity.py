"""GPU: ONE whole training step at the BASELINE.json shapes through the shipped path -- ``Trainer`` -> ``spVIPESmodule``
-> ``EncoderFC1`` (resident f16 log1p image in bf16 mode) / ``EncoderTails`` / PoE / ``DecoderFused`` (two streams,
fused-dz softmax backward, gradient sink into the flat buffer), resident uint16 counts, no graph -- against the CPU
oracle on the same parameters, minibatch rows and noise (reference path: module/spVIPESmodule.py:425-899).

Shapes: C2 (B 4096, G 10 000, H 128, 25/10; bf16 and fp32), C3's per-GPU shard (G 20 000, B 4096), C5's
(G 30 000, paired PoE on a sparse transport plan, fp32; B 4096, as bench.py --config c5 runs it).  C4 (3 groups) has no reference to compare with
(data/prepare_adatas.py:94-95) and is covered by tests/test_gpu_three_groups.py as a consistency check only.

Tolerances (bound  <-  worst value measured over the four cases on the MI355X, round 3: the encoder's first layer now runs on
IEEE f16 operand images in bf16 mode -- csrc/spv_common.h -- which is what brought the latent means inside the north star's 1e-3;
round 2's bf16 operands gave 6.0e-3 on the means and 1.3e-2 on the KL terms):
  ELBO at kl_weight = 1           fp32 mode <= 2e-4  <- 2.2e-7;    bf16 mode <= 1e-3  <- 2.2e-7      (north-star: 1e-3)
  per-cell reconstruction terms   fp32 <= 2e-4  <- 1.4e-6;         bf16 <= 5e-4  <- 9.6e-5
  per-cell KL terms               fp32 <= 1e-3  <- 3.0e-5;         bf16 <= 2e-3  <- 1.4e-3
  private / PoE logtheta_loc      fp32 <= 1e-3 of the column scale  <- 1.2e-5;   bf16 <= 1e-3  <- 6.5e-4   (north-star: 1e-3)
  gradients per parameter kind    relative L2:        fp32 <= 5e-3  <- 2.3e-3;   bf16 <= 3e-2  <- 1.6e-2
                                  max error / max:    fp32 <= 3e-2  <- 1.6e-2 (fc2: a rectifier whose pre-activation sits within
                                  rounding of zero flips for one cell and moves that unit's whole gradient row; every kind
                                  NOT behind a rectifier is below 3e-3);     bf16 <= 8e-2  <- 4.2e-2
  (bf16 mode stores the three [B, G] gradient arrays of the decoder as bf16 and rounds the decoder GEMM operands to bf16: the
  gradient noise is what that storage format gives.)
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GRAD_MAX_BF16, GRAD_L2_BF16, GRAD_MAX_FP32, GRAD_L2_FP32 = 8e-2, 3e-2, 3e-2, 5e-3   # see the module docstring


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from spvipes_amd import _abi
    _abi.load()
    return torch.device("cuda:0")


def _plan(n0, n1, k, seed):
    """a permutation + k random neighbours per row (CSR), seed 2000"""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n0), k + 1)
    cols = np.concatenate([rng.permutation(n1)[:n0, None] if n1 >= n0 else rng.integers(0, n1, (n0, 1)), rng.integers(0, n1, (n0, k))], axis=1).reshape(-1)
    vals = (rng.random(n0 * (k + 1)) + 0.05).astype(np.float32)
    m = sp.coo_matrix((vals, (rows, cols)), shape=(n0, n1)).tocsr()
    m.data = m.data.astype(np.float32)
    return m


def _kind(name: str) -> str:
    """parameter kind for the per-kind gradient bound"""
    for key in ("fc1.weight", "fc1.bias", "fc2", "mu_encoder", "lvar_encoder", "factor_regressor_private", "factor_regressor_shared",
                "sigmoid_decoder", "mixture", "px_r"):
        if key in name:
            return key
    return name


def run_step(dev, *, G, B, n_cells, H, n_s, n_p, precision, mode, seed=0):
    from oracle import spvipes_oracle as O
    from spvipes_amd.data import make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer

    groups = [make_synthetic_group(g, n_cells, G, dev) for g in range(2)]
    torch.manual_seed(seed)
    kw = {}
    plan = None
    if mode == "label":
        kw = dict(use_labels=True)
    else:
        plan = _plan(n_cells, n_cells, 8, 2000)
        kw = dict(transport_plan=plan, pair_data=True)
    module = spVIPESmodule({0: G, 1: G}, n_hidden=H, n_dimensions_shared=n_s, n_dimensions_private=n_p, dropout_rate=0.0,
                           precision=precision, **kw).to(dev)
    sd = {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
    trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups] if mode == "label" else None)
    module.train()
    gen = torch.Generator().manual_seed(seed + 1)
    rows_h = [torch.randperm(n_cells, generator=gen)[:B] for _ in range(2)]
    rows = [r.to(torch.int32).to(dev) for r in rows_h]
    noise = {f"enc_{g}_{k}": torch.randn(B, n, generator=gen) for g in range(2) for k, n in (("private", n_p), ("shared", n_s))}
    noise.update({f"poe_{g}": torch.randn(B, n_s, generator=gen) for g in range(2)})
    lo = trainer.step(rows, kl_weight=1.0, noise={k: v.to(dev) for k, v in noise.items()}, optimizer_step=False)
    inf = trainer.last_outputs[0]
    torch.cuda.synchronize()
    got = {
        "loss": float(lo.loss.detach()),
        "rec": [v.detach().cpu() for v in lo.reconstruction_loss.values()],
        "kl": [v.detach().cpu() for v in lo.kl_local.values()],
        "private_loc": [inf["private_stats"][g]["logtheta_loc"].detach().cpu() for g in range(2)],
        "poe_loc": [inf["poe_stats"][g]["logtheta_loc"].detach().cpu() for g in range(2)],
        "grads": {k: p.grad.detach().cpu().clone() for k, p in module.named_parameters()},
    }
    # ---- the oracle on the same minibatch -------------------------------------------------------------------
    counts_h = []
    for g in range(2):
        Xg = groups[g].counts.X[rows[g].long()].cpu().numpy().view(np.uint16).astype(np.float32)
        counts_h.append(torch.from_numpy(Xg))
    names = [k for k, _ in module.named_parameters()]
    leaves = {k: sd[k].clone().requires_grad_(True) for k in names}
    sd_ref = dict(sd)
    sd_ref.update(leaves)
    okw = {}
    if mode == "label":
        okw["labels"] = [groups[g].labels[rows[g].long()].cpu() for g in range(2)]
    else:
        okw["plan_block"] = torch.from_numpy(plan[rows_h[0].numpy()][:, rows_h[1].numpy()].toarray().astype(np.float32))
    out = O.forward_loss(sd_ref, counts_h, n_dimensions_shared=n_s, n_dimensions_private=n_p, noise=noise,
                         mode="label" if mode == "label" else "paired", training=True, kl_weight=1.0, **okw)
    out["loss"].backward()
    want = {
        "loss": float(out["loss"].detach()),
        "rec": [r.detach() for r in out["reconstruction_loss"]],
        "kl": [out["kl_local"][k].detach() for k in ("private_0", "poe_0", "private_1", "poe_1")],
        "private_loc": [out["private_stats"][g]["logtheta_loc"].detach() for g in range(2)],
        "poe_loc": [out["poe_stats"][g]["logtheta_loc"].detach() for g in range(2)],
        "grads": {k: leaves[k].grad for k in names},
    }
    return got, want


def check(got, want, precision, label):
    fp32 = precision == "fp32"
    m = {}
    m["elbo_rel"] = abs(got["loss"] - want["loss"]) / abs(want["loss"])
    m["rec_rel"] = max(float(((a - b).abs() / b.abs().clamp_min(1.0)).max()) for a, b in zip(got["rec"], want["rec"]))
    m["kl_rel"] = max(float(((a - b).abs() / b.abs().clamp_min(1.0)).max()) for a, b in zip(got["kl"], want["kl"]))
    loc_err = lambda a, b: float(((a - b).abs().max(0).values / b.abs().max(0).values.clamp_min(1e-6)).max())  # per latent column, relative to its scale
    m["private_loc"] = max(loc_err(a, b) for a, b in zip(got["private_loc"], want["private_loc"]))
    m["poe_loc"] = max(loc_err(a, b) for a, b in zip(got["poe_loc"], want["poe_loc"]))
    # the same latent means element by element (reported): the share of elements within a pure rtol of 1e-3, and the 99.9th
    # percentile of |a - b| / max(|b|, 1e-2 column max) -- an element close to zero cannot meet a relative bound on its own value
    def elementwise(a, b):
        den = torch.maximum(b.abs(), 1e-2 * b.abs().max(0).values.clamp_min(1e-6))
        rel = ((a - b).abs() / den).flatten().double()
        pure = ((a - b).abs() <= 1e-3 * b.abs()).double().mean()
        return float(pure), float(torch.quantile(rel[:: max(1, rel.numel() // 1_000_000)], 0.999))
    ew = [elementwise(a, b) for a, b in zip(got["private_loc"] + got["poe_loc"], want["private_loc"] + want["poe_loc"])]
    m["loc_share_within_rtol_1e-3"], m["loc_rel_p999"] = min(e[0] for e in ew), max(e[1] for e in ew)
    # gradients per parameter KIND (e.g. all fc1 weights): max |g - g_ref| over the kind / max |g_ref| over the kind, and the
    # relative L2 error of the kind.  (Per kind, not per tensor: a bias in front of a training-mode BatchNorm has an
    # analytically zero gradient and holds only rounding noise on both sides.)
    num_max, den_max, num_l2, den_l2 = {}, {}, {}, {}
    for k, g_ref in want["grads"].items():
        assert g_ref is not None, k
        kd = _kind(k)
        d = (got["grads"][k] - g_ref).double()
        num_max[kd] = max(num_max.get(kd, 0.0), float(d.abs().max()))
        den_max[kd] = max(den_max.get(kd, 0.0), float(g_ref.abs().max()))
        num_l2[kd] = num_l2.get(kd, 0.0) + float((d * d).sum())
        den_l2[kd] = den_l2.get(kd, 0.0) + float((g_ref.double() ** 2).sum())
    kinds = {k: num_max[k] / max(den_max[k], 1e-30) for k in num_max}
    l2 = {k: (num_l2[k] / max(den_l2[k], 1e-300)) ** 0.5 for k in num_l2}
    m["grad_rel_to_max"], m["grad_rel_l2"] = kinds, l2
    print(f"\n[fullsize parity] {label} {precision}: ELBO {got['loss']:.4f} vs {want['loss']:.4f}  " + ", ".join(
        f"{k}={v:.2e}" for k, v in m.items() if not isinstance(v, dict)))
    print("    grad max-err / max per kind: " + ", ".join(f"{k}={v:.1e}" for k, v in kinds.items()))
    print("    grad relative L2 per kind  : " + ", ".join(f"{k}={v:.1e}" for k, v in l2.items()))
    if os.environ.get("SPV_PARITY_REPORT_ONLY") == "1":
        return m
    assert m["elbo_rel"] <= (2e-4 if fp32 else 1e-3), m
    assert m["rec_rel"] <= (2e-4 if fp32 else 5e-4), m
    assert m["kl_rel"] <= (1e-3 if fp32 else 2e-3), m
    assert m["private_loc"] <= 1e-3 and m["poe_loc"] <= 1e-3, m   # north star: latent means within 1e-3, BOTH precision modes
    bound_max, bound_l2 = (GRAD_MAX_FP32, GRAD_L2_FP32) if fp32 else (GRAD_MAX_BF16, GRAD_L2_BF16)
    bad = {k: (kinds[k], l2[k]) for k in kinds if kinds[k] > bound_max or l2[k] > bound_l2}
    assert not bad, (bad, bound_max, bound_l2)
    return m


@pytest.mark.parametrize("precision", ["bf16", "fp32"])
def test_c2_whole_step_vs_oracle(dev, precision):
    """BASELINE configs[1]: G 10 000, B 4096, H 128, n_s 25, n_p 10, label PoE"""
    got, want = run_step(dev, G=10_000, B=4096, n_cells=6000, H=128, n_s=25, n_p=10, precision=precision, mode="label")
    check(got, want, precision, "C2 4096x10000")


def test_c3_shard_whole_step_vs_oracle(dev):
    """BASELINE configs[2], one rank's shard shape: G 20 000, B 4096, label PoE, bf16"""
    got, want = run_step(dev, G=20_000, B=4096, n_cells=5000, H=128, n_s=25, n_p=10, precision="bf16", mode="label")
    check(got, want, "bf16", "C3 shard 4096x20000")


def test_c5_paired_fp32_whole_step_vs_oracle(dev):
    """BASELINE configs[4]'s shard shape as the bench runs it: G 30 000, B 4096, paired-cells PoE on a sparse plan, fp32 mode (split-bf16
    operands; the one-pass decoder backward on hi / lo planes)"""
    got, want = run_step(dev, G=30_000, B=4096, n_cells=5000, H=128, n_s=25, n_p=10, precision="fp32", mode="paired")
    check(got, want, "fp32", "C5 4096x30000 paired")
