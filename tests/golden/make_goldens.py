#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own module on CPU.

Runs only in the build container (needs /root/reference; never on the GPU box,
never from the test-suite).  It loads the reference's unmodified
``nn/utils.py``, ``nn/networks.py`` and ``module/spVIPESmodule.py`` BY PATH
after registering stand-ins for the six absent scvi-tools symbols
(oracle/scvi_standins.py), drives ``spVIPESmodule.forward`` + ``backward`` for
the three PoE modes on tiny seeded inputs, and writes plain arrays to
``tests/golden/*.npz``:

    in/...    counts, labels / plan / indices / components, kl_weight, flags
    sd/...    the module's full state_dict (reference parameter names)
    noise/... every standard-normal draw the step consumed (captured by wrapping
              torch.distributions.normal._standard_normal) and dropout keep-masks
    out/...   encoder / PoE statistics, library, decoder rates + logits,
              per-cell reconstruction and KL terms, the loss
    grad/...  d loss / d parameter for every trainable parameter
    bn/...    BatchNorm running statistics after the (training-mode) step

Only data is committed -- no reference source travels.

    python tests/golden/make_goldens.py          # writes tests/golden/*.npz
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SPVIPES_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import scvi_standins  # noqa: E402


def load_reference():
    """Load the reference's hot-path files by path under the package name ``spVIPES``."""
    scvi_standins.install()
    src = os.path.join(REF, "src", "spVIPES")
    for name in ("spVIPES", "spVIPES.nn", "spVIPES.module"):
        m = types.ModuleType(name)
        m.__path__ = []  # mark as package
        sys.modules[name] = m

    def load(mod_name, rel):
        spec = importlib.util.spec_from_file_location(mod_name, os.path.join(src, rel))
        mod = importlib.util.module_from_spec(spec)
        sys.modules[mod_name] = mod
        spec.loader.exec_module(mod)
        return mod

    load("spVIPES.nn.utils", "nn/utils.py")
    load("spVIPES.nn.networks", "nn/networks.py")
    return load("spVIPES.module.spVIPESmodule", "module/spVIPESmodule.py")


class Recorder:
    """Captures every N(0,1) draw of Normal.rsample and every dropout keep-mask."""

    def __init__(self):
        import torch.distributions.normal as tdn
        import torch.nn.functional as F

        self.tdn, self.F = tdn, F
        self.orig_sn, self.orig_do = tdn._standard_normal, F.dropout
        self.normals, self.masks = [], []

    def __enter__(self):
        def sn(shape, dtype, device):
            e = self.orig_sn(shape, dtype=dtype, device=device)
            self.normals.append(e.clone())
            return e

        def dropout(input, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return input
            keep = (torch.rand_like(input) >= p).to(input.dtype)
            self.masks.append(keep.clone())
            return input * keep / (1.0 - p)

        self.tdn._standard_normal = sn
        self.F.dropout = dropout
        return self

    def __exit__(self, *a):
        self.tdn._standard_normal = self.orig_sn
        self.F.dropout = self.orig_do


def synth_counts(rng, B, G, p_nonzero=0.35, mean=3.0):
    c = rng.poisson(mean, size=(B, G)) * (rng.random((B, G)) < p_nonzero)
    c[:, 0] += 1  # no empty cell (library = log(sum) must be finite)
    return c.astype(np.float32)


def run_case(ref, name, *, B, G, H, n_s, n_p, mode, training=True, dropout=0.0, seed=0, kl_weight=1.0,
             labels=None, n_cells=None, inference_only=False, n_batch=1):
    torch.manual_seed(seed)
    rng = np.random.default_rng(seed)
    G0, G1 = G
    B0, B1 = B
    var_idx = [np.arange(G0), G0 + np.arange(G1)]
    n_cells = n_cells or (B0 + 5, B1 + 7)
    plan = None
    if mode in ("paired", "cluster"):
        plan = rng.random((n_cells[0], n_cells[1])).astype(np.float32)
        plan *= rng.random(plan.shape) < 0.5  # sparse-ish
        plan[1, :] = 0.0  # an all-zero row: argmax -> 0
        plan[:, 2] = 0.0  # an all-zero column
    module = ref.spVIPESmodule(
        groups_lengths={0: G0, 1: G1},
        groups_obs_names=[None, None],
        groups_var_names={0: None, 1: None},
        groups_obs_indices=[None, None],
        groups_var_indices=var_idx,
        transport_plan=None if plan is None else torch.tensor(plan),
        pair_data=(mode == "paired"),
        use_labels=(mode == "label"),
        n_labels=None,
        n_batch=n_batch,
        n_hidden=H,
        n_dimensions_shared=n_s,
        n_dimensions_private=n_p,
        dropout_rate=dropout,
    )
    # make BN affine params and px_r non-trivial so that goldens exercise them
    with torch.no_grad():
        for k, v in module.state_dict().items():
            if k.endswith(".1.weight"):
                v.copy_(1.0 + 0.2 * torch.randn_like(v))
            elif k.endswith(".1.bias"):
                v.copy_(0.1 * torch.randn_like(v))

    counts = [synth_counts(rng, B0, G0), synth_counts(rng, B1, G1)]
    idx = [rng.permutation(n_cells[0])[:B0], rng.permutation(n_cells[1])[:B1]]
    tensors = []
    for g in range(2):
        Bg = (B0, B1)[g]
        X = np.zeros((Bg, G0 + G1), np.float32)
        X[:, var_idx[g]] = counts[g]
        d = {
            "X": torch.tensor(X),
            "batch": torch.zeros(Bg, 1),
            "groups": torch.full((Bg, 1), float(g)),
            "indices": torch.tensor(idx[g], dtype=torch.float32).unsqueeze(1),
        }
        tensors.append(d)
    batch_codes = None
    if n_batch > 1:   # batch covariates (spVIPESmodule.py:132-133): codes drawn AFTER everything the n_batch = 1 cases draw
        batch_codes = [rng.integers(0, n_batch, size=Bg).astype(np.float32) for Bg in (B0, B1)]
        batch_codes[0][0], batch_codes[1][-1] = 0.0, float(n_batch - 1)
        for g in range(2):
            tensors[g]["batch"] = torch.tensor(batch_codes[g]).unsqueeze(1)
    comps = None
    if mode == "label":
        if labels is None:
            # labels: 0..3 common (unbalanced), 7 only in group 0, 9 only in group 1
            l0 = rng.choice([0, 1, 2, 3, 7], size=B0, p=[0.35, 0.25, 0.15, 0.1, 0.15])
            l1 = rng.choice([0, 1, 2, 3, 9], size=B1, p=[0.15, 0.2, 0.3, 0.2, 0.15])
        else:
            l0, l1 = (np.asarray(v) for v in labels)
        labs = [l0.astype(np.float32), l1.astype(np.float32)]
        for g in range(2):
            tensors[g]["labels"] = torch.tensor(labs[g]).unsqueeze(1)
    elif mode == "cluster":
        # components 0,1 in both groups with unequal sizes; 2 only in group 0; 3 only in group 1
        c0 = rng.choice([0, 1, 2], size=B0, p=[0.5, 0.3, 0.2])
        c1 = rng.choice([0, 1, 3], size=B1, p=[0.3, 0.5, 0.2])
        comps = [c0.astype(np.float32), c1.astype(np.float32)]
        for g in range(2):
            tensors[g]["processed_transport_labels"] = torch.tensor(comps[g]).unsqueeze(1)

    if not training:
        module.train()
        with torch.no_grad():  # move running stats off (0, 1)
            if inference_only:
                for k, v in module.state_dict().items():
                    if k.endswith("running_mean"):
                        v.copy_(0.1 * torch.randn_like(v))
                    elif k.endswith("running_var"):
                        v.copy_(0.5 + torch.rand_like(v))
            else:
                module(tuple(tensors), loss_kwargs={"kl_weight": kl_weight})
        module.eval()
    else:
        module.train()
    sd_before = {k: v.detach().clone() for k, v in module.state_dict().items()}

    with Recorder() as rec:
        if inference_only:
            # what get_latent_representation does per batch (model/spvipes.py:536-538); the loss
            # itself cannot run on ragged minibatches (rec_1 + rec_2 is elementwise, :887-888)
            with torch.no_grad():
                inf = module.inference(**module._get_inference_input(tuple(tensors)))
            gen = lo = None
        else:
            inf, gen, lo = module(tuple(tensors), loss_kwargs={"kl_weight": kl_weight})
    if not inference_only:
        module.zero_grad()
        lo.loss.backward()

    out = {}
    out["in/counts0"], out["in/counts1"] = counts
    out["in/idx0"], out["in/idx1"] = idx[0].astype(np.int64), idx[1].astype(np.int64)
    out["in/kl_weight"] = np.float32(kl_weight)
    out["in/training"] = np.int32(training)
    out["in/dropout"] = np.float32(dropout)
    out["in/dims"] = np.array([H, n_s, n_p], np.int32)
    out["in/mode"] = np.array(mode)
    if mode == "label":
        out["in/labels0"], out["in/labels1"] = labs
    if plan is not None:
        out["in/plan"] = plan
    if comps is not None:
        out["in/comp0"], out["in/comp1"] = comps
    if batch_codes is not None:
        out["in/n_batch"] = np.int32(n_batch)
        out["in/batch0"], out["in/batch1"] = batch_codes
    for k, v in sd_before.items():
        out["sd/" + k] = v.numpy()

    # --- noise bookkeeping: order of Normal.rsample calls in the reference ---
    #   4 encoder draws (g0 private, g0 shared, g1 private, g1 shared: spVIPESmodule.py:444-445),
    #   then (label / cluster modes) 2 DISCARDED draws per _poe2 call (:360,:365),
    #   then the 2 final PoE draws (:715 / :568 / :277).
    normals = rec.normals
    assert len(normals) >= 6
    out["noise/enc_0_private"], out["noise/enc_0_shared"] = normals[0].numpy(), normals[1].numpy()
    out["noise/enc_1_private"], out["noise/enc_1_shared"] = normals[2].numpy(), normals[3].numpy()
    out["noise/poe_0"], out["noise/poe_1"] = normals[-2].numpy(), normals[-1].numpy()
    out["noise/n_discarded_draws"] = np.int32(len(normals) - 6)
    if training and dropout > 0:
        assert len(rec.masks) == 4
        for (g, kind), m in zip(((0, "private"), (0, "shared"), (1, "private"), (1, "shared")), rec.masks):
            out[f"noise/drop_enc_{g}_{kind}"] = m.numpy()

    for g in range(2):
        for kind in ("private", "shared"):
            st = inf[f"{kind}_stats"][g]
            for k in ("logtheta_loc", "logtheta_logvar", "logtheta_scale", "log_z", "theta"):
                out[f"out/{kind}_{g}/{k}"] = st[k].detach().numpy()
        for k in ("logtheta_loc", "logtheta_logvar", "logtheta_scale", "logtheta_log_z", "logtheta_theta"):
            out[f"out/poe_{g}/{k}"] = inf["poe_stats"][g][k].detach().numpy()
        assert list(inf["poe_stats"][g].keys()) == [
            "logtheta_loc", "logtheta_logvar", "logtheta_scale", "logtheta_qz", "logtheta_log_z", "logtheta_theta"]
        out[f"out/library_{g}"] = inf["library"][g].detach().numpy()
        if inference_only:
            continue
        px = gen["private_poe"][str(g)]
        out[f"out/dec_{g}/rate_private"] = px["px_rate_private"].detach().numpy()
        out[f"out/dec_{g}/rate_shared"] = px["px_rate_shared"].detach().numpy()
        out[f"out/dec_{g}/mix_logits"] = px["px"].mixture_logits.detach().numpy()
    if inference_only:
        path = os.path.join(HERE, name + ".npz")
        np.savez_compressed(path, **out)
        print(f"{name}: inference only, draws={len(normals)} -> {os.path.relpath(path, ROOT)}")
        return
    out["out/rec_0"] = lo.reconstruction_loss["reconst_loss_groups_1_poe"].detach().numpy()
    out["out/rec_1"] = lo.reconstruction_loss["reconst_loss_groups_2_poe"].detach().numpy()
    out["out/kl_private_0"] = lo.kl_local["kl_divergence_groups_1_private"].detach().numpy()
    out["out/kl_poe_0"] = lo.kl_local["kl_divergence_groups_1_poe"].detach().numpy()
    out["out/kl_private_1"] = lo.kl_local["kl_divergence_groups_2_private"].detach().numpy()
    out["out/kl_poe_1"] = lo.kl_local["kl_divergence_groups_2_poe"].detach().numpy()
    out["out/loss"] = lo.loss.detach().numpy()
    for k, p in module.named_parameters():
        out["grad/" + k] = (torch.zeros_like(p) if p.grad is None else p.grad).numpy()
    if training:
        for k, v in module.state_dict().items():
            if k.endswith("running_mean") or k.endswith("running_var"):
                out["bn/" + k] = v.detach().numpy()
    # per-gene loadings diag(gamma / sqrt(running_var + eps)) W of the four factor regressors (spVIPESmodule.py:773-807), read
    # from the module AFTER this forward pass (in training mode the running statistics have just been updated)
    for g in range(2):
        for t in ("private", "shared"):
            out[f"out/loadings_{g}_{t}"] = np.asarray(module.get_loadings(g, t))
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **out)
    print(f"{name}: loss={float(lo.loss.detach()):.6f} draws={len(normals)} -> {os.path.relpath(path, ROOT)} "
          f"({os.path.getsize(path) / 1024:.0f} KiB)")


def main():
    ref = load_reference()
    base = dict(G=(24, 32), H=8, n_s=6, n_p=3)
    run_case(ref, "label_train", B=(16, 16), mode="label", seed=1, **base)
    run_case(ref, "label_train_klw", B=(16, 16), mode="label", seed=2, kl_weight=0.25, **base)
    run_case(ref, "label_infer_ragged", B=(16, 11), mode="label", training=False, inference_only=True, seed=13, **base)
    run_case(ref, "label_infer_ragged_rev", B=(5, 16), mode="label", training=False, inference_only=True, seed=14, **base)
    run_case(ref, "label_eval", B=(16, 16), mode="label", training=False, seed=3, **base)
    run_case(ref, "label_train_dropout", B=(16, 16), mode="label", dropout=0.1, seed=4, **base)
    run_case(ref, "label_train_c1dims", B=(8, 8), G=(40, 36), H=16, n_s=10, n_p=5, mode="label", seed=5)
    run_case(ref, "label_train_np_gt_ns", B=(8, 8), G=(24, 32), H=8, n_s=3, n_p=5, mode="label", seed=6)
    run_case(ref, "label_single_class", B=(8, 8), mode="label", seed=7,
             labels=(np.zeros(8), np.zeros(8)), **base)
    run_case(ref, "label_disjoint", B=(8, 8), mode="label", seed=8,
             labels=(np.array([0, 0, 1, 1, 0, 1, 0, 0]), np.array([2, 3, 2, 2, 3, 3, 2, 2])), **base)
    run_case(ref, "paired_train", B=(16, 16), mode="paired", seed=9, **base)
    run_case(ref, "paired_eval", B=(8, 8), mode="paired", training=False, seed=10, **base)
    run_case(ref, "cluster_train", B=(16, 16), mode="cluster", seed=11, **base)
    run_case(ref, "cluster_eval", B=(16, 16), mode="cluster", training=False, seed=12, **base)
    # batch covariates (batch_key given: n_batch > 1, one-hot columns into fc1 and every decoder layer)
    run_case(ref, "label_train_batch3", B=(16, 16), mode="label", seed=15, n_batch=3, **base)
    run_case(ref, "label_eval_batch2", B=(16, 16), mode="label", training=False, seed=16, n_batch=2, **base)
    run_case(ref, "paired_train_batch2", B=(16, 16), mode="paired", seed=17, n_batch=2, **base)


if __name__ == "__main__":
    main()
