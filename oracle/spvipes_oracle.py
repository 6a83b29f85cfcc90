"""TEST INFRASTRUCTURE ONLY -- CPU restatement (oracle) of the spVIPES per-minibatch hot path.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this file, and only as the checker / reported baseline.
The product package (``spvipes_amd``) never imports it and has no CPU fallback.

What it restates (all citations are into /root/reference/src/spVIPES/):

    A1  inference prologue          module/spVIPESmodule.py:428-435
    A2  Encoder.forward             nn/networks.py:85-140 (ctor :47-83)
    A4a label PoE                   module/spVIPESmodule.py:583-718 + _poe2 :282-379
    A4b paired PoE                  module/spVIPESmodule.py:511-571 + :573-581 + :474-482
    A4c cluster PoE                 module/spVIPESmodule.py:184-280
    A5  KL terms                    module/spVIPESmodule.py:827-868
    A6  generative latent slicing   module/spVIPESmodule.py:720-771
    A7  LinearDecoderSPVIPE.forward nn/networks.py:314-335 (ctor :185-262)
    A8  NB-mixture log-likelihood   module/spVIPESmodule.py:817-824 (+ third party, below)
    A9  loss assembly               module/spVIPESmodule.py:870-899

Third-party arithmetic (scvi-tools==0.20.0, pinned at pyproject.toml:24, absent
from /root/reference and from this image): ``FCLayers`` with ``n_layers=1``
(Linear -> BatchNorm1d(eps=1e-3, momentum=0.01) -> [ReLU]) and
``NegativeBinomialMixture.log_prob`` (= ``log_mixture_nb`` with a shared theta).
Restated from the published algorithm; see oracle/scvi_standins.py.

Pinning status.  The reference ships NO golden vectors or known-answer tests for
this path (tests/test_basic.py only checks ``__version__``).  The oracle is
pinned instead by outputs of the reference itself: tests/golden/make_goldens.py
loads the reference's own nn/networks.py and module/spVIPESmodule.py by path (in
the build container only), runs its ``spVIPESmodule`` forward+backward on CPU
for all three PoE modes and commits inputs/outputs as tests/golden/*.npz;
tests/test_oracle_vs_golden.py checks this file against them.  The scvi-tools
boundary (FCLayers / log_mixture_nb) is "parity unpinned" by the reference and
is covered by the known-answer tests in tests/test_oracle_known_answers.py.

The restatement is functional: parameters come in as a flat ``state_dict``
using the reference's own parameter names (``encoder_{g}_{shared,private}.*``,
``decoder_{g}.*``, ``px_r.{g}`` -- spVIPESmodule.py:118-120,172-175), so a
state_dict saved from the reference plugs straight in.  It is vectorised (no
per-cell Python loops) and differentiable with torch autograd, so it also
provides gradient references.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
SD = Dict[str, Tensor]

EPS_NB = 1e-8  # scvi-tools log_mixture_nb eps
BN_ENC = dict(eps=1e-5, momentum=0.1)  # torch default BatchNorm1d, nn/networks.py:76,82
BN_DEC = dict(eps=1e-3, momentum=0.01)  # scvi FCLayers BatchNorm1d


# ----------------------------------------------------------------------------
# parameter initialisation with the reference's names / init distributions
# ----------------------------------------------------------------------------
def _linear_init(n_out: int, n_in: int, bias: bool, gen: torch.Generator) -> Tuple[Tensor, Optional[Tensor]]:
    """torch.nn.Linear default init: kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(n_in), 1/sqrt(n_in))."""
    bound = 1.0 / (n_in ** 0.5)
    w = (torch.rand(n_out, n_in, generator=gen) * 2 - 1) * bound
    b = (torch.rand(n_out, generator=gen) * 2 - 1) * bound if bias else None
    return w, b


def _bn_init(sd: SD, prefix: str, n: int) -> None:
    sd[prefix + ".weight"] = torch.ones(n)
    sd[prefix + ".bias"] = torch.zeros(n)
    sd[prefix + ".running_mean"] = torch.zeros(n)
    sd[prefix + ".running_var"] = torch.ones(n)
    sd[prefix + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)


def init_state_dict(
    groups_lengths: Sequence[int],
    n_hidden: int = 128,
    n_dimensions_shared: int = 25,
    n_dimensions_private: int = 10,
    n_hidden_mix: int = 256,
    seed: int = 0,
    n_batch: int = 0,
) -> SD:
    """A state_dict with the reference's key names and init distributions
    (spVIPESmodule.py:118-175; nn/networks.py:66-83,200-262).  The random stream
    differs from the reference's; parity tests always load explicit weights.
    ``n_batch`` > 1 widens every layer that takes covariates by n_batch input
    columns (spVIPESmodule.py:132-133; nn/networks.py:60-68; scvi FCLayers)."""
    gen = torch.Generator().manual_seed(seed)
    sd: SD = {}
    n_s, n_p = n_dimensions_shared, n_dimensions_private
    nc = n_batch if n_batch > 1 else 0
    for g, G in enumerate(groups_lengths):
        sd[f"px_r.{g}"] = torch.randn(G, generator=gen)
    for g, G in enumerate(groups_lengths):
        for kind, n_out in (("shared", n_s), ("private", n_p)):
            p = f"encoder_{g}_{kind}"
            sd[p + ".fc1.weight"], sd[p + ".fc1.bias"] = _linear_init(n_hidden, G + nc, True, gen)
            sd[p + ".fc2.weight"], sd[p + ".fc2.bias"] = _linear_init(n_hidden, n_hidden, True, gen)
            for head in ("mu_encoder", "lvar_encoder"):
                sd[f"{p}.{head}.0.weight"], sd[f"{p}.{head}.0.bias"] = _linear_init(n_out, n_hidden, True, gen)
                _bn_init(sd, f"{p}.{head}.1", n_out)
        d = f"decoder_{g}"
        L = "fc_layers.Layer 0"
        sd[f"{d}.factor_regressor_private.{L}.0.weight"], _ = _linear_init(G, n_p + nc, False, gen)
        _bn_init(sd, f"{d}.factor_regressor_private.{L}.1", G)
        sd[f"{d}.factor_regressor_shared.{L}.0.weight"], _ = _linear_init(G, n_s + nc, False, gen)
        _bn_init(sd, f"{d}.factor_regressor_shared.{L}.1", G)
        sd[f"{d}.sigmoid_decoder.{L}.0.weight"], sd[f"{d}.sigmoid_decoder.{L}.0.bias"] = _linear_init(
            n_hidden_mix, n_s + n_p + nc, True, gen
        )
        _bn_init(sd, f"{d}.sigmoid_decoder.{L}.1", n_hidden_mix)
        sd[f"{d}.mixture.{L}.0.weight"], sd[f"{d}.mixture.{L}.0.bias"] = _linear_init(
            G, n_hidden_mix + n_s + n_p + nc, True, gen
        )
    return sd


# ----------------------------------------------------------------------------
# building blocks
# ----------------------------------------------------------------------------
def batch_norm(
    x: Tensor, sd: SD, prefix: str, training: bool, eps: float, momentum: float, new_stats: Optional[dict] = None
) -> Tensor:
    """BatchNorm1d over dim 0.  Training: biased batch variance for the
    normalisation, running stats updated with the UNBIASED variance (torch semantics)."""
    w, b = sd[prefix + ".weight"], sd[prefix + ".bias"]
    if training:
        mean = x.mean(0)
        var = x.var(0, unbiased=False)
        if new_stats is not None:
            n = x.shape[0]
            with torch.no_grad():
                rm, rv = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
                new_stats[prefix + ".running_mean"] = (1 - momentum) * rm + momentum * mean
                new_stats[prefix + ".running_var"] = (1 - momentum) * rv + momentum * var * (n / max(n - 1, 1))
    else:
        mean, var = sd[prefix + ".running_mean"], sd[prefix + ".running_var"]
    return (x - mean) / torch.sqrt(var + eps) * w + b


def encoder_forward(
    sd: SD,
    prefix: str,
    x: Tensor,
    eps: Tensor,
    training: bool,
    dropout_rate: float = 0.0,
    dropout_mask: Optional[Tensor] = None,
    new_stats: Optional[dict] = None,
    one_hot: Optional[Tensor] = None,
) -> Dict[str, Tensor]:
    """nn/networks.py:85-140 (n_batch<=1 collapses cat_list, :62; else ``one_hot`` [B, n_batch]
    is appended to the input of fc1, :105-119).
    ``eps`` is the standard-normal draw consumed by ``qz.rsample()`` (:128).
    ``dropout_mask`` is the keep-mask (1=keep) of ``self.drop`` (:121) in training."""
    if one_hot is not None:
        x = torch.cat((x, one_hot), dim=-1)  # :118
    h = F.relu(F.linear(x, sd[prefix + ".fc1.weight"], sd[prefix + ".fc1.bias"]))
    h = F.relu(F.linear(h, sd[prefix + ".fc2.weight"], sd[prefix + ".fc2.bias"]))
    if training and dropout_rate > 0:
        if dropout_mask is None:
            raise ValueError("training with dropout needs an explicit keep-mask")
        h = h * dropout_mask / (1.0 - dropout_rate)
    loc = batch_norm(
        F.linear(h, sd[prefix + ".mu_encoder.0.weight"], sd[prefix + ".mu_encoder.0.bias"]),
        sd, prefix + ".mu_encoder.1", training, new_stats=new_stats, **BN_ENC,
    )
    logvar = batch_norm(
        F.linear(h, sd[prefix + ".lvar_encoder.0.weight"], sd[prefix + ".lvar_encoder.0.bias"]),
        sd, prefix + ".lvar_encoder.1", training, new_stats=new_stats, **BN_ENC,
    )
    scale = (0.5 * logvar).exp()
    log_z = loc + scale * eps
    theta = F.softmax(log_z, -1)
    # key order is part of the contract (spVIPESmodule.py:727-730 unpacks .values())
    return {
        "logtheta_loc": loc,
        "logtheta_logvar": logvar,
        "logtheta_scale": scale,
        "log_z": log_z,
        "theta": theta,
    }


def _rank_within_label(labels: Tensor) -> Tensor:
    """rank[i] = number of earlier cells (batch order) carrying the same label."""
    lab = labels.flatten()
    order = torch.argsort(lab, stable=True)
    sorted_lab = lab[order]
    n = lab.numel()
    pos = torch.arange(n)
    is_start = torch.ones(n, dtype=torch.bool)
    is_start[1:] = sorted_lab[1:] != sorted_lab[:-1]
    seg_start = torch.cummax(torch.where(is_start, pos, torch.zeros_like(pos)), 0).values
    rank_sorted = pos - seg_start
    rank = torch.empty(n, dtype=torch.long)
    rank[order] = rank_sorted
    return rank


def label_partner(labels_self: Tensor, labels_other: Tensor) -> Tuple[Tensor, Tensor]:
    """For each cell of ``self``: (partner index in other or -1, mode).

    mode 0: partner exists (k-th same-label cell of the other minibatch),
    mode 1: label common to both minibatches but k >= its count there
            (the shorter side is padded with inverse-variance ONES and mu/var
            ZEROS: spVIPESmodule.py:299-307,317-326),
    mode 2: label absent from the other minibatch (dummy expert loc=0,
            logvar=1: spVIPESmodule.py:633-636,649-652)."""
    ls, lo = labels_self.flatten(), labels_other.flatten()
    rank = _rank_within_label(ls)
    order_o = torch.argsort(lo, stable=True)
    sorted_o = lo[order_o]
    first = torch.searchsorted(sorted_o, ls, right=False)
    last = torch.searchsorted(sorted_o, ls, right=True)
    count_o = last - first
    has = rank < count_o
    idx = torch.where(has, first + rank, torch.zeros_like(rank)).clamp(max=max(lo.numel() - 1, 0))
    partner = torch.where(has, order_o[idx] if lo.numel() else idx, torch.full_like(rank, -1))
    mode = torch.where(has, 0, torch.where(count_o > 0, 1, 2))
    return partner, mode


def _fuse(loc: Tensor, logvar: Tensor, t: Tensor, u: Tensor) -> Tuple[Tensor, Tensor]:
    """prior N(0,1) + own expert + (t, u) = (precision, precision*mean) of the other expert.
    Same operation order as _poe2 (spVIPESmodule.py:345-350)."""
    var = torch.exp(logvar)
    inv = 1.0 / var
    mus = loc / var + u
    joint = torch.ones_like(mus) + (inv + t)
    joint = 1.0 / joint
    return mus * joint, torch.log(joint)


def poe_label(
    shared0: Dict[str, Tensor], shared1: Dict[str, Tensor], labels0: Tensor, labels1: Tensor
) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """_label_based_poe + _poe2 in closed form (loc, logvar, scale per group)."""
    out = []
    for (own, other, l_own, l_other) in ((shared0, shared1, labels0, labels1), (shared1, shared0, labels1, labels0)):
        partner, mode = label_partner(l_own, l_other)
        loc, logvar = own["logtheta_loc"], own["logtheta_logvar"]
        o_loc = other["logtheta_loc"][partner.clamp(min=0)]
        o_var = torch.exp(other["logtheta_logvar"][partner.clamp(min=0)])
        m = mode.unsqueeze(1)
        e_inv = torch.exp(torch.tensor(-1.0, dtype=loc.dtype))
        t = torch.where(m == 0, 1.0 / o_var, torch.where(m == 1, torch.ones_like(o_var), e_inv * torch.ones_like(o_var)))
        u = torch.where(m == 0, o_loc / o_var, torch.zeros_like(o_var))
        j_loc, j_logvar = _fuse(loc, logvar, t, u)
        out.append({
            "logtheta_loc": j_loc,
            "logtheta_logvar": j_logvar,
            "logtheta_scale": torch.sqrt(torch.exp(j_logvar)),  # :358,:363
        })
    return out[0], out[1]


def poe_paired(
    shared0: Dict[str, Tensor], shared1: Dict[str, Tensor], plan_block: Tensor
) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """_paired_poe (:511-571): plan_block = plan[idx0][:, idx1] (:480)."""
    if shared0["logtheta_loc"].shape[0] != shared1["logtheta_loc"].shape[0]:
        raise AssertionError("Paired PoE requires equal number of cells from both groups")
    p01 = torch.argmax(plan_block, dim=1)
    p10 = torch.argmax(plan_block, dim=0)
    out = []
    for own, other, part in ((shared0, shared1, p01), (shared1, shared0, p10)):
        o_var = torch.exp(other["logtheta_logvar"][part])
        j_loc, j_logvar = _fuse(own["logtheta_loc"], own["logtheta_logvar"], 1.0 / o_var, other["logtheta_loc"][part] / o_var)
        out.append({"logtheta_loc": j_loc, "logtheta_logvar": j_logvar, "logtheta_scale": torch.exp(0.5 * j_logvar)})
    return out[0], out[1]


def _rownorm(plan: Tensor) -> Tensor:
    rs = plan.sum(dim=1, keepdim=True).clamp(min=1e-10)
    return torch.where(plan > 0, plan / rs, plan)


def poe_cluster(
    shared0: Dict[str, Tensor], shared1: Dict[str, Tensor], plan_block: Tensor, comp0: Tensor, comp1: Tensor
) -> Tuple[Dict[str, Tensor], Dict[str, Tensor]]:
    """_cluster_based_poe (:184-280).  Quirk kept: the experts of group 0 are the
    plan-weighted averages of group 0's OWN encoder stats indexed with group 1's
    component mask, and vice versa (:221-229); only shape-valid when B0 == B1."""
    keys = ("logtheta_loc", "logtheta_logvar", "logtheta_scale")
    c0, c1 = comp0.flatten(), comp1.flatten()
    B0, B1 = c0.numel(), c1.numel()
    n = shared0["logtheta_loc"].shape[1]
    dt = shared0["logtheta_loc"].dtype
    out0 = {k: torch.zeros(B0, n, dtype=dt) for k in keys}
    out1 = {k: torch.zeros(B1, n, dtype=dt) for k in keys}
    T0, T1 = plan_block, plan_block.T
    for comp in torch.unique(torch.cat([c0, c1])):
        m0, m1 = c0 == comp, c1 == comp
        i0, i1 = torch.nonzero(m0).flatten(), torch.nonzero(m1).flatten()
        if len(i0) and len(i1):
            W0 = _rownorm(T0[m0][:, m1])
            W1 = _rownorm(T1[m1][:, m0])
            E0 = {k: W0 @ shared0[k][m1] for k in keys}
            E1 = {k: W1 @ shared1[k][m0] for k in keys}
            n0, n1 = len(i0), len(i1)
            nmax = max(n0, n1)

            def pad(v, fill):
                if v.shape[0] == nmax:
                    return v
                p = torch.full((nmax, n), fill, dtype=dt)
                return torch.cat([v, p[v.shape[0]:]], 0)

            v0, v1 = torch.exp(E0["logtheta_logvar"]), torch.exp(E1["logtheta_logvar"])
            inv = pad(1.0 / v0, 1.0) + pad(1.0 / v1, 1.0)
            mus = pad(E0["logtheta_loc"] / v0, 0.0) + pad(E1["logtheta_loc"] / v1, 0.0)
            joint = 1.0 / (torch.ones_like(mus) + inv)
            j_loc, j_logvar = mus * joint, torch.log(joint)
            j_scale = torch.sqrt(torch.exp(j_logvar))
            for out, idx, cnt in ((out0, i0, n0), (out1, i1, n1)):
                out["logtheta_loc"] = out["logtheta_loc"].index_put((idx,), j_loc[:cnt])
                out["logtheta_logvar"] = out["logtheta_logvar"].index_put((idx,), j_logvar[:cnt])
                out["logtheta_scale"] = out["logtheta_scale"].index_put((idx,), j_scale[:cnt])
        else:  # unmatched component: encoder stats pass through unfused (:233-244)
            for out, idx, own in ((out0, i0, shared0), (out1, i1, shared1)):
                if len(idx):
                    for k in keys:
                        out[k] = out[k].index_put((idx,), own[k][idx])
    return out0, out1


def poe_sample(poe: Dict[str, Tensor], eps: Tensor, clamp_scale: bool) -> Dict[str, Tensor]:
    """Final draw (:711-716 label; :565-569 paired; :274-278 cluster uses clamp(min=1e-6))."""
    scale = poe["logtheta_scale"].clamp(min=1e-6) if clamp_scale else poe["logtheta_scale"]
    log_z = poe["logtheta_loc"] + scale * eps
    out = dict(poe)
    out["logtheta_qz_scale"] = scale
    out["logtheta_log_z"] = log_z
    out["logtheta_theta"] = F.softmax(log_z, -1)
    return out


def kl_normal_std(loc: Tensor, scale: Tensor) -> Tensor:
    """torch.distributions.kl_divergence(Normal(loc, scale), Normal(0, 1)).sum(1)  (:841-868)."""
    var_ratio = scale.pow(2)
    return (0.5 * (var_ratio + loc.pow(2) - 1 - var_ratio.log())).sum(dim=1)


def decoder_forward(
    sd: SD, prefix: str, z_private: Tensor, z_shared: Tensor, library: Tensor, training: bool,
    new_stats: Optional[dict] = None, one_hot: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Tensor]:
    """nn/networks.py:314-325 (the dead px_scale/mixing lines :327-328 are skipped).
    Returns (px_rate_private, px_rate_shared, px_mixing logits).
    ``one_hot`` [B, n_batch]: every one of the four FCLayers sees cat(input, one_hot)
    (scvi FCLayers.forward with n_cat_list=[n_batch], inject into the single layer)."""
    L = "fc_layers.Layer 0"
    cov = (lambda t: t) if one_hot is None else (lambda t: torch.cat((t, one_hot), dim=-1))
    raw_p = batch_norm(
        F.linear(cov(z_private), sd[f"{prefix}.factor_regressor_private.{L}.0.weight"]),
        sd, f"{prefix}.factor_regressor_private.{L}.1", training, new_stats=new_stats, **BN_DEC,
    )
    rate_p = torch.exp(library) * torch.softmax(raw_p, dim=-1)
    raw_s = batch_norm(
        F.linear(cov(z_shared), sd[f"{prefix}.factor_regressor_shared.{L}.0.weight"]),
        sd, f"{prefix}.factor_regressor_shared.{L}.1", training, new_stats=new_stats, **BN_DEC,
    )
    rate_s = torch.exp(library) * torch.softmax(raw_s, dim=-1)
    zcat = torch.cat([z_private, z_shared], dim=1)
    m = F.relu(
        batch_norm(
            F.linear(cov(zcat), sd[f"{prefix}.sigmoid_decoder.{L}.0.weight"], sd[f"{prefix}.sigmoid_decoder.{L}.0.bias"]),
            sd, f"{prefix}.sigmoid_decoder.{L}.1", training, new_stats=new_stats, **BN_DEC,
        )
    )
    logits = F.linear(
        cov(torch.cat([m, zcat], dim=-1)), sd[f"{prefix}.mixture.{L}.0.weight"], sd[f"{prefix}.mixture.{L}.0.bias"]
    )
    return rate_p, rate_s, logits


def log_mixture_nb(x: Tensor, mu_1: Tensor, mu_2: Tensor, theta: Tensor, pi_logits: Tensor, eps: float = EPS_NB) -> Tensor:
    """scvi-tools 0.20.0 log_mixture_nb, shared-theta branch (see scvi_standins.py)."""
    if theta.ndimension() == 1:
        theta = theta.view(1, theta.size(0))
    l1 = torch.log(theta + mu_1 + eps)
    l2 = torch.log(theta + mu_2 + eps)
    common = torch.lgamma(x + theta) - torch.lgamma(theta) - torch.lgamma(x + 1)
    lt = torch.log(theta + eps)
    nb1 = theta * (lt - l1) + x * (torch.log(mu_1 + eps) - l1) + common
    nb2 = theta * (lt - l2) + x * (torch.log(mu_2 + eps) - l2) + common
    lse = torch.logsumexp(torch.stack((nb1, nb2 - pi_logits)), dim=0)
    return lse - F.softplus(-pi_logits)


# ----------------------------------------------------------------------------
# the whole minibatch step
# ----------------------------------------------------------------------------
def split_latents(private_log_z: Tensor, poe_log_z: Tensor, n_s: int, n_p: int) -> Tuple[Tensor, Tensor]:
    """A6 quirk (spVIPESmodule.py:733,753-754): Z = cat(private, poe); the decoder's
    ``z_private`` is Z[:, n_s:n_s+n_p] and its ``z_shared`` is Z[:, :n_s]."""
    Z = torch.cat((private_log_z, poe_log_z), dim=-1)
    return Z[:, n_s: n_s + n_p], Z[:, :n_s]


def forward_loss(
    sd: SD,
    counts: Sequence[Tensor],
    *,
    n_dimensions_shared: int,
    n_dimensions_private: int,
    noise: Dict[str, Tensor],
    mode: str = "label",
    labels: Optional[Sequence[Tensor]] = None,
    plan_block: Optional[Tensor] = None,
    components: Optional[Sequence[Tensor]] = None,
    kl_weight: float = 1.0,
    training: bool = True,
    dropout_rate: float = 0.0,
    dropout_masks: Optional[Dict[str, Tensor]] = None,
    update_running_stats: bool = False,
    batch_index: Optional[Sequence[Tensor]] = None,
    n_batch: int = 0,
) -> Dict[str, object]:
    """One pass of inference -> generative -> loss for two groups.

    counts[g]  : [B_g, G_g] raw counts of group g's OWN genes (the reference slices
                 them out of the outer-joined matrix at :428-430 and :818).
    noise      : standard-normal draws: "enc_{g}_private", "enc_{g}_shared" [B_g, n],
                 "poe_{g}" [B_g, n_s].
    mode       : "label" (A4a) | "paired" (A4b) | "cluster" (A4c).
    batch_index: with n_batch > 1, the integer batch code of every cell of every group
                 (one-hot encoded and appended to the covariate-taking layers' inputs).
    """
    n_s, n_p = n_dimensions_shared, n_dimensions_private
    one_hot = [None, None]
    if n_batch > 1:
        one_hot = [F.one_hot(b.flatten().long(), n_batch).to(torch.float32) for b in batch_index]
    new_stats: Optional[dict] = {} if (training and update_running_stats) else None
    x = [torch.log(1 + c) for c in counts]  # :432-433
    library = [torch.log(xs.sum(1)).unsqueeze(1) for xs in x]  # :435 (of the log1p'd values)
    private, shared = [], []
    for g in range(2):
        for kind, store in (("private", private), ("shared", shared)):  # call order :444-445
            dm = None if dropout_masks is None else dropout_masks.get(f"enc_{g}_{kind}")
            store.append(
                encoder_forward(
                    sd, f"encoder_{g}_{kind}", x[g], noise[f"enc_{g}_{kind}"], training,
                    dropout_rate=dropout_rate, dropout_mask=dm, new_stats=new_stats, one_hot=one_hot[g],
                )
            )
    if mode == "label":
        p0, p1 = poe_label(shared[0], shared[1], labels[0], labels[1])
        clamp = False
    elif mode == "paired":
        p0, p1 = poe_paired(shared[0], shared[1], plan_block)
        clamp = True
    elif mode == "cluster":
        p0, p1 = poe_cluster(shared[0], shared[1], plan_block, components[0], components[1])
        clamp = True
    else:
        raise ValueError("Either transport plan or labels must be provided for supervised POE.")
    poe = [poe_sample(p0, noise["poe_0"], clamp), poe_sample(p1, noise["poe_1"], clamp)]

    rec, rates, kls = [], [], {}
    for g in range(2):
        z_private, z_shared = split_latents(private[g]["log_z"], poe[g]["logtheta_log_z"], n_s, n_p)
        rate_p, rate_s, logits = decoder_forward(
            sd, f"decoder_{g}", z_private, z_shared, library[g], training, new_stats=new_stats, one_hot=one_hot[g]
        )
        px_r = torch.exp(sd[f"px_r.{g}"])  # :758
        logp = log_mixture_nb(x[g], rate_p, rate_s, px_r, logits)  # evaluated at x = log1p(count), :818-824
        rec.append(-logp.sum(-1))
        rates.append((rate_p, rate_s, logits))
        kls[f"private_{g}"] = kl_normal_std(private[g]["logtheta_loc"], private[g]["logtheta_scale"])
        kls[f"poe_{g}"] = kl_normal_std(poe[g]["logtheta_loc"], poe[g]["logtheta_qz_scale"])
    loss = torch.mean(
        rec[0] + rec[1]
        + kl_weight * kls["private_0"] + kl_weight * kls["poe_0"]
        + kl_weight * kls["private_1"] + kl_weight * kls["poe_1"]
    )
    return {
        "loss": loss,
        "reconstruction_loss": rec,
        "kl_local": kls,
        "private_stats": private,
        "shared_stats": shared,
        "poe_stats": poe,
        "library": library,
        "decoder": rates,
        "new_running_stats": new_stats,
    }


def param_names(sd: SD) -> List[str]:
    """Trainable entries of a state_dict (everything except BN buffers)."""
    return [k for k in sd if not (k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked"))]


def get_loadings(sd: SD, dataset: int, type_latent: str, eps: float = 1e-3, n_batch: int = 0) -> Tensor:
    """spVIPESmodule.get_loadings (spVIPESmodule.py:773-807): diag(gamma / sqrt(running_var + eps)) @ W of one factor
    regressor ([genes, latent dims]); the BatchNorm eps is the scvi FCLayers value 1e-3."""
    if type_latent not in ["shared", "private"]:
        raise ValueError(f"Invalid value for type_latent: {type_latent}. It can only be 'shared' or 'private'")
    L = "fc_layers.Layer 0"
    pre = f"decoder_{dataset}.factor_regressor_{type_latent}.{L}"
    b = sd[pre + ".1.weight"] / torch.sqrt(sd[pre + ".1.running_var"] + eps)
    loadings = torch.matmul(torch.diag(b), sd[pre + ".0.weight"]).detach()
    return loadings[:, :-n_batch] if n_batch > 1 else loadings  # :804-805 (the covariate columns)
