"""Product-of-Experts fusion of the two groups' shared posteriors, on device, in closed form.

Replaces /root/reference/src/spVIPES/module/spVIPESmodule.py:
    _label_based_poe :583-718 + _poe2 :282-379      -> label_based_poe
    _paired_poe      :511-571 + _product_of_experts -> paired_poe
    _cluster_based_poe :184-280                     -> cluster_based_poe
    _get_batch_transport_plans :474-482             -> batch_transport_plan
    KL terms of loss :841-868                       -> kl_normal_std

TEST INFRASTRUCTURE (lives under tests/, imported by no product file).  The product path runs all three pairings in HIP
(spvipes_amd.nn_ops.PoELabel / PoEPaired / PoECluster over csrc/spv_small.h); the functions below are the same arithmetic
in plain torch tensor ops: what the HIP kernels were first validated against on the GPU, and what
tests/test_poe_host_logic.py checks against the reference's goldens as a second, independent restatement.

The reference walks labels and cells in Python (three ``.item()`` syncs per cell, :685-701) and
allocates its outputs on the CPU (:661-709); here the pairing is a rank-within-label computed
with a stable device sort, and the fusion is elementwise:

    precision* = 1 + 1/var_self + t          loc* = (loc_self/var_self + u) / precision*
    (t, u) = (1/var_partner, loc_partner/var_partner)   partner exists            (mode 0)
             (1, 0)        label common, rank >= partner count (ones/zeros padding of _poe2)  (mode 1)
             (exp(-1), 0)  label absent from the other minibatch (dummy expert loc=0, logvar=1) (mode 2)
"""
from __future__ import annotations

from collections import OrderedDict
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def kl_normal_std(loc: Tensor, scale: Tensor) -> Tensor:
    """KL(N(loc, scale) || N(0, 1)) summed over the latent dimension."""
    var = scale * scale
    return (0.5 * (var + loc * loc - 1.0 - torch.log(var))).sum(dim=1)


def _rank_within_label(lab: Tensor) -> Tensor:
    n = lab.numel()
    order = torch.argsort(lab, stable=True)
    sorted_lab = lab[order]
    pos = torch.arange(n, device=lab.device)
    is_start = torch.ones(n, dtype=torch.bool, device=lab.device)
    if n > 1:
        is_start[1:] = sorted_lab[1:] != sorted_lab[:-1]
    seg_start = torch.cummax(torch.where(is_start, pos, torch.zeros_like(pos)), 0).values
    rank = torch.empty(n, dtype=torch.long, device=lab.device)
    rank[order] = pos - seg_start
    return rank


def label_partner(labels_self: Tensor, labels_other: Tensor) -> Tuple[Tensor, Tensor]:
    """(partner index in the other minibatch or -1, mode) per cell; see module docstring."""
    ls, lo = labels_self.flatten(), labels_other.flatten()
    rank = _rank_within_label(ls)
    order_o = torch.argsort(lo, stable=True)
    sorted_o = lo[order_o].contiguous()
    first = torch.searchsorted(sorted_o, ls.contiguous(), right=False)
    last = torch.searchsorted(sorted_o, ls.contiguous(), right=True)
    count_o = last - first
    has = rank < count_o
    idx = torch.where(has, first + rank, torch.zeros_like(rank)).clamp(max=max(lo.numel() - 1, 0))
    partner = torch.where(has, order_o[idx], torch.full_like(rank, -1))
    mode = torch.where(has, torch.zeros_like(rank), torch.where(count_o > 0, torch.ones_like(rank), torch.full_like(rank, 2)))
    return partner, mode


def _fuse(loc: Tensor, logvar: Tensor, t: Tensor, u: Tensor) -> Tuple[Tensor, Tensor]:
    var = torch.exp(logvar)
    joint = 1.0 / (1.0 + (1.0 / var + t))
    return (loc / var + u) * joint, torch.log(joint)


def _finish(loc: Tensor, logvar: Tensor, scale: Tensor, eps: Optional[Tensor], clamp: bool) -> "OrderedDict[str, Tensor]":
    """Final reparameterised draw; dict key order is part of the contract (model/spvipes.py:539-540
    and spVIPESmodule.py:729-730 unpack ``.values()`` positionally)."""
    qscale = scale.clamp(min=1e-6) if clamp else scale
    if eps is None:
        eps = torch.randn_like(loc)
    log_z = loc + qscale * eps
    return OrderedDict([
        ("logtheta_loc", loc), ("logtheta_logvar", logvar), ("logtheta_scale", scale),
        ("logtheta_qz", torch.distributions.Normal(loc, qscale, validate_args=False)), ("logtheta_log_z", log_z),
        ("logtheta_theta", F.softmax(log_z, -1)),
    ])


def label_based_poe(shared_stats: Dict[int, dict], labels: Dict[int, Tensor], noise: dict) -> Dict[int, dict]:
    out = {}
    for g, o in ((0, 1), (1, 0)):
        own, other = shared_stats[g], shared_stats[o]
        partner, mode = label_partner(labels[g], labels[o])
        loc, logvar = own["logtheta_loc"], own["logtheta_logvar"]
        pidx = partner.clamp(min=0)
        o_var = torch.exp(other["logtheta_logvar"].index_select(0, pidx))
        o_loc = other["logtheta_loc"].index_select(0, pidx)
        m = mode.unsqueeze(1)
        one = torch.ones_like(o_var)
        t = torch.where(m == 0, 1.0 / o_var, torch.where(m == 1, one, one * 0.36787944117144233))
        u = torch.where(m == 0, o_loc / o_var, torch.zeros_like(o_var))
        j_loc, j_logvar = _fuse(loc, logvar, t, u)
        out[g] = _finish(j_loc, j_logvar, torch.sqrt(torch.exp(j_logvar)), noise.get(f"poe_{g}"), clamp=False)
    return out


def batch_transport_plan(plan: Tensor, global_indices) -> Tensor:
    """plan[idx_0][:, idx_1] (spVIPESmodule.py:480) gathered on device."""
    i0 = global_indices[0].flatten().long()
    i1 = global_indices[1].flatten().long()
    if plan.device != i0.device:
        plan = plan.to(i0.device)
    return plan[i0][:, i1]


def paired_poe(shared_stats: Dict[int, dict], plan_block: Tensor, noise: dict) -> Dict[int, dict]:
    if shared_stats[0]["logtheta_loc"].shape[0] != shared_stats[1]["logtheta_loc"].shape[0]:
        raise AssertionError("Paired PoE requires equal number of cells from both groups")
    part = {0: torch.argmax(plan_block, dim=1), 1: torch.argmax(plan_block, dim=0)}
    out = {}
    for g, o in ((0, 1), (1, 0)):
        own, other = shared_stats[g], shared_stats[o]
        o_var = torch.exp(other["logtheta_logvar"].index_select(0, part[g]))
        j_loc, j_logvar = _fuse(own["logtheta_loc"], own["logtheta_logvar"], 1.0 / o_var, other["logtheta_loc"].index_select(0, part[g]) / o_var)
        out[g] = _finish(j_loc, j_logvar, torch.exp(0.5 * j_logvar), noise.get(f"poe_{g}"), clamp=True)
    return out


def _rownorm(plan: Tensor) -> Tensor:
    rs = plan.sum(dim=1, keepdim=True).clamp(min=1e-10)
    return torch.where(plan > 0, plan / rs, plan)


def cluster_based_poe(shared_stats: Dict[int, dict], plan_block: Tensor, processed_labels, noise: dict) -> Dict[int, dict]:
    """Component-wise plan-weighted experts + _poe2, vectorised over components with masked matmuls.

    For component c present in both minibatches (masks m0, m1):
        E0 = rownorm(T[m0][:, m1]) @ stats0[m1],   E1 = rownorm(T^T[m1][:, m0]) @ stats1[m0]
    (group 0's OWN stats indexed by group 1's mask and vice versa -- reference quirk, :221-229),
    then the k-th cell of c in group g takes row k of _poe2(E0, E1) (padding rule as label PoE).
    Components present in one minibatch only keep their encoder statistics."""
    keys = ("logtheta_loc", "logtheta_logvar", "logtheta_scale")
    c = [processed_labels[0].flatten(), processed_labels[1].flatten()]
    B0, B1 = c[0].numel(), c[1].numel()
    if B0 != B1:
        raise RuntimeError("cluster-based PoE needs equally sized minibatches (the reference indexes one group's statistics with the other's mask)")
    same = c[0].unsqueeze(1) == c[1].unsqueeze(0)  # [B0, B1] same component
    T = [plan_block, plan_block.t()]
    same_g = [same, same.t()]
    rank = [_rank_within_label(c[0]), _rank_within_label(c[1])]
    E = []
    for g in (0, 1):
        W = _rownorm(torch.where(same_g[g], T[g], torch.zeros_like(T[g])))  # rows: cells of g; cols: cells of other in the same comp
        # expert rows for the cells of g: W @ (own stats indexed by the OTHER group's mask) -- the masked
        # columns of W select stats_g[m_other] because both minibatches share the row numbering (B0 == B1)
        E.append({k: W @ shared_stats[g][k] for k in keys})
    out = {}
    for g, o in ((0, 1), (1, 0)):
        partner, mode = label_partner(c[g], c[o])
        own = shared_stats[g]
        pidx = partner.clamp(min=0)
        v_self = torch.exp(E[g]["logtheta_logvar"])
        v_oth = torch.exp(E[o]["logtheta_logvar"].index_select(0, pidx))
        m = mode.unsqueeze(1)
        t = torch.where(m == 0, 1.0 / v_oth, torch.ones_like(v_oth))
        u = torch.where(m == 0, E[o]["logtheta_loc"].index_select(0, pidx) / v_oth, torch.zeros_like(v_oth))
        joint = 1.0 / (1.0 + (1.0 / v_self + t))
        j_loc = (E[g]["logtheta_loc"] / v_self + u) * joint
        j_logvar = torch.log(joint)
        j_scale = torch.sqrt(torch.exp(j_logvar))
        lone = m == 2  # component absent from the other minibatch: encoder stats pass through (:233-244)
        loc = torch.where(lone, own["logtheta_loc"], j_loc)
        logvar = torch.where(lone, own["logtheta_logvar"], j_logvar)
        scale = torch.where(lone, own["logtheta_scale"], j_scale)
        out[g] = _finish(loc, logvar, scale, noise.get(f"poe_{g}"), clamp=True)
    return out
