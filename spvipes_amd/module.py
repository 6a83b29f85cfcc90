"""HIP-backed ``spVIPESmodule``: the reference's module protocol for the per-minibatch VAE step.

Mirrors /root/reference/src/spVIPES/module/spVIPESmodule.py (scvi ``BaseModuleClass`` protocol):
``_get_inference_input`` (:381) -> ``inference`` (:425) -> ``_get_generative_input`` (:407) ->
``generative`` (:720) -> ``loss`` (:809), same argument meaning, dict key order and error
behaviour, same parameter names (``encoder_{g}_{shared,private}.*``, ``decoder_{g}.*``,
``px_r.{g}``: :118-120,:172-175) so a reference ``state_dict`` loads unchanged.

The heavy arithmetic runs in hand-written HIP kernels (spvipes_amd/ops.py -> libspvipes_hip.so):
count slicing + log1p + library + both encoders' first layer, and the whole decoder + NB-mixture
likelihood with its backward.  PyTorch carries the [B, <=256] glue between them and autograd.
Deliberately NOT reproduced: the reference's per-cell Python loops and host round trips
(:685-701, :476-480, :817) -- the same results are computed on device in closed form.

Inputs.  ``tensors_by_group`` is a sequence (one entry per group) of dicts.  Either the
reference's layout -- ``"X"`` float32 [B, sum G] in HBM, of which only the group's own columns
``groups_var_indices[g]`` are read -- or the resident layout this build adds for its own train
loop: ``"counts"`` (``ops.GroupCounts``) + ``"rows"`` (int32 [B] cell indices), which fuses the
data loader's row gather into the kernels.
"""
from __future__ import annotations

from collections import OrderedDict
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
from torch import nn

from . import _abi
from ._abi import DEC_CELLS_PER_WG, round_up
from .ops import N_HIDDEN_MIX, EncoderFC1, GroupCounts, Workspace, fork, group_streams, join

X_KEY, BATCH_KEY = "X", "batch"  # scvi.REGISTRY_KEYS.X_KEY / BATCH_KEY


@dataclass
class LossOutput:
    """Same fields as scvi.module.base.LossOutput as used at spVIPESmodule.py:895-897."""

    loss: torch.Tensor
    reconstruction_loss: Optional[dict] = None
    kl_local: Optional[dict] = None
    kl_global: Optional[torch.Tensor] = None
    extra_metrics: dict = field(default_factory=dict)
    reconstruction_loss_mean: Optional[torch.Tensor] = None  # 0-dim: batch mean of the summed reconstruction terms (this build's extra)


# ---- parameter containers with the reference's state_dict layout ---------------------------------
class _LazyMeans(dict):
    """extra_metrics of LossOutput (spVIPESmodule.py:884-897): batch means, evaluated when first read so that a
    training step that never logs them launches nothing for them."""

    def __init__(self, vectors):
        super().__init__()
        self._vectors = dict(vectors)
        for k in vectors:
            dict.__setitem__(self, k, None)

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if v is None and k in self._vectors:
            v = self._vectors[k].detach().mean()
            dict.__setitem__(self, k, v)
        return v

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]

    def get(self, k, default=None):
        return self[k] if k in self else default


class Encoder(nn.Module):
    """Parameters of nn/networks.py:47-83 (forward lives in spVIPESmodule.inference)."""

    def __init__(self, n_input: int, n_topics: int, hidden: int, dropout: float):
        super().__init__()
        self.n_topics, self.dropout = n_topics, dropout
        self.fc1 = nn.Linear(n_input, hidden)
        self.fc2 = nn.Linear(hidden, hidden)
        self.mu_encoder = nn.Sequential(nn.Linear(hidden, n_topics, bias=True), nn.BatchNorm1d(n_topics))
        self.lvar_encoder = nn.Sequential(nn.Linear(hidden, n_topics, bias=True), nn.BatchNorm1d(n_topics))


class _FC1(nn.Module):
    """scvi FCLayers with n_layers=1: fc_layers = Sequential{"Layer 0": Sequential(Linear[, BatchNorm1d(eps=1e-3, momentum=0.01)])}."""

    def __init__(self, n_in: int, n_out: int, bias: bool, batch_norm: bool):
        super().__init__()
        mods: List[nn.Module] = [nn.Linear(n_in, n_out, bias=bias)]
        if batch_norm:
            mods.append(nn.BatchNorm1d(n_out, momentum=0.01, eps=0.001))
        self.fc_layers = nn.Sequential(OrderedDict([("Layer 0", nn.Sequential(*mods))]))

    @property
    def linear(self) -> nn.Linear:
        return self.fc_layers[0][0]

    @property
    def bn(self) -> nn.BatchNorm1d:
        return self.fc_layers[0][1]


class LinearDecoderSPVIPE(nn.Module):
    """Parameters of nn/networks.py:185-262.  ``n_cov``: one-hot batch covariate columns every layer's input is extended by
    (scvi FCLayers with n_cat_list=[n_batch], inject_covariates: the layer sees cat(x, one_hot(batch)))."""

    def __init__(self, n_input_private: int, n_input_shared: int, n_output: int, n_hidden: int = N_HIDDEN_MIX, n_cov: int = 0):
        super().__init__()
        self.factor_regressor_private = _FC1(n_input_private + n_cov, n_output, bias=False, batch_norm=True)
        self.factor_regressor_shared = _FC1(n_input_shared + n_cov, n_output, bias=False, batch_norm=True)
        self.sigmoid_decoder = _FC1(n_input_shared + n_input_private + n_cov, n_hidden, bias=True, batch_norm=True)
        self.mixture = _FC1(n_hidden + n_input_shared + n_input_private + n_cov, n_output, bias=True, batch_norm=False)


class CovariateColumns(torch.autograd.Function):
    """W [R][b + n_cov] (reference column order: ..., columns a..b-1, then the n_cov covariate columns) -> the column order the
    fused decoder contracts against, [W[:, :a] | covariates | W[:, a:b] | n_cov zero columns]: the decoder op sees the covariates
    as n_cov extra latent dimensions behind z_private AND behind z_shared (spVIPESmodule.loss), a layer that reads both gets the
    second copy with zero weights.  The backward pass writes the parameter's gradient buffer itself when the trainer's gradient
    sink is on (nn_ops.grad_out), like every other parameter-gradient kernel of the step."""

    @staticmethod
    def forward(ctx, W, a: int, b: int):
        n_cov = W.shape[1] - b
        ctx.dims = (a, b, n_cov)
        ctx.save_for_backward(W)
        return torch.cat([W[:, :a], W[:, b:], W[:, a:b], W.new_zeros(W.shape[0], n_cov)], 1)

    @staticmethod
    def backward(ctx, g):
        from .nn_ops import grad_out

        (W,) = ctx.saved_tensors
        a, b, n_cov = ctx.dims
        dst, ret = grad_out(W)
        dst[:, :a].copy_(g[:, :a])
        dst[:, a:b].copy_(g[:, a + n_cov: b + n_cov])
        dst[:, b:].copy_(g[:, a: a + n_cov])
        return ret, None, None


class LazyNBMixture:
    """What ``generative`` hands to ``loss`` in place of scvi's NegativeBinomialMixture (spVIPESmodule.py:759): the decoder
    inputs.  The training step evaluates the likelihood in the fused HIP kernel inside ``loss`` and never materialises a
    [B, G] rate; reading ``mu1`` / ``mu2`` / ``theta1`` / ``mixture_logits`` (the attributes of the scvi distribution) runs the
    materialising HIP path (dec_ops.materialize_decoder -> spv_dec_materialize) once and caches the result."""

    def __init__(self, group: int, private_log_z, poe_log_z, library, module=None):
        self.group, self.private_log_z, self.poe_log_z, self.library = group, private_log_z, poe_log_z, library
        self._module, self._cache = module, None

    def materialize(self) -> dict:
        if self._cache is None:
            from .dec_ops import materialize_decoder

            m = self._module
            if m is None:
                raise _abi.SpvError("this LazyNBMixture was built without its module and cannot materialise the decoder outputs")
            with torch.no_grad():
                first, second, par = m._decoder_operands(self.group, self.private_log_z.detach(), self.poe_log_z.detach())
            self._cache = materialize_decoder(m.decoders[self.group], m.px_r[self.group], first, second,
                                              self.library.detach(), m.training, m.nsplit, m._workspace(self.group, self.library.device), par=par)
        return self._cache

    mu1 = property(lambda self: self.materialize()["px_rate_private"])
    mu2 = property(lambda self: self.materialize()["px_rate_shared"])
    theta1 = property(lambda self: self.materialize()["px_r"])
    mixture_logits = property(lambda self: self.materialize()["px_mixing"])


class _GenerativeGroup(dict):
    """One group's entry of generative()["private_poe"] with the reference's keys and key order (spVIPESmodule.py:760-767):
    px_scale_private, px_scale_shared, px_rate_private, px_rate_shared, px, pz.  The four [B, G] tensors are produced on
    first access (LazyNBMixture.materialize); ``loss`` only reads "px"."""

    _LAZY = ("px_scale_private", "px_scale_shared", "px_rate_private", "px_rate_shared")

    def __init__(self, px: LazyNBMixture, pz):
        super().__init__()
        for k in self._LAZY:
            dict.__setitem__(self, k, None)
        dict.__setitem__(self, "px", px)
        dict.__setitem__(self, "pz", pz)

    def __getitem__(self, k):
        v = dict.__getitem__(self, k)
        if v is None and k in self._LAZY:
            v = dict.__getitem__(self, "px").materialize()[k]
            dict.__setitem__(self, k, v)
        return v

    def get(self, k, default=None):
        return self[k] if k in self else default

    def items(self):
        return [(k, self[k]) for k in self.keys()]

    def values(self):
        return [self[k] for k in self.keys()]


class spVIPESmodule(nn.Module):
    """See the module docstring.  Constructor arguments as spVIPESmodule.py:74-95; ``precision`` is
    this build's knob: "bf16" (bf16 MFMA operands, fp32 accumulate) or "fp32" (split-bf16 operands)."""

    def __init__(
        self,
        groups_lengths,
        groups_obs_names=None,
        groups_var_names=None,
        groups_obs_indices=None,
        groups_var_indices=None,
        transport_plan: Optional[torch.Tensor] = None,
        pair_data: bool = False,
        use_labels: bool = False,
        n_labels: Optional[int] = None,
        n_batch: int = 0,
        n_hidden: int = 128,
        n_dimensions_shared: int = 25,
        n_dimensions_private: int = 10,
        dropout_rate: float = 0.1,
        use_batch_norm: bool = True,
        use_layer_norm: bool = False,
        log_variational_inference: bool = True,
        log_variational_generative: bool = True,
        dispersion: str = "gene",
        precision: str = "bf16",
        allow_more_groups: bool = False,
        n_components: Optional[int] = None,
    ):
        super().__init__()
        if not (log_variational_inference and log_variational_generative):
            raise NotImplementedError("the fused kernels implement the reference defaults log_variational_*=True")
        if precision not in ("bf16", "fp32"):
            raise ValueError("precision must be 'bf16' or 'fp32'")
        lengths = list(groups_lengths.values()) if isinstance(groups_lengths, dict) else list(groups_lengths)
        if len(lengths) != 2 and not (allow_more_groups and 2 < len(lengths) <= 4):
            raise ValueError(f"Number of groups is {len(lengths)}, the only supported value is 2")
        # More than two groups (BASELINE config 4) is this build's throughput-only extension: the reference stops at two
        # (data/prepare_adatas.py:94-95).  It runs the cluster-matched PoE with component-mean experts (nn_ops.PoEComponents).
        self.n_groups, self.n_components = len(lengths), n_components
        self.n_dimensions_shared, self.n_dimensions_private = n_dimensions_shared, n_dimensions_private
        self.n_batch, self.n_hidden, self.dropout_rate = n_batch, n_hidden, dropout_rate
        # batch covariates (spVIPESmodule.py:132-133, nn/networks.py:60-68): n_batch <= 1 collapses to no covariate column at all
        self.n_cov = n_cov = n_batch if n_batch > 1 else 0
        if n_cov:
            from .dec_ops import DEC_KP, DEC_KS, KMP
            if len(lengths) != 2:
                raise NotImplementedError("batch covariates are implemented for two groups")
            if n_dimensions_private + n_cov + 1 > DEC_KP or n_dimensions_shared + n_cov + 1 > DEC_KS or \
                    N_HIDDEN_MIX + n_dimensions_private + n_dimensions_shared + 2 * n_cov + 1 > KMP:
                raise NotImplementedError(f"n_batch = {n_batch} does not fit the decoder kernels' operand slots: n_private + n_batch <= {DEC_KP - 1}, "
                                          f"n_shared + n_batch <= {DEC_KS - 1}")
        self.input_dims = {i: g for i, g in enumerate(lengths)}
        self.groups_barcodes, self.groups_genes = groups_obs_names, groups_var_names
        self.groups_obs_indices = groups_obs_indices
        if groups_var_indices is None:
            offs = np.concatenate([[0], np.cumsum(lengths)])
            groups_var_indices = [np.arange(offs[i], offs[i + 1]) for i in range(len(lengths))]
        self.groups_var_indices = [np.asarray(v) for v in groups_var_indices]
        self.dispersion, self.precision = dispersion, precision
        self.use_batch_norm, self.use_layer_norm = use_batch_norm, use_layer_norm
        self.px_r = nn.ParameterList([nn.Parameter(torch.randn(g)) for g in lengths])
        self.encoders, self.decoders = {}, {}
        for g, G in self.input_dims.items():
            enc = {
                "shared": Encoder(G + n_cov, n_dimensions_shared, n_hidden, dropout_rate),
                "private": Encoder(G + n_cov, n_dimensions_private, n_hidden, dropout_rate),
            }
            dec = LinearDecoderSPVIPE(n_dimensions_private, n_dimensions_shared, G, n_cov=n_cov)
            self.encoders[g], self.decoders[g] = enc, dec
            self.add_module(f"encoder_{g}_shared", enc["shared"])
            self.add_module(f"encoder_{g}_private", enc["private"])
            self.add_module(f"decoder_{g}", dec)
        self.use_transport_plan = transport_plan is not None
        self.transport_plan = transport_plan
        self.use_labels, self.n_labels, self.pair_data = use_labels, n_labels, pair_data
        self._ws: Dict[int, Workspace] = {}

    # ---- helpers -----------------------------------------------------------------------------
    @property
    def nsplit(self) -> int:
        return 3 if self.precision == "fp32" else 1

    def enable_device_rng(self, device, key: int, counter0: int = 0) -> None:
        """Draw the step's standard-normal noise with ``spv_randn`` (counter-based, keyed by ``key`` and a device-resident step counter
        that the optimiser's Adam launch increments) instead of torch's generator; the dropout masks use the same counter as seed.
        The noise is a function of (key, counter) alone: setting ``_rng_counter`` reproduces a step.  Eval-mode forward passes
        (validation, ``get_latent_representation``) advance the counter themselves, one tick per batch."""
        self._rng_counter = torch.full((), int(counter0), dtype=torch.int64, device=device)
        self._rng_key = int(key) & 0xFFFFFFFFFFFFFFFF

    def _workspace(self, g: int, device) -> Workspace:
        ws = self._ws.get(g)
        if ws is None or ws.device != device:
            ws = self._ws[g] = Workspace(device)
        return ws

    def _counts_of(self, g: int, group: dict):
        """(GroupCounts, rows, B) for one group's minibatch in either input layout."""
        if "counts" in group:
            rows = group["rows"]
            return group["counts"], rows, int(rows.numel())
        X = group[X_KEY]
        if not X.is_cuda:
            raise _abi.SpvError("spvipes_amd runs on the GPU only: move the minibatch to HBM (no CPU fallback)")
        idx = self.groups_var_indices[g]
        if len(idx) != self.input_dims[g]:
            raise ValueError("groups_var_indices does not match groups_lengths")
        X = X.float().contiguous()
        if len(idx) and np.array_equal(idx, np.arange(idx[0], idx[0] + len(idx))):
            return GroupCounts(X, len(idx), int(idx[0])), None, X.shape[0]
        sel = X.index_select(1, torch.as_tensor(idx, device=X.device)).contiguous()
        return GroupCounts(sel, len(idx), 0), None, X.shape[0]

    def _covariates(self, batch_index, x, dev):
        """Per group (batch codes int32 [B], one_hot(batch) fp32 [B][n_batch]) for a module built with n_batch > 1 (the reference
        appends the one-hot rows to the input of fc1 and of every decoder layer: nn/networks.py:105-119, 314-325), else None.
        Codes of a raw ("X") minibatch are range-checked here (torch's one_hot raises in the reference); a resident data set's are
        checked once by train.Trainer."""
        if not self.n_cov:
            return None
        out = {}
        for g in sorted(x.keys()):
            b = batch_index[g] if batch_index is not None else None
            if b is None:
                raise ValueError(f"n_batch = {self.n_batch}: the minibatch of group {g} carries no '{BATCH_KEY}' codes")
            idx = b.flatten().to(device=dev, dtype=torch.int32).contiguous()
            if idx.numel() != self._step_inputs[g][2]:
                raise ValueError("batch codes do not match the minibatch size")
            if X_KEY in x[g] and idx.numel() and not bool(((idx >= 0) & (idx < self.n_cov)).all()):
                raise ValueError(f"batch codes must lie in [0, n_batch = {self.n_batch})")
            oh = torch.zeros((idx.numel(), self.n_cov), dtype=torch.float32, device=dev)
            oh.scatter_(1, idx.long().unsqueeze(1), 1.0)
            out[g] = (idx, oh)
        return out

    def _decoder_operands(self, g: int, private_log_z, poe_log_z):
        """(first latent, second latent, the decoder's 13 parameters) as DecoderFused / materialize_decoder take them.  Without
        covariates: the two latents and the parameters themselves.  With n_batch > 1 the decoder's four layers see
        cat(input, one_hot(batch)) (nn/networks.py:314-325): the one-hot rows become n_batch extra latent dimensions behind the
        decoder's z_private and behind its z_shared -- Z' = [z_shared | one_hot | z_private | one_hot] is cut by the op's own slicing
        (spVIPESmodule.py:753-754) into exactly those two inputs -- and the two layers that read both (mixing trunk, mixture) get
        their weight columns in that order with zeros for the second copy (CovariateColumns)."""
        from .dec_ops import decoder_params

        par = decoder_params(self.decoders[g], self.px_r[g])
        if not self.n_cov:
            return private_log_z, poe_log_z, par
        n_p, n_s, nb = self.n_dimensions_private, self.n_dimensions_shared, self.n_cov
        oh = self._cov[g][1]
        Z = torch.cat([private_log_z, poe_log_z], 1)
        Zx = torch.cat([Z[:, :n_s], oh, Z[:, n_s:], oh], 1)   # [decoder z_shared | one_hot | decoder z_private | one_hot]
        n_m = par[6].shape[0]
        par = list(par)
        par[6] = CovariateColumns.apply(par[6], n_p, n_p + n_s)
        par[10] = CovariateColumns.apply(par[10], n_m + n_p, n_m + n_p + n_s)
        return Zx[:, :n_p + nb].contiguous(), Zx[:, n_p + nb:].contiguous(), par

    # ---- protocol ----------------------------------------------------------------------------
    def _get_inference_input(self, tensors_by_group):
        x = {i: group for i, group in enumerate(tensors_by_group)}
        batch_index = [group.get(BATCH_KEY) for group in tensors_by_group]
        groups = [group.get("groups") for group in tensors_by_group]
        global_indices = [group.get("indices") for group in tensors_by_group]
        input_dict = {"x": x, "batch_index": batch_index, "groups": groups, "global_indices": global_indices}
        if self.use_transport_plan and not self.pair_data:
            required_key = "processed_transport_labels"
            if required_key not in tensors_by_group[0]:
                raise ValueError(f"{required_key} are required when using transport plan.")
            input_dict["processed_labels"] = [group[required_key] for group in tensors_by_group]
        if self.use_labels:
            if "labels" not in tensors_by_group[0]:
                raise ValueError("Labels are required when using label-based POE.")
            input_dict["labels"] = [group["labels"].flatten() for group in tensors_by_group]
        return input_dict

    def _get_generative_input(self, tensors_by_group, inference_outputs):
        return {
            "private_stats": inference_outputs["private_stats"],
            "shared_stats": inference_outputs["shared_stats"],
            "poe_stats": inference_outputs["poe_stats"],
            "library": inference_outputs["library"],
            "groups": [group.get("groups") for group in tensors_by_group],
            "batch_index": [group.get(BATCH_KEY) for group in tensors_by_group],
        }

    def inference(self, x, batch_index, groups, global_indices, noise: Optional[dict] = None,
                  dropout_masks: Optional[dict] = None, **kwargs):
        """Runs the encoders and the PoE (spVIPESmodule.py:425-472).  ``noise`` optionally injects the
        standard-normal draws ("enc_{g}_private", "enc_{g}_shared", "poe_{g}") for parity tests."""
        from .nn_ops import EncoderSpec, EncoderTails, PoELabel

        noise = noise or {}
        private_stats, shared_stats, library = {}, {}, {}
        self._step_inputs, self._kl_private, self._kl_poe = {}, {}, {}
        h1s, eps_enc = {}, {}
        H = self.n_hidden
        groups_ = sorted(x.keys())
        for g, group in x.items():
            self._step_inputs[g] = self._counts_of(g, group)
        dev0 = self._step_inputs[groups_[0]][0].X.device
        self._cov = self._covariates(batch_index, x, dev0)   # None, or per group (batch codes int32 [B], one_hot fp32 [B][n_batch])
        # Everything that does not depend on the encoders is issued BEFORE the fc1 GEMMs, so that it runs in the shadow of the
        # step's first launches instead of on the critical chain between fc1 and the encoder tails:
        # (1) every standard-normal draw of the step (encoder heads + PoE) out of ONE generator launch,
        n_p_, n_s_ = self.n_dimensions_private, self.n_dimensions_shared
        want = {}
        for g in groups_:
            Bg = self._step_inputs[g][2]
            want[f"enc_{g}_private"], want[f"enc_{g}_shared"], want[f"poe_{g}"] = (Bg, n_p_), (Bg, n_s_), (Bg, n_s_)
        missing = [k for k in want if noise.get(k) is None]
        draws = {}
        if missing:
            total = sum(want[k][0] * want[k][1] for k in missing)
            ctr = getattr(self, "_rng_counter", None)
            if ctr is not None and ctr.device == dev0:   # counter-based device generator (train.Trainer.DEVICE_RNG, spv_randn)
                flat = torch.empty(total, dtype=torch.float32, device=dev0)
                _abi.call("spv_randn", _abi.ptr(flat), total, _abi.ptr(ctr), self._rng_key, _abi.stream_ptr())
                if not self.training:
                    # eval-mode forward passes (get_latent_representation, validation) are followed by no optimiser step, which is what
                    # advances the counter during training: advance it here, or every batch would be perturbed by the SAME noise matrix
                    # (the reference draws fresh torch.randn noise per batch: model/spvipes.py:536-538, nn/networks.py:128-134)
                    _abi.call("spv_counter_bump", _abi.ptr(ctr), _abi.stream_ptr())
            else:
                flat = torch.randn(total, device=dev0)
            off = 0
            for k in missing:
                n = want[k][0] * want[k][1]
                draws[k] = flat[off:off + n].view(want[k])
                off += n
        draw = lambda k: noise[k] if noise.get(k) is not None else draws[k]
        for g in groups_:
            eps_enc[g] = (draw(f"enc_{g}_private"), draw(f"enc_{g}_shared"))
        # (2) the dropout seed of this step,
        ctr = getattr(self, "_rng_counter", None)
        if ctr is not None and ctr.device == dev0:
            self._seed_dev = ctr   # the dropout masks are keyed by the same device-resident step counter (incremented by the Adam launch)
        else:
            if getattr(self, "_seed_dev", None) is None or self._seed_dev.device != dev0 or self._seed_dev is getattr(self, "_rng_counter", None):
                self._seed_dev = torch.zeros((), dtype=torch.int64, device=dev0)  # device-resident: survives hipGraph replay
            if self.training and self.dropout_rate > 0:
                self._seed_dev.add_(1)
        from . import ops as _ops_mod
        _ops_mod.set_step_shape(max(self._step_inputs[g][2] for g in groups_), max(self._step_inputs[g][0].G for g in groups_), len(groups_))
        # the label pairing (rank within label, one workgroup per group, 13-14 us at B 4096) depends on the labels alone.  LABEL_PRE 1: on a
        # side stream beside the fc1 GEMMs (measured: no gain -- its two workgroups hold two CUs' LDS, two of the one-per-CU fc1 workgroups
        # wait for them and run as a second round); LABEL_PRE 2: forked BEHIND the fc1 launches, beside the encoder tails, whose kernels leave
        # most of the chip idle, and joined in front of the fusion kernel
        label_pre, pre_stream = None, None
        label_pre_ok = bool(self.n_groups == 2 and self.use_labels and kwargs.get("labels") is not None and not _ops_mod.serial_streams())

        def fork_label_pairing():
            from .nn_ops import label_partners
            st = group_streams(dev0, 3)[2]
            st.wait_stream(torch.cuda.current_stream(dev0))
            with torch.cuda.stream(st):
                res = label_partners([kwargs["labels"][0], kwargs["labels"][1]], self._workspace(0, dev0))
            return res, st

        label_pre_mode = _ops_mod.label_pre_for(max(self._step_inputs[g][2] for g in groups_))
        if label_pre_mode == 1 and label_pre_ok:
            label_pre, pre_stream = fork_label_pairing()
        streams = None
        grouped = bool(_ops_mod.FC1_GROUPED and len(groups_) >= 2)
        if grouped:   # one autograd node, one launch per kernel for every pair of groups (ops.EncoderFC1Grouped)
            cl, rl, bl, wl, pl = [], [], [], [], []
            for g in groups_:
                counts, rows, B = self._step_inputs[g]
                ep, es = self.encoders[g]["private"], self.encoders[g]["shared"]
                cl.append(counts); rl.append(rows); bl.append(B); wl.append(self._workspace(g, counts.X.device))
                pl += [ep.fc1.weight, ep.fc1.bias, es.fc1.weight, es.fc1.bias]
            outs_fc1 = _ops_mod.EncoderFC1Grouped.apply(cl, rl, bl, self.nsplit, wl, ([self._cov[g] for g in groups_] if self._cov else None), *pl)
            for i, g in enumerate(groups_):
                h1s[g] = outs_fc1[2 * i]
                library[g] = outs_fc1[2 * i + 1].unsqueeze(1)
        for g in ([] if grouped else groups_):
            counts, rows, B = self._step_inputs[g]
            ep, es = self.encoders[g]["private"], self.encoders[g]["shared"]
            ws = self._workspace(g, counts.X.device)
            if streams is None:  # the groups' fc1 GEMMs are independent: group g > 0 runs on a side stream (autograd replays
                streams = group_streams(counts.X.device, len(x))  # each node's backward on its forward stream as well)
                fork(streams)
            with torch.cuda.stream(streams[g % len(streams)]):
                h1, lib = EncoderFC1.apply(counts, rows, B, ep.fc1.weight, ep.fc1.bias, es.fc1.weight, es.fc1.bias, self.nsplit, ws,
                                           self._cov[g] if self._cov else None)
            h1s[g] = h1
            library[g] = lib.unsqueeze(1)
        if streams is not None:
            join(streams)
        if label_pre_mode == 2 and label_pre_ok:
            label_pre, pre_stream = fork_label_pairing()
        # all encoder tails (fc2, dropout, heads, BatchNorm, draw, KL) as a few batched HIP launches: one batch over the
        # four encoders when the groups' minibatches have the same size (training), one batch per group otherwise (ragged
        # inference batches: the last step of get_latent_representation).  Injected dropout keep-masks (parity tests)
        # replace the kernels' counter-based draw.
        dm = dropout_masks or {}
        same_B = len({self._step_inputs[g][2] for g in groups_}) == 1
        for gset in ([groups_[i:i + 2] for i in range(0, len(groups_), 2)] if same_B else [[g] for g in groups_]):   # (the batched kernels carry 8 problem slots: two groups)
            specs, eps_list, masks = [], [], []
            for slot, g in enumerate(gset):
                specs += [EncoderSpec(self.encoders[g]["private"], slot, 0), EncoderSpec(self.encoders[g]["shared"], slot, H)]
                eps_list += [eps_enc[g][0], eps_enc[g][1]]
                masks += [dm.get(f"enc_{g}_private"), dm.get(f"enc_{g}_shared")]
            flat = [p for s in specs for p in s.params()]
            outs = EncoderTails.apply(specs, eps_list, self.training, float(self.dropout_rate), self._seed_dev,
                                      self._workspace(gset[0], h1s[gset[0]].device), masks if dropout_masks else None,
                                      *[h1s[g] for g in gset], *flat)
            for i, s in enumerate(specs):
                g = gset[s.h1_group]
                loc, logvar, scale, log_z, theta, kl = outs[6 * i: 6 * i + 6]
                st = OrderedDict([("logtheta_loc", loc), ("logtheta_logvar", logvar), ("logtheta_scale", scale), ("log_z", log_z),
                                  ("theta", theta), ("qz", torch.distributions.Normal(loc, scale, validate_args=False))])
                if i % 2 == 0:
                    private_stats[g] = st
                    self._kl_private[g] = kl
                else:
                    shared_stats[g] = st

        labels = processed_labels = None
        if self.use_labels and "labels" in kwargs:
            labels = dict(enumerate(kwargs["labels"]))
        if self.use_transport_plan and not self.pair_data:
            processed_labels = kwargs.get("processed_labels")
        if pre_stream is not None and not (self.n_groups == 2 and self.use_labels and labels is not None):
            torch.cuda.current_stream(dev0).wait_stream(pre_stream)
            pre_stream = None
        if self.n_groups > 2 or (isinstance(self.transport_plan, str) and self.transport_plan == "components"):
            # (the N-expert component PoE; with exactly two groups it is reachable through transport_plan="components" only)
            poe_stats = self._components_poe_hip(shared_stats, processed_labels if processed_labels is not None else (list(labels.values()) if labels else None), noise)
        elif self.use_labels and labels is not None:
            # label-based PoE (priority as spVIPESmodule.py:492-493): pairing + fusion + draw + KL in HIP
            dev = shared_stats[0]["logtheta_loc"].device
            e = [draw(f"poe_{g}") for g in (0, 1)]
            if pre_stream is not None:
                torch.cuda.current_stream(dev).wait_stream(pre_stream)
                pre_stream = None
            o = PoELabel.apply([labels[0], labels[1]], e, self._workspace(0, dev), label_pre, shared_stats[0]["logtheta_loc"], shared_stats[0]["logtheta_logvar"],
                               shared_stats[1]["logtheta_loc"], shared_stats[1]["logtheta_logvar"])
            poe_stats = {}
            for g in (0, 1):
                loc, logvar, scale, log_z, theta, kl = o[6 * g: 6 * g + 6]
                poe_stats[g] = OrderedDict([("logtheta_loc", loc), ("logtheta_logvar", logvar), ("logtheta_scale", scale),
                                            ("logtheta_qz", torch.distributions.Normal(loc, scale, validate_args=False)),
                                            ("logtheta_log_z", log_z), ("logtheta_theta", theta)])
                self._kl_poe[g] = kl
        else:
            poe_stats = self._supervised_poe(shared_stats, global_indices, processed_labels, labels, noise)
        return {"private_stats": private_stats, "shared_stats": shared_stats, "poe_stats": poe_stats, "library": library}

    def _supervised_poe(self, shared_stats, global_indices, processed_labels, labels, noise):
        """Dispatch of spVIPESmodule.py:484-509 (same priorities and errors); label-based PoE, first in that order, has
        already been taken by ``inference``."""
        if self.use_transport_plan:
            if self.pair_data:
                return self._paired_poe_hip(shared_stats, global_indices, noise)
            if processed_labels is None:
                raise ValueError("Processed labels are required when using transport plan.")
            return self._cluster_poe_hip(shared_stats, global_indices, processed_labels, noise)
        raise ValueError("Either transport plan or labels must be provided for supervised POE.")

    def sparse_plan(self, device):
        """The transport plan as CSR + CSR of the transpose on ``device`` (built once from whatever the constructor got:
        the reference's dense tensor, a scipy.sparse matrix or a ready SparsePlan)."""
        from .plan import SparsePlan

        sp = getattr(self, "_sparse_plan", None)
        if sp is None or sp.device != torch.device(device):
            sp = self._sparse_plan = SparsePlan.from_any(self.transport_plan, device)
        return sp

    def _paired_poe_hip(self, shared_stats, global_indices, noise):
        """spVIPESmodule.py:511-571 on the sparse plan: arg max partners + fusion + draw + KL in HIP."""
        from .nn_ops import PoEPaired

        if global_indices is None or global_indices[0] is None:
            raise ValueError("paired PoE needs the cells' dataset indices ('indices') to look up the transport plan")
        dev = shared_stats[0]["logtheta_loc"].device
        e = [noise.get(f"poe_{g}") for g in (0, 1)]
        e = [torch.randn_like(shared_stats[g]["logtheta_loc"]) if e[g] is None else e[g] for g in (0, 1)]
        o = PoEPaired.apply(self.sparse_plan(dev), [global_indices[0], global_indices[1]], e, self._workspace(0, dev),
                            shared_stats[0]["logtheta_loc"], shared_stats[0]["logtheta_logvar"],
                            shared_stats[1]["logtheta_loc"], shared_stats[1]["logtheta_logvar"])
        out = {}
        for g in (0, 1):
            loc, logvar, scale, log_z, theta, kl, qscale = o[7 * g: 7 * g + 7]
            out[g] = OrderedDict([("logtheta_loc", loc), ("logtheta_logvar", logvar), ("logtheta_scale", scale),
                                  ("logtheta_qz", torch.distributions.Normal(loc, qscale, validate_args=False)),
                                  ("logtheta_log_z", log_z), ("logtheta_theta", theta)])
            self._kl_poe[g] = kl
        return out

    def _cluster_poe_hip(self, shared_stats, global_indices, processed_labels, noise):
        """spVIPESmodule.py:184-280 on the sparse plan: plan-weighted component experts + _poe2 fusion in HIP."""
        from .nn_ops import PoECluster

        if global_indices is None or global_indices[0] is None:
            raise ValueError("cluster-based PoE needs the cells' dataset indices ('indices') to look up the transport plan")
        dev = shared_stats[0]["logtheta_loc"].device
        e = [noise.get(f"poe_{g}") for g in (0, 1)]
        e = [torch.randn_like(shared_stats[g]["logtheta_loc"]) if e[g] is None else e[g] for g in (0, 1)]
        o = PoECluster.apply(self.sparse_plan(dev), [global_indices[0], global_indices[1]], [processed_labels[0], processed_labels[1]], e,
                             self._workspace(0, dev), shared_stats[0]["logtheta_loc"], shared_stats[0]["logtheta_logvar"],
                             shared_stats[1]["logtheta_loc"], shared_stats[1]["logtheta_logvar"])
        out = {}
        for g in (0, 1):
            loc, logvar, scale, log_z, theta, kl, qscale = o[7 * g: 7 * g + 7]
            out[g] = OrderedDict([("logtheta_loc", loc), ("logtheta_logvar", logvar), ("logtheta_scale", scale),
                                  ("logtheta_qz", torch.distributions.Normal(loc, qscale, validate_args=False)),
                                  ("logtheta_log_z", log_z), ("logtheta_theta", theta)])
            self._kl_poe[g] = kl
        return out

    def _components_poe_hip(self, shared_stats, components, noise):
        """N-group cluster-matched PoE with component-mean experts (csrc/spv_poe_n.h): this build's extension for more than
        two groups, throughput only."""
        from .nn_ops import PoEComponents

        if components is None:
            raise ValueError("Processed labels are required when using transport plan.")
        gs = sorted(shared_stats.keys())
        dev = shared_stats[gs[0]]["logtheta_loc"].device
        e = [noise.get(f"poe_{g}") for g in gs]
        e = [torch.randn_like(shared_stats[g]["logtheta_loc"]) if e[i] is None else e[i] for i, g in enumerate(gs)]
        if self.n_components is None:
            raise ValueError("n_components must be given to the module for more than two groups")
        flat = [t for g in gs for t in (shared_stats[g]["logtheta_loc"], shared_stats[g]["logtheta_logvar"])]
        o = PoEComponents.apply([components[g] for g in gs], int(self.n_components), e, self._workspace(gs[0], dev), *flat)
        out = {}
        for i, g in enumerate(gs):
            loc, logvar, scale, log_z, theta, kl, qscale = o[7 * i: 7 * i + 7]
            out[g] = OrderedDict([("logtheta_loc", loc), ("logtheta_logvar", logvar), ("logtheta_scale", scale),
                                  ("logtheta_qz", torch.distributions.Normal(loc, qscale, validate_args=False)),
                                  ("logtheta_log_z", log_z), ("logtheta_theta", theta)])
            self._kl_poe[g] = kl
        return out

    def generative(self, private_stats, shared_stats, poe_stats, library, groups, batch_index):
        """spVIPESmodule.py:720-771: latent concatenation + slicing quirk; the decoder itself is
        evaluated (fused with the likelihood) in ``loss``."""
        if ((len(private_stats.items()) > 2) or (len(shared_stats.items()) > 2)) and self.n_groups <= 2:
            raise ValueError(
                f"Number of groups passed to `generative` is shared:{len(shared_stats.keys())}, private:{len(private_stats.keys())}, the only supported value is 2"
            )
        out = {}
        for g in sorted(private_stats.keys()):
            # Z = cat(private_log_z, poe_log_z) and the slicing quirk of :733,:753-754 happen inside the fused decoder op
            # (spv_zsplit_fwd); the lazy entry carries the two latents and the library, and materialises px_scale_* /
            # px_rate_* / the mixing logits through spv_dec_materialize only if somebody reads them
            pl, ql = private_stats[g]["log_z"], poe_stats[g]["logtheta_log_z"]
            nz = pl.shape[1] + ql.shape[1]
            key = (pl.shape[0], nz, pl.device)
            cache = self.__dict__.setdefault("_pz_cache", {})
            if key not in cache:   # pz = Normal(0, 1) over the concatenated latent (:760): constant, built once per shape
                cache[key] = torch.distributions.Normal(torch.zeros((pl.shape[0], nz), device=pl.device),
                                                        torch.ones((pl.shape[0], nz), device=pl.device), validate_args=False)
            pz = cache[key]
            out[str(g)] = _GenerativeGroup(LazyNBMixture(g, pl, ql, library[g], module=self), pz)
        return {"private_shared": {}, "private_poe": out}

    def loss(self, tensors_by_group, inference_outputs, generative_outputs, kl_weight: float = 1.0):
        """spVIPESmodule.py:809-899."""
        from .dec_ops import DecoderFused, decoder_params

        gs = sorted(self._step_inputs.keys())
        B0 = self._step_inputs[gs[0]][2]
        for g in gs[1:]:
            if self._step_inputs[g][2] != B0:
                raise RuntimeError(f"The size of tensor a ({B0}) must match the size of tensor b ({self._step_inputs[g][2]}) at non-singleton dimension 0")
        dev = inference_outputs["library"][gs[0]].device
        Bp = round_up(B0, DEC_CELLS_PER_WG)
        cache = getattr(self, "_w_cache", None)
        if cache is None or cache[0] != (B0, dev):
            w_pad = torch.zeros(Bp, dtype=torch.float32, device=dev)
            w_pad[:B0] = 1.0 / B0  # the batch mean of spVIPESmodule.py:870-872 as per-cell weights
            self._w_cache = cache = ((B0, dev), w_pad)
        w_pad = cache[1]
        if isinstance(kl_weight, torch.Tensor):
            klw = kl_weight if (kl_weight.dtype == torch.float32 and kl_weight.device == dev) else kl_weight.to(dev, torch.float32)
        else:
            klw = torch.full((), float(kl_weight), dtype=torch.float32, device=dev)
        px = {g: generative_outputs["private_poe"][str(g)]["px"] for g in gs}
        # the four KL terms (spVIPESmodule.py:841-868) come out of the kernels that sampled the latents (spv_enc_sample_fwd, spv_poe_fuse_fwd /
        # spv_poe_comp_fwd) -- every batch shape, ragged and injected-mask inference batches included, goes through them
        if any(g not in self._kl_private or g not in self._kl_poe for g in gs):
            raise RuntimeError("loss() needs the outputs of this module's latest inference() call (the KL terms are produced there)")
        kl_p = {g: self._kl_private[g] for g in gs}
        kl_q = {g: self._kl_poe[g] for g in gs}
        lat = [t for g in gs for t in (px[g].private_log_z, px[g].poe_log_z)]
        kls = [t for g in gs for t in (kl_p[g], kl_q[g])]
        self._cut = None
        if getattr(self, "split_backward", False) and torch.is_grad_enabled():
            # data-parallel training (train.Trainer): the decoder half of the backward pass ends at detached copies of
            # the tensors that cross from the encoders into the decoder / loss (two latents and two KL vectors per group);
            # the encoder half is started from their gradients afterwards, so the decoder's gradient bucket can be
            # all-reduced in between
            cut = [t.detach().requires_grad_(True) for t in lat + kls]
            self._cut = (lat + kls, cut)
            lat, kls = cut[:len(lat)], cut[len(lat):]
        # reconstruction + kl_weight * mean_b(sum of the KL terms), assembled by one kernel (spv_loss_assemble) per chunk of
        # two groups (the reference has exactly one such chunk)
        loss = rec_mean = None
        rec = {}
        for c0 in range(0, len(gs), 2):
            ch = gs[c0:c0 + 2]
            ops_ = [self._decoder_operands(g, lat[2 * (c0 + i)], lat[2 * (c0 + i) + 1]) for i, g in enumerate(ch)]
            params = [t for o in ops_ for t in o[2]]
            res = DecoderFused.apply([self._step_inputs[g][0] for g in ch], [self._step_inputs[g][1] for g in ch], B0,
                                     [self.decoders[g] for g in ch], [px[g].library for g in ch], w_pad, self.training, self.nsplit,
                                     [self._workspace(g, dev) for g in ch], klw, 2 * len(ch), *[t for o in ops_ for t in o[:2]], *params,
                                     *kls[2 * c0: 2 * c0 + 2 * len(ch)])
            loss = res[0] if loss is None else loss + res[0]
            rec_mean = res[1] if rec_mean is None else rec_mean + res[1]
            for i, g in enumerate(ch):
                rec[g] = res[2 + i]
        return LossOutput(
            loss=loss,
            reconstruction_loss_mean=rec_mean,
            reconstruction_loss={f"reconst_loss_groups_{g + 1}_poe": rec[g] for g in gs},
            kl_local={k: v for g in gs for k, v in ((f"kl_divergence_groups_{g + 1}_private", kl_p[g]), (f"kl_divergence_groups_{g + 1}_poe", kl_q[g]))},
            extra_metrics=_LazyMeans({k: v for g in gs for k, v in ((f"kl_divergence_private_groups_{g + 1}", kl_p[g]), (f"kl_divergence_poe_groups_{g + 1}", kl_q[g]))}),
        )

    def forward(self, tensors, inference_kwargs=None, generative_kwargs=None, loss_kwargs=None, compute_loss=True):
        """scvi BaseModuleClass.forward: inference -> generative -> loss."""
        inference_inputs = self._get_inference_input(tensors)
        inference_outputs = self.inference(**inference_inputs, **(inference_kwargs or {}))
        generative_inputs = self._get_generative_input(tensors, inference_outputs)
        generative_outputs = self.generative(**generative_inputs, **(generative_kwargs or {}))
        if compute_loss:
            return inference_outputs, generative_outputs, self.loss(tensors, inference_outputs, generative_outputs, **(loss_kwargs or {}))
        return inference_outputs, generative_outputs

    @torch.inference_mode()
    def get_loadings(self, dataset: int, type_latent: str) -> np.ndarray:
        """spVIPESmodule.py:773-807: diag(gamma / sqrt(running_var + eps)) @ W of a factor regressor."""
        if type_latent not in ["shared", "private"]:
            raise ValueError(f"Invalid value for type_latent: {type_latent}. It can only be 'shared' or 'private'")
        reg = getattr(self.decoders[dataset], f"factor_regressor_{type_latent}")
        b = reg.bn.weight / torch.sqrt(reg.bn.running_var + reg.bn.eps)
        loadings = (b[:, None] * reg.linear.weight).detach().cpu().numpy()
        if self.n_batch > 1:
            loadings = loadings[:, : -self.n_batch]   # (the covariate columns: spVIPESmodule.py:804-805)
        return loadings
