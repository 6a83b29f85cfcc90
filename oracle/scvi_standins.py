"""TEST INFRASTRUCTURE ONLY -- never imported by the product package.

Stand-ins for the six third-party symbols the reference's hot path imports from
``scvi-tools==0.20.0`` (pinned at /root/reference/pyproject.toml:24), which is
NOT installed in this image and is not under /root/reference:

    scvi.REGISTRY_KEYS                          (spVIPESmodule.py:7)
    scvi.nn.FCLayers                            (nn/networks.py:5)
    scvi.distributions.NegativeBinomialMixture  (spVIPESmodule.py:8)
    scvi.module.base.BaseModuleClass            (spVIPESmodule.py:9)
    scvi.module.base.LossOutput                 (spVIPESmodule.py:9)
    scvi.module.base.auto_move_data             (spVIPESmodule.py:9)

plus one symbol of the reference's HOST-side callers (tests/golden/make_host_goldens.py):

    scvi.dataloaders._data_splitting.validate_data_split   (data/_multi_datasplitter.py:9,55-60)

They restate the *published* scvi-tools 0.20.0 algorithm (scvi/nn/_base_components.py
``FCLayers``; scvi/distributions/_negative_binomial.py ``log_mixture_nb``;
scvi/module/base/_base_module.py ``BaseModuleClass.forward``).  The reference
holds no test that pins results at that boundary, so this part of the parity
chain is "unpinned by the reference" and is instead pinned by known-answer tests
this repo owns (tests/test_oracle_known_answers.py: NB mixture vs
scipy.stats.nbinom; FCLayers layout vs the indexing the reference itself relies
on at spVIPESmodule.py:782-789).

``tests/golden/make_goldens.py`` installs these into ``sys.modules`` so that the
reference's OWN ``nn/networks.py`` and ``module/spVIPESmodule.py`` can be loaded
by path, unmodified, and run on CPU to produce golden vectors for every piece
of spVIPES-authored arithmetic.
"""
from __future__ import annotations

import collections
import sys
import types
from dataclasses import dataclass, field
from typing import Iterable, Optional

import torch
import torch.nn.functional as F
from torch import nn


# ----------------------------------------------------------------------------
# scvi.REGISTRY_KEYS  (only X_KEY / BATCH_KEY are read: spVIPESmodule.py:382-383)
# ----------------------------------------------------------------------------
class _RegistryKeys:
    X_KEY = "X"
    BATCH_KEY = "batch"
    LABELS_KEY = "labels"


REGISTRY_KEYS = _RegistryKeys()


# ----------------------------------------------------------------------------
# scvi.nn.FCLayers (0.20.0).  Call sites: nn/networks.py:200,214,242,253.
# One block per layer:
#   Sequential(Linear(n_in + cat_dim, n_out, bias),
#              BatchNorm1d(n_out, momentum=0.01, eps=0.001) | None,
#              LayerNorm(n_out, elementwise_affine=False)   | None,
#              activation_fn()                              | None,
#              Dropout(p)                                   | None)
# with the None entries dropped, so fc_layers[0][0] is the Linear and
# fc_layers[0][1] the BatchNorm -- exactly what get_loadings indexes
# (spVIPESmodule.py:782-789).
# ----------------------------------------------------------------------------
def _one_hot(index: torch.Tensor, n_cat: int) -> torch.Tensor:
    onehot = torch.zeros(index.size(0), n_cat, device=index.device)
    onehot.scatter_(1, index.type(torch.long), 1)
    return onehot.type(torch.float32)


class FCLayers(nn.Module):
    def __init__(
        self,
        n_in: int,
        n_out: int,
        n_cat_list: Iterable[int] = None,
        n_layers: int = 1,
        n_hidden: int = 128,
        dropout_rate: float = 0.1,
        use_batch_norm: bool = True,
        use_layer_norm: bool = False,
        use_activation: bool = True,
        bias: bool = True,
        inject_covariates: bool = True,
        activation_fn: nn.Module = nn.ReLU,
    ):
        super().__init__()
        self.inject_covariates = inject_covariates
        layers_dim = [n_in] + (n_layers - 1) * [n_hidden] + [n_out]
        if n_cat_list is not None:
            self.n_cat_list = [n_cat if n_cat > 1 else 0 for n_cat in n_cat_list]
        else:
            self.n_cat_list = []
        cat_dim = sum(self.n_cat_list)
        blocks = []
        for i, (d_in, d_out) in enumerate(zip(layers_dim[:-1], layers_dim[1:])):
            mods = [
                nn.Linear(d_in + cat_dim * self.inject_into_layer(i), d_out, bias=bias),
                nn.BatchNorm1d(d_out, momentum=0.01, eps=0.001) if use_batch_norm else None,
                nn.LayerNorm(d_out, elementwise_affine=False) if use_layer_norm else None,
                activation_fn() if use_activation else None,
                nn.Dropout(p=dropout_rate) if dropout_rate > 0 else None,
            ]
            blocks.append((f"Layer {i}", nn.Sequential(*[m for m in mods if m is not None])))
        self.fc_layers = nn.Sequential(collections.OrderedDict(blocks))

    def inject_into_layer(self, layer_num) -> bool:
        return layer_num == 0 or (layer_num > 0 and self.inject_covariates)

    def forward(self, x: torch.Tensor, *cat_list: int):
        one_hot_cat_list = []
        for n_cat, cat in zip(self.n_cat_list, cat_list):
            if n_cat and cat is None:
                raise ValueError("cat not provided while n_cat != 0 in init. params.")
            if n_cat > 1:
                one_hot_cat = _one_hot(cat, n_cat) if cat.size(1) != n_cat else cat
                one_hot_cat_list += [one_hot_cat]
        for i, layers in enumerate(self.fc_layers):
            for layer in layers:
                if layer is not None:
                    if isinstance(layer, nn.BatchNorm1d) and x.dim() == 3:
                        x = torch.cat([(layer(slice_x)).unsqueeze(0) for slice_x in x], dim=0)
                    else:
                        if isinstance(layer, nn.Linear) and self.inject_into_layer(i):
                            if x.dim() == 3:
                                one_hot_cat_list_layer = [
                                    o.unsqueeze(0).expand((x.size(0), o.size(0), o.size(1)))
                                    for o in one_hot_cat_list
                                ]
                            else:
                                one_hot_cat_list_layer = one_hot_cat_list
                            x = torch.cat((x, *one_hot_cat_list_layer), dim=-1)
                        x = layer(x)
        return x


# ----------------------------------------------------------------------------
# scvi.distributions.NegativeBinomialMixture.log_prob  (0.20.0)
# Call sites: spVIPESmodule.py:759 (construction), :823-824 (log_prob).
# ----------------------------------------------------------------------------
def log_mixture_nb(x, mu_1, mu_2, theta_1, theta_2, pi_logits, eps=1e-8):
    """scvi-tools 0.20.0 ``log_mixture_nb`` with a shared inverse dispersion
    (``theta_2 is None`` branch -- the only one the reference reaches, because it
    passes ``theta1=px_r`` only, spVIPESmodule.py:759)."""
    if theta_2 is not None:
        raise NotImplementedError("reference never passes theta2")
    theta = theta_1
    if theta.ndimension() == 1:
        theta = theta.view(1, theta.size(0))
    log_theta_mu_1_eps = torch.log(theta + mu_1 + eps)
    log_theta_mu_2_eps = torch.log(theta + mu_2 + eps)
    lgamma_x_theta = torch.lgamma(x + theta)
    lgamma_theta = torch.lgamma(theta)
    lgamma_x_plus_1 = torch.lgamma(x + 1)
    log_nb_1 = (
        theta * (torch.log(theta + eps) - log_theta_mu_1_eps)
        + x * (torch.log(mu_1 + eps) - log_theta_mu_1_eps)
        + lgamma_x_theta
        - lgamma_theta
        - lgamma_x_plus_1
    )
    log_nb_2 = (
        theta * (torch.log(theta + eps) - log_theta_mu_2_eps)
        + x * (torch.log(mu_2 + eps) - log_theta_mu_2_eps)
        + lgamma_x_theta
        - lgamma_theta
        - lgamma_x_plus_1
    )
    logsumexp = torch.logsumexp(torch.stack((log_nb_1, log_nb_2 - pi_logits)), dim=0)
    softplus_pi = F.softplus(-pi_logits)
    return logsumexp - softplus_pi


class NegativeBinomialMixture:
    def __init__(self, mu1, mu2, theta1, mixture_logits, theta2=None, validate_args=False):
        self.mu1, self.mu2, self.theta1, self.theta2 = mu1, mu2, theta1, theta2
        self.mixture_logits = mixture_logits

    def log_prob(self, value: torch.Tensor) -> torch.Tensor:
        return log_mixture_nb(
            value, self.mu1, self.mu2, self.theta1, self.theta2, self.mixture_logits, eps=1e-08
        )


# ----------------------------------------------------------------------------
# scvi.module.base.{BaseModuleClass, LossOutput, auto_move_data}
# ----------------------------------------------------------------------------
@dataclass
class LossOutput:
    loss: torch.Tensor
    reconstruction_loss: Optional[dict] = None
    kl_local: Optional[dict] = None
    kl_global: Optional[torch.Tensor] = None
    extra_metrics: dict = field(default_factory=dict)


def auto_move_data(fn):
    """CPU-only stand-in: the real decorator moves tensor args to the module's device."""
    return fn


class BaseModuleClass(nn.Module):
    """Restates ``BaseModuleClass.forward`` (the generic inference->generative->loss chain)."""

    def forward(self, tensors, inference_kwargs=None, generative_kwargs=None, loss_kwargs=None, compute_loss=True):
        inference_kwargs = inference_kwargs or {}
        generative_kwargs = generative_kwargs or {}
        loss_kwargs = loss_kwargs or {}
        inference_inputs = self._get_inference_input(tensors)
        inference_outputs = self.inference(**inference_inputs, **inference_kwargs)
        generative_inputs = self._get_generative_input(tensors, inference_outputs)
        generative_outputs = self.generative(**generative_inputs, **generative_kwargs)
        if compute_loss:
            losses = self.loss(tensors, inference_outputs, generative_outputs, **loss_kwargs)
            return inference_outputs, generative_outputs, losses
        return inference_outputs, generative_outputs

# ----------------------------------------------------------------------------
# scvi.dataloaders._data_splitting.validate_data_split (0.20.0).  Call site: data/_multi_datasplitter.py:55-60.
#   n_train = ceil(train_size * n); n_val = n - n_train, or floor(n * validation_size) when a validation size is given.
# ----------------------------------------------------------------------------
def validate_data_split(n_samples: int, train_size: float, validation_size: Optional[float] = None):
    import math

    if train_size > 1.0 or train_size <= 0.0:
        raise ValueError("Invalid train_size. Must be: 0 < train_size <= 1")
    n_train = math.ceil(train_size * n_samples)
    if validation_size is None:
        n_val = n_samples - n_train
    elif validation_size >= 1.0 or validation_size < 0.0:
        raise ValueError("Invalid validation_size. Must be 0 <= validation_size < 1")
    elif (train_size + validation_size) > 1:
        raise ValueError("train_size + validation_size must be between 0 and 1")
    else:
        n_val = math.floor(n_samples * validation_size)
    if n_train == 0:
        raise ValueError(f"With n_samples={n_samples}, train_size={train_size} and validation_size={validation_size}, the resulting train set will be empty.")
    return n_train, n_val


def install() -> None:
    """Register the stand-ins as ``scvi`` / ``scvi.nn`` / ``scvi.distributions`` / ``scvi.module.base``."""
    if "scvi" in sys.modules and not getattr(sys.modules["scvi"], "_spv_standin", False):
        return  # a real scvi-tools is importable: use it
    scvi = types.ModuleType("scvi")
    scvi._spv_standin = True
    scvi.REGISTRY_KEYS = REGISTRY_KEYS
    scvi_nn = types.ModuleType("scvi.nn")
    scvi_nn.FCLayers = FCLayers
    scvi_dist = types.ModuleType("scvi.distributions")
    scvi_dist.NegativeBinomialMixture = NegativeBinomialMixture
    scvi_module = types.ModuleType("scvi.module")
    scvi_base = types.ModuleType("scvi.module.base")
    scvi_base.BaseModuleClass = BaseModuleClass
    scvi_base.LossOutput = LossOutput
    scvi_base.auto_move_data = auto_move_data
    scvi.nn, scvi.distributions, scvi.module = scvi_nn, scvi_dist, scvi_module
    scvi_module.base = scvi_base
    sys.modules.update(
        {
            "scvi": scvi,
            "scvi.nn": scvi_nn,
            "scvi.distributions": scvi_dist,
            "scvi.module": scvi_module,
            "scvi.module.base": scvi_base,
        }
    )
