"""scvi-tools-style user surface of the MI355X hot path: ``setup_anndata`` / ``train`` /
``get_latent_representation`` / ``get_loadings`` / ``save(dir)`` / ``load(dir, adata=...)``.

Mirrors /root/reference/src/spVIPES/model/spvipes.py (class ``spVIPES``: ctor :216-283,
``setup_anndata`` :285-422, ``get_latent_representation`` :424-650, ``get_loadings`` :652-677) and
model/base/training_mixin.py (``train`` :19-123).  anndata / scvi-tools / Lightning are not part of
this build: the AnnData object is duck-typed (``.X`` dense array, ``.obs`` mapping of columns,
``.uns`` dict, ``.n_obs``) and ``prepare_adatas``' output schema is the input contract --
``uns["groups_lengths" | "groups_var_indices" | "groups_obs_indices" | ...]``, ``obs["groups"]``,
``obs["indices"]`` (data/prepare_adatas.py:97-132).

What stays out: ``process_transport_plan`` (Leiden + Hungarian preprocessing, needs
scanpy) -- cluster matching expects ``obs["processed_transport_labels"]`` to be present already.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _abi
from .data import MinibatchSampler, format_latent_results, latent_loader_mode, latent_steps, to_group_counts
from .module import spVIPESmodule
from .train import Trainer, default_max_epochs

_SETUP_KEY = "_spvipes_amd_setup"


def _codes(column) -> np.ndarray:
    """Categorical codes the way scvi's CategoricalObsField produces them (sorted unique values)."""
    arr = np.asarray(column)
    _, inv = np.unique(arr, return_inverse=True)
    return inv.astype(np.float32)


class spVIPES:
    """Shared-private VAE with Product-of-Experts integration, HIP-accelerated.  Constructor arguments as
    the reference (:216-224); ``precision`` ("bf16" | "fp32") and ``device`` are this build's additions."""

    def __init__(self, adata, n_hidden: int = 128, n_dimensions_shared: int = 25, n_dimensions_private: int = 10,
                 dropout_rate: float = 0.1, precision: str = "bf16", device: Optional[str] = None, **model_kwargs):
        if _SETUP_KEY not in adata.uns:
            raise ValueError("Please run `spVIPES.setup_anndata(adata, groups_key=...)` before creating the model")
        if not torch.cuda.is_available():
            raise _abi.SpvError("spvipes_amd needs an MI355X: there is no CPU fallback for the hot path")
        _abi.load()
        self.adata = adata
        self.device = torch.device(device or "cuda:0")
        self.n_dimensions_private, self.n_dimensions_shared = n_dimensions_private, n_dimensions_shared
        setup = adata.uns[_SETUP_KEY]
        self._setup = setup
        groups_lengths = adata.uns["groups_lengths"]
        var_idx = [np.asarray(v) for v in adata.uns["groups_var_indices"]]
        self.obs_idx = [np.asarray(v) for v in adata.uns["groups_obs_indices"]]
        plan_key = setup.get("transport_plan_key")
        transport_plan = None
        if plan_key:
            if adata.uns.get(plan_key) is None:
                raise ValueError(f"Transport plan not found in adata.uns['{plan_key}']")
            transport_plan = torch.tensor(np.asarray(adata.uns[plan_key]), dtype=torch.float32, device=self.device)
        pair_data = "processed_transport_labels" not in adata.obs  # :249
        use_labels = setup.get("label_key") is not None  # :251
        # batch covariates (setup_anndata(batch_key=...): CategoricalObsField(BATCH_KEY), :358; n_batch = its number of categories, :230):
        # integer codes over the WHOLE AnnData; without a batch_key scvi registers one dummy category, i.e. no covariate column
        batch_codes = _codes(adata.obs[setup["batch_key"]]) if setup.get("batch_key") else None
        n_batch = int(batch_codes.max()) + 1 if batch_codes is not None and len(batch_codes) else 1
        self.module = spVIPESmodule(
            groups_lengths=groups_lengths, groups_obs_names=adata.uns.get("groups_obs_names"),
            groups_var_names=adata.uns.get("groups_var_names"), groups_var_indices=var_idx, groups_obs_indices=self.obs_idx,
            transport_plan=transport_plan, pair_data=pair_data, use_labels=use_labels,
            n_labels=(len(np.unique(np.asarray(adata.obs[setup["label_key"]]))) if use_labels else None), n_batch=n_batch,
            n_hidden=n_hidden, n_dimensions_shared=n_dimensions_shared, n_dimensions_private=n_dimensions_private,
            dropout_rate=dropout_rate, precision=precision, **model_kwargs,
        ).to(self.device)
        # resident per-group count matrices: rows = the group's cells (obs order), columns = its own genes
        X = np.asarray(adata.X if setup.get("layer") is None else adata.layers[setup["layer"]])
        self.counts = [to_group_counts(X[self.obs_idx[g]][:, var_idx[g]], self.device) for g in range(2)]
        obs = adata.obs
        self._labels = [torch.tensor(_codes(obs[setup["label_key"]])[self.obs_idx[g]], device=self.device) for g in range(2)] if use_labels else None
        self._components = ([torch.tensor(_codes(obs["processed_transport_labels"])[self.obs_idx[g]], device=self.device) for g in range(2)]
                            if (transport_plan is not None and not pair_data) else None)
        self._plan_indices = [torch.tensor(np.asarray(obs["indices"])[self.obs_idx[g]].astype(np.float32), device=self.device) for g in range(2)]
        self._batch = ([torch.tensor(np.asarray(batch_codes)[self.obs_idx[g]].astype(np.int32), device=self.device) for g in range(2)]
                       if n_batch > 1 else None)
        self.is_trained_ = False
        self.history: Dict[str, List[float]] = {}
        # what load() re-creates the model from (scvi's BaseModelClass keeps the same record: _get_init_params(locals()), :281)
        self.init_params_ = dict(n_hidden=n_hidden, n_dimensions_shared=n_dimensions_shared, n_dimensions_private=n_dimensions_private,
                                 dropout_rate=dropout_rate, precision=precision, device=str(self.device), **model_kwargs)

    # ------------------------------------------------------------------------------------------
    @classmethod
    def setup_anndata(cls, adata, groups_key: str, match_clusters: bool = False, transport_plan_key: Optional[str] = None,
                      label_key: Optional[str] = None, batch_key: Optional[str] = None, layer: Optional[str] = None, **kwargs) -> None:
        """Registers the fields the model reads (:355-422).  Same arguments, priorities and errors."""
        if batch_key is not None and batch_key not in adata.obs:
            raise KeyError(f"batch_key '{batch_key}' not found in adata.obs")
        if groups_key not in adata.obs:
            raise KeyError(f"groups_key '{groups_key}' not found in adata.obs")
        if transport_plan_key is not None:
            if transport_plan_key not in adata.uns:
                raise ValueError(f"Transport plan key '{transport_plan_key}' not found in adata.uns")
            adata.uns["transport_plan"] = adata.uns[transport_plan_key]
            if match_clusters and "processed_transport_labels" not in adata.obs:
                raise NotImplementedError(
                    "match_clusters=True needs obs['processed_transport_labels'] (the reference derives it with "
                    "scanpy Leiden + Hungarian matching in process_transport_plan, which is outside this build)")
            if "indices" not in adata.obs:
                raise ValueError("'indices' must be present in adata.obs when using a transport plan")
        if label_key is not None and label_key not in adata.obs:
            raise KeyError(f"label_key '{label_key}' not found in adata.obs")
        for k in ("groups_lengths", "groups_var_indices", "groups_obs_indices"):
            if k not in adata.uns:
                raise ValueError(f"adata.uns['{k}'] missing: build the AnnData with prepare_adatas (or the same schema)")
        adata.uns[_SETUP_KEY] = {"groups_key": groups_key, "match_clusters": match_clusters, "transport_plan_key": transport_plan_key,
                                 "label_key": label_key, "batch_key": batch_key, "layer": layer}

    # ------------------------------------------------------------------------------------------
    def _local_rows(self, g: int, indices: Sequence[int]) -> np.ndarray:
        """adata row numbers -> rows of group g's resident matrix."""
        pos = np.full(self.adata.n_obs, -1, dtype=np.int64)
        pos[self.obs_idx[g]] = np.arange(len(self.obs_idx[g]))
        loc = pos[np.asarray(indices)]
        if (loc < 0).any():
            raise ValueError(f"group_indices_list[{g}] contains cells that do not belong to group {g}")
        return loc

    def _minibatch(self, rows: Sequence[torch.Tensor]):
        out = []
        for g, r in enumerate(rows):
            rl = r.long()
            d = {"counts": self.counts[g], "rows": r, "indices": self._plan_indices[g][rl].unsqueeze(1), "groups": None, "batch": None}
            if self._batch is not None:
                d["batch"] = self._batch[g][rl].unsqueeze(1)
            if self._labels is not None:
                d["labels"] = self._labels[g][rl].unsqueeze(1)
            if self._components is not None:
                d["processed_transport_labels"] = self._components[g][rl].unsqueeze(1)
            out.append(d)
        return tuple(out)

    def train(self, group_indices_list: List[List[int]], batch_size: Optional[int] = 128, max_epochs: Optional[int] = None,
              use_gpu=None, train_size: float = 0.9, validation_size: Optional[float] = None, early_stopping: bool = False,
              plan_kwargs: Optional[dict] = None, n_steps_kl_warmup: Optional[int] = None, n_epochs_kl_warmup: Optional[int] = 400,
              seed: int = 0, **trainer_kwargs) -> None:
        """training_mixin.py:19-123, same signature.  ``validation_size`` / ``train_size`` hold cells out exactly as
        MultiGroupDataSplitter does; ``early_stopping`` monitors ``elbo_validation`` with scvi-tools' defaults (patience 45;
        ``early_stopping_patience`` / ``early_stopping_min_delta`` / ``check_val_every_n_epoch`` in ``trainer_kwargs`` as for
        scvi's Trainer).  ``use_gpu`` is accepted and ignored (the path only runs on the GPU).  Inside an initialised
        ``torch.distributed`` job every rank trains on its own shard of the training cells from identical
        initial weights (rank 0's are broadcast)."""
        import torch.distributed as dist

        if max_epochs is None:
            max_epochs = default_max_epochs(self.adata.n_obs)
        plan_kwargs = dict(plan_kwargs or {})
        world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        rank = dist.get_rank() if world > 1 else 0
        local = [self._local_rows(g, group_indices_list[g]) for g in range(2)]
        sampler = MinibatchSampler([len(l) for l in local], batch_size, self.device, seed=seed, train_size=train_size,
                                   validation_size=validation_size, group_indices_list=local, rank=rank, world=world)
        trainer = Trainer(self.module, self.counts, lr=plan_kwargs.get("lr", 1e-3), eps=plan_kwargs.get("eps", 0.01),
                          weight_decay=plan_kwargs.get("weight_decay", 1e-6), n_epochs_kl_warmup=n_epochs_kl_warmup,
                          n_steps_kl_warmup=n_steps_kl_warmup, batch_codes=self._batch)
        if world > 1:
            dist.broadcast(trainer.fp.flat, src=0)
            trainer.parameters_changed()
        trainer.minibatch = self._minibatch  # labels / components / plan indices of this AnnData
        self.trainer_, self.sampler_ = trainer, sampler
        self.history = trainer.fit(sampler, max_epochs, log_every=trainer_kwargs.get("log_every", 1), use_graph=trainer_kwargs.get("use_graph", True),
                                   early_stopping=bool(trainer_kwargs.get("early_stopping", early_stopping)),
                                   early_stopping_patience=trainer_kwargs.get("early_stopping_patience", 45),
                                   early_stopping_min_delta=trainer_kwargs.get("early_stopping_min_delta", 0.0),
                                   check_val_every_n_epoch=trainer_kwargs.get("check_val_every_n_epoch"))
        self.is_trained_ = True

    # ------------------------------------------------------------------------------------------
    @torch.no_grad()
    def get_latent_representation(self, group_indices_list: List[List[int]], adata=None, indices=None, normalized: bool = False,
                                  give_mean: bool = True, mc_samples: int = 5000, batch_size: Optional[int] = None,
                                  drop_last: Optional[bool] = None, _noise=None) -> dict:
        """model/spvipes.py:424-650.  Returns the SAMPLED ``log_z`` (as the reference does) as
        {"shared","private","shared_reordered","private_reordered"} -> {0: ndarray, 1: ndarray}.  ``_noise`` (tests only):
        callable (step, B0, B1) -> dict of injected standard-normal draws for module.inference."""
        if normalized:
            raise NotImplementedError("normalized=True is broken in the reference (nothing is collected for the shared latents, :542-544)")
        batch_size = batch_size or 128
        n1, n2 = (len(g) for g in group_indices_list)
        local = [self._local_rows(g, group_indices_list[g]) for g in range(2)]
        drop_last, use_cycling = latent_loader_mode(self.module.use_labels, self.module.use_transport_plan, self.module.pair_data, drop_last)   # :468-503
        res = {k: [] for k in ("s0", "s1", "p0", "p1", "i1")}
        was_training = self.module.training
        self.module.eval()
        for step, (r0, r1) in enumerate(latent_steps(local, batch_size, drop_last, use_cycling)):
            rows = [torch.as_tensor(r, dtype=torch.int32, device=self.device) for r in (r0, r1)]
            tensors = self._minibatch(rows)
            kw = {} if _noise is None else {"noise": _noise(step, len(r0), len(r1))}
            out = self.module.inference(**self.module._get_inference_input(tensors), **kw)
            res["s0"].append(out["poe_stats"][0]["logtheta_log_z"].cpu().numpy())
            res["s1"].append(out["poe_stats"][1]["logtheta_log_z"].cpu().numpy())
            res["p0"].append(out["private_stats"][0]["log_z"].cpu().numpy())
            res["p1"].append(out["private_stats"][1]["log_z"].cpu().numpy())
            res["i1"].append(tensors[1]["indices"].cpu().numpy())
        self.module.train(was_training)
        return format_latent_results(res["p0"], res["p1"], res["s0"], res["s1"], res["i1"], n1, n2)

    def get_loadings(self) -> dict:
        """:652-677: per-gene weights of the linear decoder, {(group, "private" | "shared"): DataFrame [genes, latent dims]}
        with the reference's column names ``Z_private_{n}`` / ``Z_shared_{n}`` and the group's own var_names as the index
        (``adata.var_names``, or ``adata.uns["groups_var_names"]``; a plain RangeIndex when the duck-typed input has neither)."""
        import pandas as pd

        out = {}
        var_idx = [np.asarray(v) for v in self.adata.uns["groups_var_indices"]]
        for i in range(len(self.module.input_dims)):
            names = None
            if getattr(self.adata, "var_names", None) is not None:
                names = np.asarray(self.adata.var_names)[var_idx[i]]
            elif self.adata.uns.get("groups_var_names") is not None:
                gv = self.adata.uns["groups_var_names"]
                names = np.asarray(gv[i] if not isinstance(gv, dict) else list(gv.values())[i])
            cols = {"private": [f"Z_private_{n}" for n in range(self.module.n_dimensions_private)],
                    "shared": [f"Z_shared_{n}" for n in range(self.module.n_dimensions_shared)]}
            for t in ("private", "shared"):
                out[(i, t)] = pd.DataFrame(self.module.get_loadings(i, t), index=names, columns=cols[t])
        return out

    # ------------------------------------------------------------------------------------------
    def save(self, dir_path: str, prefix: Optional[str] = None, overwrite: bool = False, save_anndata: bool = False, **anndata_write_kwargs) -> None:
        """``spvipes.save("spvipes_model")`` (docs/notebooks/Tutorial.ipynb:487; inherited scvi-tools 0.20.0 ``BaseModelClass.save``):
        creates ``dir_path`` (an existing one is an error unless ``overwrite``) and writes ``{prefix}model.pt`` =
        ``{"model_state_dict", "var_names", "attr_dict"}``.  The state_dict carries the reference's parameter names (BatchNorm
        buffers included), so the file's weights load into the reference module and back; ``attr_dict`` holds the constructor
        arguments (``init_params_``), the ``setup_anndata`` arguments, ``is_trained_`` and the training history."""
        import os

        if os.path.exists(dir_path) and not overwrite:
            raise ValueError(f"{dir_path} already exists. Please provide an unexisting directory for saving.")
        if save_anndata:
            raise NotImplementedError("save_anndata=True needs anndata's writer, which is outside this build: keep the AnnData yourself and pass it to load()")
        os.makedirs(dir_path, exist_ok=True)
        var_names = getattr(self.adata, "var_names", None)
        attr = {"init_params_": dict(self.init_params_), "setup_args_": dict(self._setup), "is_trained_": bool(self.is_trained_),
                "history_": {k: list(v) for k, v in self.history.items()}}
        torch.save({"model_state_dict": {k: v.detach().cpu() for k, v in self.module.state_dict().items()},
                    "var_names": None if var_names is None else np.asarray(var_names).astype(str), "attr_dict": attr},
                   os.path.join(dir_path, f"{prefix or ''}model.pt"))

    @classmethod
    def load(cls, dir_path: str, adata=None, use_gpu=None, prefix: Optional[str] = None, backup_url: Optional[str] = None,
             device: Optional[str] = None) -> "spVIPES":
        """``spVIPES.load("spvipes_model", adata=adata)`` (Tutorial.ipynb:538; scvi-tools ``BaseModelClass.load``): rebuilds the
        model from the saved constructor arguments on ``adata`` (registering it with the saved ``setup_anndata`` arguments when
        it has not been set up), loads the state_dict, leaves the module in eval mode.  ``adata`` is required (the AnnData is
        never written by this build); ``use_gpu`` is accepted and ignored, ``backup_url`` is not supported (no network)."""
        import os
        import warnings

        path = os.path.join(dir_path, f"{prefix or ''}model.pt")
        if not os.path.exists(path):
            raise ValueError(f"Failed to load model file at {path}. If attempting to load a saved model from <v0.15.0, please use the util function `convert_legacy_save` to convert to an updated format.")
        if adata is None:
            raise ValueError("Save path contains no saved anndata and no adata was passed.")
        blob = torch.load(path, map_location="cpu", weights_only=False)
        attr = blob["attr_dict"]
        if _SETUP_KEY not in adata.uns:
            cls.setup_anndata(adata, **attr["setup_args_"])
        saved_names, names = blob.get("var_names"), getattr(adata, "var_names", None)
        if saved_names is not None and names is not None and not np.array_equal(np.asarray(names).astype(str), saved_names):
            warnings.warn("var_names for adata passed in does not match var_names of adata used to train the model. "
                          "For valid results, the vars need to be the same and in the same order as the adata used to train the model.")
        init = dict(attr["init_params_"])
        if device is not None:
            init["device"] = device
        model = cls(adata, **init)
        model.module.load_state_dict(blob["model_state_dict"])
        model.module.eval()
        model.is_trained_ = bool(attr["is_trained_"])
        model.history = {k: list(v) for k, v in attr.get("history_", {}).items()}
        return model
