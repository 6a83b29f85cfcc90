#!/usr/bin/env python3
"""Generate golden vectors for the HOST-side semantics around the hot path by running the
REFERENCE's own loaders / splitter / model methods on CPU.

Runs only in the build container (needs /root/reference; never on the GPU box, never from the test-suite).  It loads
the reference's unmodified

    dataloaders/_ann_dataloader.py      (BatchSampler over Random / SequentialSampler, drop_last)
    dataloaders/_concat_dataloader.py   (largest loader leads: np.argmax, the others in itertools.cycle; zip)
    data/_multi_datasplitter.py         (RandomState(settings.seed).permutation split, train loader shuffle + drop_last)
    model/base/training_mixin.py        (max_epochs heuristic, KL warm-up kwargs)
    model/spvipes.py                    (get_latent_representation: drop_last / cycling choice, _process_batches,
                                         _process_all_cells_with_cycling, _format_results)

BY PATH after registering stand-ins for their third-party imports (scvi-tools, Lightning, anndata, scanpy: absent from
this image and not under /root/reference) and a fake ``AnnDataManager`` whose torch dataset returns, for a batch of
positions, the cells' global row numbers and their ``obs["indices"]`` values.  A recording module in place of
``spVIPESmodule`` returns latents that are functions of the cell's row number, so the arrays the reference assembles
say exactly which cell went where.  Only data is committed (tests/golden/host_*.npz) -- no reference source travels.

Third-party pieces that stay stand-ins (parity unpinned by the reference): scvi-tools 0.20.0 ``validate_data_split``
(oracle/scvi_standins.py) and torch's own samplers (real torch here).

    python tests/golden/make_host_goldens.py          # writes tests/golden/host_*.npz
"""
from __future__ import annotations

import contextlib
import importlib.util
import io
import os
import sys
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("SPVIPES_REFERENCE", "/root/reference")
sys.path.insert(0, ROOT)

from oracle import scvi_standins  # noqa: E402


# ------------------------------------------------------------------------------------------------------------------
# fakes for what the loaders read: an AnnData-like object, the manager, the torch dataset
# ------------------------------------------------------------------------------------------------------------------
class FakeAdata:
    def __init__(self, n_obs, obs_indices, groups_obs_indices):
        self.n_obs, self.shape = n_obs, (n_obs, 4)
        self.uns = {"groups_obs_indices": groups_obs_indices}
        self.obs_indices_column = np.asarray(obs_indices, np.float32)


class FakeDataset:
    """What scvi's AnnTorchDataset does for these loaders: ``ds[list of positions]`` -> dict of [b, 1] float32 arrays."""

    def __init__(self, adata, indices):
        self.adata, self.rows = adata, np.asarray(indices)

    def __len__(self):
        return len(self.rows)

    def __getitem__(self, pos):
        r = self.rows[np.asarray(pos)]
        return {"cell": r.astype(np.float32).reshape(-1, 1), "indices": self.adata.obs_indices_column[r].reshape(-1, 1)}


class FakeManager:
    def __init__(self, adata, registry=()):
        self.adata, self.data_registry = adata, {k: None for k in registry}

    def create_torch_dataset(self, indices=None, data_and_attributes=None):
        return FakeDataset(self.adata, indices)


class Recorded:
    plans, runners = [], []


def install_stubs():
    scvi_standins.install()
    scvi = sys.modules["scvi"]
    mods = {}

    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        mods[name] = m
        return m

    settings = types.SimpleNamespace(seed=0, dl_pin_memory_gpu_training=False, batch_size=128)
    scvi.settings = settings
    mod("scvi.dataloaders")
    mod("scvi.dataloaders._data_splitting", validate_data_split=scvi_standins.validate_data_split)
    mod("scvi.model")
    mod("scvi.model._utils", parse_use_gpu_arg=lambda use_gpu, return_device=True: ("cpu", None, torch.device("cpu")))

    class BaseModelClass:   # the two inherited members get_latent_representation touches
        def _validate_anndata(self, adata=None):
            return self.adata if adata is None else adata

    mod("scvi.model.base", BaseModelClass=BaseModelClass)
    mod("scvi.data")
    mod("scvi.data.fields", CategoricalObsField=object, LayerField=object)
    mod("scvi.utils", setup_anndata_dsp=types.SimpleNamespace(dedent=lambda f: f))

    class TrainingPlan:
        def __init__(self, module, **kw):
            Recorded.plans.append(dict(kw))

    class TrainRunner:
        def __init__(self, model, training_plan, data_splitter, max_epochs, use_gpu=None, **kw):
            Recorded.runners.append({"max_epochs": max_epochs, "data_splitter": data_splitter, **kw})

        def __call__(self):
            return None

    mod("scvi.train", TrainingPlan=TrainingPlan, TrainRunner=TrainRunner)

    class LightningDataModule:
        def __init__(self):
            pass

    mod("pytorch_lightning", LightningDataModule=LightningDataModule)
    mod("anndata", AnnData=FakeAdata)
    mod("scanpy")
    for name in ("spVIPES", "spVIPES.nn", "spVIPES.module", "spVIPES.dataloaders", "spVIPES.model", "spVIPES.model.base"):
        if name not in sys.modules:
            m = mod(name)
            m.__path__ = []
    d = mod("spVIPES.data", AnnDataManager=FakeManager)
    d.__path__ = []
    sys.modules.update(mods)
    return settings


def load_reference():
    src = os.path.join(REF, "src", "spVIPES")

    def load(mod_name, rel):
        spec = importlib.util.spec_from_file_location(mod_name, os.path.join(src, rel))
        m = importlib.util.module_from_spec(spec)
        sys.modules[mod_name] = m
        spec.loader.exec_module(m)
        return m

    load("spVIPES.nn.utils", "nn/utils.py")
    load("spVIPES.nn.networks", "nn/networks.py")
    load("spVIPES.module.spVIPESmodule", "module/spVIPESmodule.py")
    load("spVIPES.dataloaders._ann_dataloader", "dataloaders/_ann_dataloader.py")
    concat = load("spVIPES.dataloaders._concat_dataloader", "dataloaders/_concat_dataloader.py")
    split = load("spVIPES.data._multi_datasplitter", "data/_multi_datasplitter.py")
    mixin = load("spVIPES.model.base.training_mixin", "model/base/training_mixin.py")
    model = load("spVIPES.model.spvipes", "model/spvipes.py")
    return concat, split, mixin, model


def ragged(out, key, arrays):
    """a list of 1-D arrays as values + offsets"""
    out[key + "/values"] = np.concatenate([np.asarray(a).reshape(-1) for a in arrays]).astype(np.int64) if arrays else np.zeros(0, np.int64)
    out[key + "/offsets"] = np.cumsum([0] + [np.asarray(a).size for a in arrays]).astype(np.int64)


def make_adata(n0, n1, rng):
    """two groups interleaved in one AnnData (so that row numbers, group positions and obs['indices'] all differ)"""
    n = n0 + n1
    rows = rng.permutation(n)
    g0, g1 = np.sort(rows[:n0]), np.sort(rows[n0:])
    col = np.zeros(n, np.float32)
    col[g0] = rng.permutation(n0)          # prepare_adatas writes a within-group id; any permutation exercises the argsort
    col[g1] = rng.permutation(n1)
    return FakeAdata(n, col, [g0, g1]), [g0.tolist(), g1.tolist()]


# ------------------------------------------------------------------------------------------------------------------
def splitter_cases(split_mod, settings, out):
    """MultiGroupDataSplitter: the split, then two epochs of the train loader (shuffle, drop_last, cycle)."""
    cases = [((40, 33), 0.9, None, 0, 4), ((101, 57), 0.8, 0.1, 3, 8), ((16, 16), 1.0, None, 7, 4), ((250, 31), 0.5, 0.25, 11, 16),
             ((23, 9), 1.0, None, 1, 4), ((9, 50), 1.0, None, 5, 3), ((37, 37), 1.0, None, 2, 5), ((64, 23), 0.9, None, 4, 8)]
    out["split/n_cases"] = np.int64(len(cases))
    for ci, (sizes, train_size, val_size, seed, B) in enumerate(cases):
        rng = np.random.default_rng(100 + ci)
        adata, gi = make_adata(sizes[0], sizes[1], rng)
        settings.seed = seed
        sp = split_mod.MultiGroupDataSplitter(FakeManager(adata), group_indices_list=gi, train_size=train_size,
                                              validation_size=val_size, batch_size=B)
        sp.setup()
        k = f"split/{ci}"
        out[k + "/sizes"], out[k + "/batch_size"], out[k + "/seed"] = np.asarray(sizes, np.int64), np.int64(B), np.int64(seed)
        out[k + "/train_size"] = np.float64(train_size)
        out[k + "/validation_size"] = np.float64(-1.0 if val_size is None else val_size)
        for g in range(2):
            out[f"{k}/group{g}"] = np.asarray(gi[g], np.int64)
            out[f"{k}/train{g}"] = np.asarray(sp.train_idx[g], np.int64)
            out[f"{k}/val{g}"] = np.asarray(sp.val_idx[g], np.int64)
            out[f"{k}/test{g}"] = np.asarray(sp.test_idx[g], np.int64)
        dl = sp.train_dataloader()
        out[k + "/len"] = np.int64(len(dl))
        torch.manual_seed(1000 + ci)
        for ep in range(2):
            steps = [(b0["cell"].numpy().reshape(-1), b1["cell"].numpy().reshape(-1)) for b0, b1 in dl]
            ragged(out, f"{k}/epoch{ep}/g0", [s[0] for s in steps])
            ragged(out, f"{k}/epoch{ep}/g1", [s[1] for s in steps])
        has_val = all(len(v) > 0 for v in sp.val_idx)
        out[k + "/has_val_loader"] = np.int64(sp.val_dataloader() is not None)
        assert bool(out[k + "/has_val_loader"]) == has_val


class RecordingModule:
    """In place of spVIPESmodule for get_latent_representation: latents are functions of the cell's row number."""

    def __init__(self, use_labels, use_transport_plan, pair_data):
        self.use_labels, self.use_transport_plan, self.pair_data = use_labels, use_transport_plan, pair_data
        self.steps = []

    def _get_inference_input(self, tensors_by_group):
        return {"t": tensors_by_group}

    def inference(self, t):
        self.steps.append(tuple(d["cell"].numpy().reshape(-1).astype(np.int64) for d in t))
        poe, prv = {}, {}
        for g, d in enumerate(t):
            c = d["cell"].reshape(-1).double()
            shared = torch.stack([c, 2 * c + 1 + g, -c], 1)
            private = torch.stack([3 * c + g, c * c], 1)
            poe[g] = OrderedDict(logtheta_loc=None, logtheta_logvar=None, logtheta_scale=None, logtheta_qz=None,
                                 logtheta_log_z=shared, logtheta_theta=None)
            prv[g] = OrderedDict(logtheta_loc=None, logtheta_logvar=None, logtheta_scale=None, log_z=private, theta=None, qz=None)
        return {"poe_stats": poe, "private_stats": prv}


def latent_cases(model_mod, out):
    """spVIPES.get_latent_representation on the recording module: which steps it runs (sequential loaders, cycling for the
    paired PoE) and what it returns after truncation and the group-1 reorder."""
    modes = {"label": (True, False, False, ("labels",)), "paired": (False, True, True, ()), "cluster": (False, True, False, ()),
             "label_and_plan": (True, True, True, ("labels",))}
    cases = []
    for n0, n1, B in [(10, 10, 4), (10, 7, 4), (7, 10, 4), (12, 12, 4), (5, 23, 8), (23, 5, 8), (3, 3, 8), (64, 17, 16), (17, 64, 16)]:
        for mode in ("label", "paired", "cluster"):
            for dl in (None, False, True):
                cases.append((n0, n1, B, mode, dl))
    cases += [(11, 7, 4, "label_and_plan", None), (9, 9, 128, "paired", None)]
    out["latent/n_cases"] = np.int64(len(cases))
    for ci, (n0, n1, B, mode, dl) in enumerate(cases):
        rng = np.random.default_rng(500 + ci)
        adata, gi = make_adata(n0, n1, rng)
        # the user may pass the groups' cells in any order: shuffle group 1's list in some cases
        if ci % 3 == 1:
            gi = [gi[0], list(rng.permutation(gi[1]))]
        use_labels, use_plan, pair, registry = modes[mode]
        m = object.__new__(model_mod.spVIPES)
        m.adata, m.adata_manager = adata, FakeManager(adata, registry)
        m.module = RecordingModule(use_labels, use_plan, pair)
        k = f"latent/{ci}"
        err = ""
        try:
            with contextlib.redirect_stdout(io.StringIO()):
                res = m.get_latent_representation(gi, batch_size=B, drop_last=dl)
        except ValueError as e:   # no step at all (every batch dropped): torch.cat of an empty list in _format_results (:630)
            err, res = type(e).__name__, None
        out[k + "/error"] = np.array(err)
        out[k + "/n"], out[k + "/batch_size"] = np.asarray([n0, n1], np.int64), np.int64(B)
        out[k + "/mode"] = np.array(mode)
        out[k + "/drop_last"] = np.int64(-1 if dl is None else int(dl))
        out[k + "/obs_indices_column"] = adata.obs_indices_column
        for g in range(2):
            out[f"{k}/group{g}"] = np.asarray(gi[g], np.int64)
            ragged(out, f"{k}/steps_g{g}", [s[g] for s in m.module.steps])
        for name in ("shared", "private", "shared_reordered", "private_reordered"):
            for g in range(2):
                if res is not None:
                    out[f"{k}/{name}{g}"] = np.asarray(res[name][g], np.float64)


def mixin_cases(mixin_mod, out):
    """MultiGroupTrainingMixin.train: what reaches TrainingPlan / TrainRunner."""
    rows = []
    for n_obs, max_epochs, n_steps, n_epochs, es in [(50_000, None, None, 400, False), (1_000, None, None, 400, False), (20_000, None, 100, 400, True),
                                                     (400_000, None, None, None, False), (33_333, None, 7, 12, False), (1_000_000, None, None, 400, False),
                                                     (19_999, None, None, 400, False), (5_000, 17, None, 400, False)]:
        rng = np.random.default_rng(1)
        adata, gi = make_adata(8, 8, rng)
        adata.n_obs = n_obs
        obj = object.__new__(mixin_mod.MultiGroupTrainingMixin)
        obj.adata, obj.adata_manager, obj.module = adata, FakeManager(adata), None
        Recorded.plans.clear(), Recorded.runners.clear()
        obj.train(gi, batch_size=4, max_epochs=max_epochs, n_steps_kl_warmup=n_steps, n_epochs_kl_warmup=n_epochs, early_stopping=es,
                  plan_kwargs={"lr": 5e-4} if n_obs == 33_333 else None)
        p, r = Recorded.plans[0], Recorded.runners[0]
        rows.append([n_obs, -1 if max_epochs is None else max_epochs, r["max_epochs"], -1 if p["n_steps_kl_warmup"] is None else p["n_steps_kl_warmup"],
                     -1 if p["n_epochs_kl_warmup"] is None else p["n_epochs_kl_warmup"], int(r["early_stopping"]), int("lr" in p),
                     int(r["data_splitter"].train_size * 1000), r["data_splitter"].data_loader_kwargs["batch_size"]])
    out["mixin/columns"] = np.array("n_obs,max_epochs_arg,max_epochs_used,plan_n_steps_kl_warmup,plan_n_epochs_kl_warmup,early_stopping,plan_has_lr,"
                                    "splitter_train_size_permille,splitter_batch_size")
    out["mixin/rows"] = np.asarray(rows, np.int64)


def main():
    settings = install_stubs()
    concat, split_mod, mixin_mod, model_mod = load_reference()
    out = {}
    splitter_cases(split_mod, settings, out)
    path = os.path.join(HERE, "host_split_and_epochs.npz")
    np.savez_compressed(path, **out)
    print(f"{len(out)} arrays -> {os.path.relpath(path, ROOT)} ({os.path.getsize(path) / 1024:.0f} KiB)")
    out = {}
    latent_cases(model_mod, out)
    path = os.path.join(HERE, "host_latent_assembly.npz")
    np.savez_compressed(path, **out)
    print(f"{len(out)} arrays -> {os.path.relpath(path, ROOT)} ({os.path.getsize(path) / 1024:.0f} KiB)")
    out = {}
    mixin_cases(mixin_mod, out)
    path = os.path.join(HERE, "host_train_kwargs.npz")
    np.savez_compressed(path, **out)
    print(f"{len(out)} arrays -> {os.path.relpath(path, ROOT)} ({os.path.getsize(path) / 1024:.0f} KiB)")


if __name__ == "__main__":
    main()
