"""CPU restatement (numpy, loops) of the reference's HOST-side semantics around the hot path -- SURVEY.md section 8(f-1, f-2):
which cells meet in which minibatch, and how per-batch latents are assembled into the arrays the user gets.

TEST INFRASTRUCTURE: imported by tests/ only (the product's counterparts are spvipes_amd/data.py and
spvipes_amd/model.py).  Parity status: PINNED by reference-run fixtures -- tests/golden/make_host_goldens.py loads the
reference's own dataloaders/_ann_dataloader.py, dataloaders/_concat_dataloader.py, data/_multi_datasplitter.py,
model/base/training_mixin.py and model/spvipes.py by path (third-party imports stubbed, a fake AnnDataManager behind the
loaders, a recording module behind get_latent_representation) and records the split, two epochs of the train loader, the
steps and the assembled arrays of get_latent_representation for 83 cases, and what train() hands to TrainingPlan /
TrainRunner (tests/golden/host_*.npz); tests/test_host_semantics.py holds every function below to them.  Still a stand-in:
scvi-tools' ``validate_data_split`` (oracle/scvi_standins.py).  Citations are into /root/reference/src/spVIPES.
"""
from __future__ import annotations

import math
from itertools import cycle
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np


from .scvi_standins import validate_data_split  # noqa: E402,F401  (third-party scvi-tools 0.20.0: stand-in, see there)


def split_groups(group_indices_list: Sequence[Sequence[int]], train_size: float, validation_size: Optional[float], seed: int) -> Dict[str, List[np.ndarray]]:
    """MultiGroupDataSplitter.setup (data/_multi_datasplitter.py:65-79): ONE RandomState(seed) permutes the groups in order;
    validation cells first, then training cells, the rest is the test set."""
    rs = np.random.RandomState(seed=seed)
    out = {"train": [], "val": [], "test": []}
    for idx in group_indices_list:
        n_train, n_val = validate_data_split(len(idx), train_size, validation_size)
        perm = rs.permutation(np.asarray(idx))
        out["val"].append(perm[:n_val])
        out["train"].append(perm[n_val:n_val + n_train])
        out["test"].append(perm[n_val + n_train:])
    return out


def loader_batches(indices: Sequence[int], batch_size: int, drop_last: bool, order: Optional[Sequence[int]] = None) -> List[np.ndarray]:
    """One AnnDataLoader (dataloaders/_ann_dataloader.py:86-92): a BatchSampler over ``indices`` visited in ``order`` (the
    sampler's permutation when shuffle=True; sequential when None), last partial batch kept unless drop_last."""
    idx = np.asarray(indices)
    if order is not None:
        idx = idx[np.asarray(order)]
    out = [idx[lo:lo + batch_size] for lo in range(0, len(idx), batch_size)]
    if drop_last and out and len(out[-1]) < batch_size:
        out.pop()
    return out


def concat_loader_steps(indices_list: Sequence[Sequence[int]], batch_size: int, drop_last: bool,
                        orders: Optional[Sequence[Optional[Sequence[int]]]] = None) -> List[Tuple[np.ndarray, ...]]:
    """ConcatDataLoader.__iter__ (dataloaders/_concat_dataloader.py:101-110): one loader per group; the loader with the most
    batches (the FIRST one on ties: np.argmax, :101-102) leads, every other loader is wrapped in itertools.cycle -- which
    caches the batches of its first pass and replays THOSE when it is exhausted (no reshuffle inside an epoch) -- and the
    step tuples are zip(*loaders)."""
    orders = orders or [None] * len(indices_list)
    loaders = [loader_batches(ix, batch_size, drop_last, od) for ix, od in zip(indices_list, orders)]
    lens = [len(l) for l in loaders]
    largest = int(np.argmax(lens))
    iters = [iter(l) if i == largest else cycle(l) for i, l in enumerate(loaders)]
    return list(zip(*iters))   # (a group without a single batch makes cycle() empty and with it the whole epoch, as in the reference)


def loader_mode(use_labels: bool, labels_registered: bool, use_transport_plan: bool, pair_data: bool,
                drop_last: Optional[bool]) -> Tuple[bool, bool]:
    """get_latent_representation's choice of (drop_last, use_cycling) (model/spvipes.py:468-503): every branch of the
    ``drop_last is None`` ladder sets False (:470-480); cycling = a transport plan, pair_data, no dropped batch, and not the
    label-based PoE (:497-503)."""
    if drop_last is None:
        drop_last = False
    label_poe = bool(use_labels and labels_registered)
    return bool(drop_last), bool(use_transport_plan and pair_data and not drop_last and not label_poe)


def default_max_epochs(n_obs: int) -> int:
    """MultiGroupTrainingMixin.train (model/base/training_mixin.py:89-91)."""
    return np.min([round((20000 / n_obs) * 400), 400]).item()


def cycling_chunks(group_indices_list: Sequence[Sequence[int]]) -> List[Tuple[List[int], List[int]]]:
    """_process_all_cells_with_cycling (model/spvipes.py:578-605): chunks of min_group_size cells per group, the shorter
    group's indices wrapping around (modulo)."""
    n0, n1 = len(group_indices_list[0]), len(group_indices_list[1])
    mn, mx = min(n0, n1), max(n0, n1)
    if mn == 0:
        raise ValueError("One of the groups is empty")
    out = []
    for start in range(0, mx, mn):
        c0 = [group_indices_list[0][(start + i) % n0] for i in range(mn)]
        c1 = [group_indices_list[1][(start + i) % n1] for i in range(mn)]
        out.append((c0, c1))
    return out


def latent_steps(group_indices_list: Sequence[Sequence[int]], batch_size: int, drop_last: bool, use_cycling: bool) -> List[Tuple[np.ndarray, np.ndarray]]:
    """The sequence of (group-0 batch, group-1 batch) index arrays that get_latent_representation feeds to
    module.inference (model/spvipes.py:497-523, :578-626): sequential ConcatDataLoader steps, per cycling chunk when the
    cycling path is taken."""
    if use_cycling:
        steps: List[Tuple[np.ndarray, np.ndarray]] = []
        for c0, c1 in cycling_chunks(group_indices_list):
            steps += concat_loader_steps([c0, c1], batch_size, drop_last=False)
        return steps
    return concat_loader_steps(group_indices_list, batch_size, drop_last)


def format_results(results: Dict[str, List[np.ndarray]], n_groups_1: int, n_groups_2: int) -> Dict[str, Dict[int, np.ndarray]]:
    """_format_results (model/spvipes.py:628-650): concatenate the per-step arrays, truncate to the group sizes, and
    reorder GROUP 1 ONLY (the second group) by argsort of its (truncated) 'indices' column."""
    cat = lambda k: np.concatenate(results[k])
    idx2 = cat("groups_2_original_indices").flatten()[:n_groups_2]
    p = {0: cat("groups_1_latent")[:n_groups_1], 1: cat("groups_2_latent")[:n_groups_2]}
    s = {0: cat("groups_1_latent_shared")[:n_groups_1], 1: cat("groups_2_latent_shared")[:n_groups_2]}
    order = np.argsort(idx2)
    return {"shared": s, "private": p, "shared_reordered": {0: s[0], 1: s[1][order]}, "private_reordered": {0: p[0], 1: p[1][order]}}
