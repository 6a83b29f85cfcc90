"""Device-resident data for the hot path: count matrices, minibatch sampler, synthetic generator.

Replaces the reference's host-side iterators
    data/_multi_datasplitter.py:65-98   per-group permutation split, train loader shuffle + drop_last
    dataloaders/_concat_dataloader.py:101-110  zip of per-group loaders, shorter ones cycled
    dataloaders/_ann_dataloader.py      batch sampler + AnnTorchDataset row densification
with an index sampler: the count matrices stay in HBM (uint16 when the counts are integral and
< 65536, else float32) and a minibatch is just an int32 row-index vector per group that the
kernels gather through.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np
import torch

from .ops import GroupCounts


def to_group_counts(X, device, prefer_u16: bool = True, col_off: int = 0, G: Optional[int] = None) -> GroupCounts:
    """Move a dense [cells, genes] count array into HBM (uint16 if lossless, else float32)."""
    if isinstance(X, torch.Tensor):
        Xn = X.detach().cpu().numpy()
    else:
        Xn = np.asarray(X)
    G = Xn.shape[1] - col_off if G is None else G
    if Xn.size and float(Xn.min()) < 0:
        raise ValueError("count matrices must be non-negative (the likelihood is evaluated at log1p(count); the reference gives NaN for counts <= -1)")
    if prefer_u16 and Xn.size and float(Xn.min()) >= 0 and float(Xn.max()) < 65536 and np.all(Xn == np.floor(Xn)):
        t = torch.from_numpy(np.ascontiguousarray(Xn.astype(np.uint16)).view(np.int16))
    else:
        t = torch.from_numpy(np.ascontiguousarray(Xn.astype(np.float32)))
    return GroupCounts(t.to(device), G, col_off, resident=True)


@dataclass
class SyntheticGroup:
    counts: GroupCounts
    labels: torch.Tensor  # float32 [n_cells] label codes (the reference carries labels as float32)


def make_synthetic_group(g: int, n_cells: int, n_genes: int, device, n_labels: int = 10, dtype: str = "u16",
                         chunk: int = 4096, shard: int = 0) -> SyntheticGroup:
    """SURVEY.md 8d synthetic counts: X[c,j] ~ NB(mean = s_c * exp(a[label(c), j]), dispersion 2) * Bernoulli(0.2),
    s_c ~ LogNormal(0, 0.25), a ~ N(0, 1), 10 labels (group 0 never has label 0, group 1 never has
    label n_labels-1 so the non-common-label branch of the PoE runs).  Small tables come from
    numpy.random.default_rng(1000 + g); the [cells, genes] draws are generated on the device with a
    torch generator seeded 1000 + g (a 5e8-entry numpy NB draw per group would dominate start-up).  ``shard`` (a rank of a
    data-parallel job) offsets both seeds by 10 per shard: ranks hold DIFFERENT cells, as a sharded data set does."""
    rng = np.random.default_rng(1000 + g + 10 * shard)
    allowed = [l for l in range(n_labels) if l != (0 if g == 0 else n_labels - 1)]
    labels = rng.choice(allowed, size=n_cells)
    a = rng.normal(size=(n_labels, n_genes)).astype(np.float32)
    s = rng.lognormal(0.0, 0.25, size=n_cells).astype(np.float32)
    gen = torch.Generator(device=device).manual_seed(1000 + g + 10 * shard)
    a_d, s_d = torch.tensor(a, device=device), torch.tensor(s, device=device)
    lab_d = torch.tensor(labels, device=device)
    out = torch.empty((n_cells, n_genes), dtype=torch.int16 if dtype == "u16" else torch.float32, device=device)
    for lo in range(0, n_cells, chunk):
        hi = min(lo + chunk, n_cells)
        mean = s_d[lo:hi, None] * torch.exp(a_d[lab_d[lo:hi]])
        lam = torch._standard_gamma(torch.full_like(mean, 2.0), generator=gen) * (mean / 2.0)  # Gamma(shape 2, mean `mean`)
        cnt = torch.poisson(lam, generator=gen)
        cnt = cnt * (torch.rand(cnt.shape, device=device, generator=gen) < 0.2)
        cnt[:, 0] += (cnt.sum(1) == 0)  # no empty cell: library = log(sum log1p(x)) must be finite
        cnt = cnt.clamp_(max=65535.0)
        out[lo:hi] = cnt.to(torch.int32).to(torch.int16) if dtype == "u16" else cnt
    return SyntheticGroup(GroupCounts(out, n_genes, 0, resident=True), torch.tensor(labels, dtype=torch.float32, device=device))


class MinibatchSampler:
    """Per-group row-index minibatches with the reference's epoch semantics.

    * split: ``np.random.RandomState(seed).permutation(group_indices)`` per group, validation rows
      first, then training rows (data/_multi_datasplitter.py:66-79; scvi ``validate_data_split``:
      n_train = ceil(train_size * n), n_val = n - n_train by default);
    * training epochs: every group reshuffled, ``drop_last=True``; the epoch has as many steps as
      the group with the most batches; a group that runs out REPLAYS the batches of its first pass
      in order (``itertools.cycle`` caches them, dataloaders/_concat_dataloader.py:108-110);
    * evaluation order: sequential, ``drop_last=False``.

    A rank of a data-parallel job keeps the contiguous shard [rank::world] of every group's
    training rows, so ranks draw disjoint minibatches."""

    def __init__(self, n_cells: Sequence[int], batch_size: int, device, seed: int = 0, train_size: float = 1.0,
                 validation_size: Optional[float] = None, rank: int = 0, world: int = 1,
                 group_indices_list: Optional[Sequence[Sequence[int]]] = None):
        self.batch_size, self.device = batch_size, device
        rs = np.random.RandomState(seed=seed)
        self.train_idx: List[np.ndarray] = []
        self.val_idx: List[np.ndarray] = []
        for g, n in enumerate(n_cells):
            idx = np.arange(n) if group_indices_list is None else np.asarray(group_indices_list[g])
            n_train = int(np.ceil(train_size * len(idx)))
            n_val = (len(idx) - n_train) if validation_size is None else int(np.floor(validation_size * len(idx)))
            perm = rs.permutation(idx)
            self.val_idx.append(perm[:n_val])
            tr = perm[n_val:n_val + n_train]
            if world > 1:
                per = len(tr) // world
                tr = tr[rank * per:(rank + 1) * per]
            self.train_idx.append(tr)
        self._train_dev = [torch.as_tensor(t, dtype=torch.int32, device=device) for t in self.train_idx]
        self._gen = torch.Generator(device=device).manual_seed(seed + 7919 * (rank + 1))
        self.batches_per_group = [len(t) // batch_size for t in self.train_idx]
        if min(self.batches_per_group) == 0:
            raise ValueError("batch_size larger than a group's training shard")
        self.steps_per_epoch = max(self.batches_per_group)

    def draw_permutations(self):
        """one fresh permutation of every group's training rows (the per-epoch reshuffle of the reference's train loaders)"""
        return [t[torch.randperm(len(t), device=self.device, generator=self._gen)] for t in self._train_dev]

    def epoch_from_permutations(self, perms):
        """The steps of one epoch given each group's visiting order: drop_last batches; the group with the most batches
        leads; a group that runs out REPLAYS the batches of its first pass in order (ConcatDataLoader wraps the shorter
        loaders in itertools.cycle, dataloaders/_concat_dataloader.py:108-110)."""
        B = self.batch_size
        for step in range(self.steps_per_epoch):
            out = []
            for g, p in enumerate(perms):
                i = step % self.batches_per_group[g]
                out.append(p[i * B:(i + 1) * B].contiguous())
            yield out

    def epoch(self):
        """Yields a list (one int32 [batch_size] row-index tensor per group) per training step."""
        yield from self.epoch_from_permutations(self.draw_permutations())

    @staticmethod
    def sequential(indices: Sequence[int], batch_size: int, device):
        idx = torch.as_tensor(np.asarray(indices), dtype=torch.int32, device=device)
        for lo in range(0, len(idx), batch_size):
            yield idx[lo:lo + batch_size].contiguous()


def latent_loader_mode(use_labels: bool, use_transport_plan: bool, pair_data: bool, drop_last: Optional[bool]):
    """(drop_last, use_cycling) as get_latent_representation decides them (model/spvipes.py:468-503): ``drop_last=None``
    means False for every PoE flavour; the cycling path is taken by the paired PoE only (a transport plan, ``pair_data``, no
    labels) and only when no batch is dropped."""
    drop_last = bool(drop_last) if drop_last is not None else False
    return drop_last, bool(use_transport_plan and pair_data and not drop_last and not use_labels)


def latent_steps(local: Sequence[Sequence[int]], batch_size: int, drop_last: bool, use_cycling: bool):
    """The (group-0 rows, group-1 rows) pairs get_latent_representation feeds to the module, in order
    (model/spvipes.py:497-523 and the cycling path :578-626): sequential batches, last partial batch kept unless
    ``drop_last``; the group with more batches leads (the first on ties) and the other one's batches are replayed
    cyclically; with ``use_cycling`` the cells are first cut into chunks of min(n0, n1) cells per group, the shorter group's
    cells wrapping around, and every chunk is batched on its own with the partial batch kept."""
    def batches(idx, dl):
        idx = np.asarray(idx)
        out = [idx[lo:lo + batch_size] for lo in range(0, len(idx), batch_size)]
        if dl and out and len(out[-1]) < batch_size:
            out.pop()
        return out

    n0, n1 = len(local[0]), len(local[1])
    if use_cycling:
        mn, mx = min(n0, n1), max(n0, n1)
        if mn == 0:
            raise ValueError("One of the groups is empty")
        a0, a1 = np.asarray(local[0]), np.asarray(local[1])
        chunks = [(a0[(s + np.arange(mn)) % n0], a1[(s + np.arange(mn)) % n1]) for s in range(0, mx, mn)]
        dl = False
    else:
        chunks, dl = [(local[0], local[1])], drop_last
    steps = []
    for c0, c1 in chunks:
        b0, b1 = batches(c0, dl), batches(c1, dl)
        if not b0 or not b1:
            continue   # (a loader without a batch ends the reference's zip() at once)
        n = max(len(b0), len(b1))   # the loader with the most batches leads, the other is replayed cyclically
        steps += [(b0[i % len(b0)], b1[i % len(b1)]) for i in range(n)]
    return steps


def format_latent_results(private0, private1, shared0, shared1, indices1, n0: int, n1: int) -> dict:
    """model/spvipes.py:628-650: per-step arrays concatenated, truncated to the group sizes; the ``*_reordered`` entries sort
    GROUP 1 ONLY by its 'indices' column (group 0 is returned as is)."""
    i1 = np.concatenate(indices1).flatten()[:n1]
    p = {0: np.concatenate(private0)[:n0], 1: np.concatenate(private1)[:n1]}
    s = {0: np.concatenate(shared0)[:n0], 1: np.concatenate(shared1)[:n1]}
    order = np.argsort(i1)
    return {"shared": s, "private": p, "shared_reordered": {0: s[0], 1: s[1][order]}, "private_reordered": {0: p[0], 1: p[1][order]}}
