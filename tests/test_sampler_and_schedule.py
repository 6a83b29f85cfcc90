"""CPU: host logic of the train loop -- sampler epoch semantics, KL warm-up, epochs heuristic."""
import pytest
import numpy as np
import torch

from spvipes_amd.data import MinibatchSampler
from spvipes_amd.train import default_max_epochs, kl_weight_at

CPU = torch.device("cpu")


def test_split_follows_randomstate_permutation():
    s = MinibatchSampler([10, 7], 2, CPU, seed=0, train_size=0.8)
    rs = np.random.RandomState(seed=0)
    p0, p1 = rs.permutation(np.arange(10)), rs.permutation(np.arange(7))
    assert s.val_idx[0].tolist() == p0[:2].tolist() and s.train_idx[0].tolist() == p0[2:].tolist()
    assert s.val_idx[1].tolist() == p1[:1].tolist() and s.train_idx[1].tolist() == p1[1:].tolist()  # ceil(0.8*7)=6 train


def test_epoch_length_drop_last_and_cycle_replay():
    s = MinibatchSampler([23, 9], 4, CPU, seed=1)
    assert s.batches_per_group == [5, 2] and s.steps_per_epoch == 5
    ep = list(s.epoch())
    assert len(ep) == 5 and all(b[0].numel() == 4 and b[1].numel() == 4 for b in ep)
    big = torch.cat([b[0] for b in ep])
    assert big.unique().numel() == 20  # drop_last: 20 of 23 cells, no repeats within the epoch
    # the smaller group replays the batches of its first pass, in order (itertools.cycle)
    assert torch.equal(ep[2][1], ep[0][1]) and torch.equal(ep[3][1], ep[1][1]) and torch.equal(ep[4][1], ep[0][1])
    ep2 = list(s.epoch())
    assert not torch.equal(torch.cat([b[0] for b in ep2]), big)  # reshuffled every epoch


def test_ranks_get_disjoint_shards():
    a = MinibatchSampler([40, 40], 4, CPU, seed=0, rank=0, world=2)
    b = MinibatchSampler([40, 40], 4, CPU, seed=0, rank=1, world=2)
    for g in range(2):
        assert len(a.train_idx[g]) == len(b.train_idx[g]) == 20
        assert not set(a.train_idx[g].tolist()) & set(b.train_idx[g].tolist())


def test_sequential_eval_order_keeps_last_partial_batch():
    out = list(MinibatchSampler.sequential([5, 3, 9, 1, 7], 2, CPU))
    assert [o.tolist() for o in out] == [[5, 3], [9, 1], [7]]


def test_kl_warmup_and_epoch_heuristic():
    assert kl_weight_at(0, 0, 400, None) == 0.0 and kl_weight_at(200, 0, 400, None) == 0.5 and kl_weight_at(999, 0, 400, None) == 1.0
    assert kl_weight_at(5, 50, 400, 100) == 5 / 400  # scvi-tools 0.20.0 _compute_kl_weight: the epoch criterion is checked first
    assert kl_weight_at(5, 50, None, 100) == 0.5 and kl_weight_at(5, 150, None, 100) == 1.0  # steps only without an epoch count
    assert kl_weight_at(3, 7, None, None) == 1.0
    assert default_max_epochs(50_000) == 160 and default_max_epochs(1_000) == 400  # training_mixin.py:89-91


@pytest.mark.parametrize("B", [64, 128, 500, 4096, 8192])
@pytest.mark.parametrize("G", [32, 2000, 10000, 20000, 30000])
def test_launch_splits_cover_every_gene_for_any_shape(B, G):
    """the host-side split arithmetic the decoder kernels trust (spv_dec_* reject anything else): gene splits are multiples of 32
    genes that cover the padded gene count, likelihood splits fit their LDS slice"""
    from spvipes_amd._abi import DEC_CELLS_PER_WG
    from spvipes_amd.ops import NB_GSPL_MAX, _gene_splits, _nb_splits
    Bp, Gp = -(-B // DEC_CELLS_PER_WG) * DEC_CELLS_PER_WG, -(-G // 32) * 32
    splits, per = _gene_splits(Bp, Gp)
    assert splits >= 1 and per % 32 == 0 and splits * per >= Gp and (splits - 1) * per < Gp
    nbs, nbper = _nb_splits(Gp)
    assert nbs >= 1 and nbper % 32 == 0 and nbper <= NB_GSPL_MAX and nbs * nbper >= Gp and (nbs - 1) * nbper < Gp
