"""Sparse transport plans on device: the CSR kernels against dense torch indexing of the same plan."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from spvipes_amd import _abi
    _abi.load()  # raises if libspvipes_hip.so is missing: no fallback
    return torch.device("cuda:0")


def _random_plan(n0, n1, k, seed):
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n0), k)
    cols = rng.integers(0, n1, size=n0 * k)
    vals = rng.random(n0 * k).astype(np.float32) + 0.01
    m = sp.coo_matrix((vals, (rows, cols)), shape=(n0, n1)).tocsr()  # duplicates are summed
    m.data = m.data.astype(np.float32)
    return m


@pytest.mark.parametrize("n0,n1,B,k", [(300, 280, 64, 6), (5000, 4000, 512, 9), (50, 50, 50, 50)])
def test_plan_argmax_matches_dense_block(dev, n0, n1, B, k):
    from spvipes_amd import _abi
    from spvipes_amd._abi import ptr, stream_ptr
    from spvipes_amd.plan import SparsePlan
    m = _random_plan(n0, n1, k, seed=n0 + B)
    plan = SparsePlan.from_scipy(m, dev)
    rng = np.random.default_rng(B)
    idx0 = torch.tensor(rng.permutation(n0)[:B].astype(np.int32), device=dev)
    idx1 = torch.tensor(rng.permutation(n1)[:B].astype(np.int32), device=dev)
    plan.bind_minibatch(idx0, idx1)
    p0 = torch.empty(B, dtype=torch.int32, device=dev)
    p1 = torch.empty(B, dtype=torch.int32, device=dev)
    ps = plan.c_struct()
    _abi.call("spv_plan_argmax", C.byref(ps), ptr(idx0), ptr(idx1), ptr(plan.inv0), ptr(plan.inv1), B, B, ptr(p0), ptr(p1), stream_ptr())
    block = torch.tensor(m.toarray(), device=dev)[idx0.long()][:, idx1.long()]
    assert torch.equal(plan.dense_block(idx0, idx1), block)
    assert torch.equal(p0.long(), torch.argmax(block, dim=1))   # first maximum; all-zero rows -> 0
    assert torch.equal(p1.long(), torch.argmax(block, dim=0))
    assert int((block.sum(1) == 0).sum()) > 0 or k >= n1        # the small-k cases do exercise empty rows


def test_sparse_plan_from_dense_equals_from_scipy(dev):
    from spvipes_amd.plan import SparsePlan
    m = _random_plan(120, 90, 4, seed=3)
    a = SparsePlan.from_scipy(m, dev)
    b = SparsePlan.from_dense(torch.tensor(m.toarray()), dev)
    for x, y in ((a.ptr0, b.ptr0), (a.ind0, b.ind0), (a.val0, b.val0), (a.ptr1, b.ptr1), (a.ind1, b.ind1), (a.val1, b.val1)):
        assert torch.equal(x, y)


@pytest.mark.parametrize("mode", ["paired", "cluster"])
def test_module_sparse_plan_equals_dense_plan_and_trains(dev, mode):
    """The module fed a scipy.sparse plan (never densified) gives what it gives with the reference's dense tensor, on a
    data set whose dense plan would still fit; then a few optimiser steps run through the captured graph."""
    from spvipes_amd.data import make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    n, G, B = 3000, 400, 256
    groups = [make_synthetic_group(g, n, G, dev) for g in (0, 1)]
    m_sp = _random_plan(n, n, 8, seed=11)
    rng = np.random.default_rng(5)
    comps = [torch.tensor(rng.integers(0, 6, size=n).astype(np.float32), device=dev) for _ in (0, 1)]
    kw = dict(pair_data=(mode == "paired"), use_labels=False, n_hidden=64, n_dimensions_shared=12, n_dimensions_private=6, dropout_rate=0.0,
              precision="fp32")
    torch.manual_seed(0)
    mods = []
    for plan in (torch.tensor(m_sp.toarray()), m_sp):
        torch.manual_seed(0)
        mods.append(spVIPESmodule({0: G, 1: G}, transport_plan=plan, **kw).to(dev))
    mods[1].load_state_dict(mods[0].state_dict())
    rows = [torch.tensor(rng.permutation(n)[:B].astype(np.int32), device=dev) for _ in (0, 1)]
    trainers = [Trainer(m, [g.counts for g in groups], components=(comps if mode == "cluster" else None)) for m in mods]
    outs = []
    for tr, m in zip(trainers, mods):
        m.train()
        torch.manual_seed(123)
        _, _, lo = m(tr.minibatch(rows), loss_kwargs={"kl_weight": 1.0})
        outs.append(float(lo.loss))
    assert outs[0] == outs[1], outs
    tr = trainers[1]
    tr.capture(rows)
    losses = [float(tr.step(rows, kl_weight=1.0).loss) for _ in range(6)]
    assert all(np.isfinite(losses)) and losses[-1] < losses[0], losses
