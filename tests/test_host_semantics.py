"""CPU: the product's sampler / batching / result-assembly logic (spvipes_amd/data.py) against the restatement of the
reference's host-side semantics in oracle/host_semantics.py (data/_multi_datasplitter.py:65-98,
dataloaders/_concat_dataloader.py:101-110, model/spvipes.py:497-523, :578-650)."""
import numpy as np
import pytest
import torch

from oracle import host_semantics as HS
from spvipes_amd.data import MinibatchSampler, format_latent_results, latent_loader_mode, latent_steps

CPU = torch.device("cpu")


@pytest.mark.parametrize("sizes,train_size,val_size,seed", [((40, 33), 0.9, None, 0), ((101, 57), 0.8, 0.1, 3), ((16, 16), 1.0, None, 7), ((250, 31), 0.5, 0.25, 11)])
def test_train_validation_split_matches_reference_splitter(sizes, train_size, val_size, seed):
    gi = [np.arange(100, 100 + sizes[0]), np.arange(1000, 1000 + sizes[1])]
    want = HS.split_groups(gi, train_size, val_size, seed)
    s = MinibatchSampler(list(sizes), 4, CPU, seed=seed, train_size=train_size, validation_size=val_size, group_indices_list=gi)
    for g in range(2):
        assert s.train_idx[g].tolist() == want["train"][g].tolist()
        assert s.val_idx[g].tolist() == want["val"][g].tolist()


@pytest.mark.parametrize("sizes,B", [((40, 16), 4), ((16, 40), 4), ((37, 37), 5), ((64, 23), 8), ((9, 50), 3)])
def test_training_epoch_order_replays_the_shorter_groups_first_pass(sizes, B):
    """drop_last batches; the group with the most batches leads; the other one replays the batches of its FIRST pass
    (itertools.cycle caches them) -- given the same visiting order per group."""
    s = MinibatchSampler(list(sizes), B, CPU, seed=1)
    rng = np.random.default_rng(5)
    orders = [rng.permutation(len(t)) for t in s.train_idx]
    perms = [torch.as_tensor(t[o], dtype=torch.int32) for t, o in zip(s.train_idx, orders)]
    got = [[r.tolist() for r in step] for step in s.epoch_from_permutations(perms)]
    want = HS.concat_loader_steps(s.train_idx, B, drop_last=True, orders=orders)
    assert len(got) == len(want) == max(n // B for n in sizes)
    for a, b in zip(got, want):
        assert a[0] == b[0].tolist() and a[1] == b[1].tolist()
    # a new epoch draws new permutations (the reference builds new iterators, hence a fresh shuffle and a fresh cycle)
    e1 = [r.tolist() for step in s.epoch() for r in step]
    e2 = [r.tolist() for step in s.epoch() for r in step]
    assert e1 != e2


@pytest.mark.parametrize("n0,n1,B", [(10, 10, 4), (10, 7, 4), (7, 10, 4), (12, 12, 4), (5, 23, 8), (23, 5, 8), (3, 3, 8), (64, 17, 16)])
@pytest.mark.parametrize("drop_last", [False, True])
@pytest.mark.parametrize("cycling", [False, True])
def test_latent_steps_follow_the_reference_loaders(n0, n1, B, drop_last, cycling):
    gi = [list(range(50, 50 + n0)), list(range(500, 500 + n1))]
    if cycling and drop_last:
        pytest.skip("the reference only cycles with drop_last=False (model/spvipes.py:497-503)")
    got = latent_steps(gi, B, drop_last, cycling)
    want = HS.latent_steps(gi, B, drop_last, cycling)
    assert len(got) == len(want)
    for a, b in zip(got, want):
        assert np.asarray(a[0]).tolist() == np.asarray(b[0]).tolist() and np.asarray(a[1]).tolist() == np.asarray(b[1]).tolist()


def test_format_results_truncates_and_reorders_group_1_only():
    rng = np.random.default_rng(0)
    n0, n1, d = 11, 7, 3
    steps = HS.latent_steps([list(range(n0)), list(range(n1))], 4, False, False)
    idx1_all = rng.permutation(40)[:n1]   # the 'indices' column of group 1's cells
    res = {k: [] for k in ("groups_1_latent", "groups_2_latent", "groups_1_latent_shared", "groups_2_latent_shared", "groups_2_original_indices")}
    for r0, r1 in steps:
        res["groups_1_latent"].append(rng.normal(size=(len(r0), d)))
        res["groups_2_latent"].append(rng.normal(size=(len(r1), d)))
        res["groups_1_latent_shared"].append(rng.normal(size=(len(r0), d + 1)))
        res["groups_2_latent_shared"].append(rng.normal(size=(len(r1), d + 1)))
        res["groups_2_original_indices"].append(idx1_all[np.asarray(r1)][:, None].astype(np.float32))
    want = HS.format_results(res, n0, n1)
    got = format_latent_results(res["groups_1_latent"], res["groups_2_latent"], res["groups_1_latent_shared"], res["groups_2_latent_shared"],
                                res["groups_2_original_indices"], n0, n1)
    for k in want:
        for g in (0, 1):
            np.testing.assert_array_equal(got[k][g], want[k][g])
    assert got["shared"][0].shape == (n0, d + 1) and got["private"][1].shape == (n1, d)
    assert np.array_equal(got["private_reordered"][0], got["private"][0])          # group 0 is never reordered
    assert not np.array_equal(got["private_reordered"][1], got["private"][1])     # group 1 is sorted by its indices


# ======================================================================================================================
# reference-run fixtures (tests/golden/host_*.npz, recorded from the reference's OWN loaders / splitter / model methods by
# tests/golden/make_host_goldens.py): they pin oracle/host_semantics.py, and the product's host logic is held to them too
# ======================================================================================================================
from tests._golden import host_golden, unragged  # noqa: E402

_SPLIT = host_golden("split_and_epochs")
_LATENT = host_golden("latent_assembly")
_MIXIN = host_golden("train_kwargs")


@pytest.mark.parametrize("ci", range(int(_SPLIT["split/n_cases"])))
def test_split_matches_the_reference_run_splitter(ci):
    """MultiGroupDataSplitter.setup() of the reference (data/_multi_datasplitter.py:65-79) on interleaved groups."""
    k = f"split/{ci}"
    gi = [_SPLIT[f"{k}/group{g}"] for g in range(2)]
    val = float(_SPLIT[k + "/validation_size"])
    val = None if val < 0 else val
    seed, B, train_size = int(_SPLIT[k + "/seed"]), int(_SPLIT[k + "/batch_size"]), float(_SPLIT[k + "/train_size"])
    want = HS.split_groups(gi, train_size, val, seed)
    s = MinibatchSampler([len(g) for g in gi], B, CPU, seed=seed, train_size=train_size, validation_size=val, group_indices_list=gi)
    for g in range(2):
        for part in ("train", "val", "test"):
            assert want[part][g].tolist() == _SPLIT[f"{k}/{part}{g}"].tolist()
        assert s.train_idx[g].tolist() == _SPLIT[f"{k}/train{g}"].tolist() and s.val_idx[g].tolist() == _SPLIT[f"{k}/val{g}"].tolist()
    assert bool(_SPLIT[k + "/has_val_loader"]) == all(len(v) > 0 for v in s.val_idx)


def _orders_from_first_pass(train_idx, batches):
    """the visiting order of one loader, recovered from the batches of its first pass (cells that drop_last cut off go last)"""
    pos = {int(c): i for i, c in enumerate(train_idx)}
    seen = [pos[int(c)] for b in batches for c in b]
    return np.asarray(seen + [i for i in range(len(train_idx)) if i not in set(seen)])


@pytest.mark.parametrize("ci", range(int(_SPLIT["split/n_cases"])))
def test_training_epochs_match_the_reference_run_train_loader(ci):
    """Two epochs of ``MultiGroupDataSplitter.train_dataloader()`` (ConcatDataLoader(shuffle=True, drop_last=True),
    dataloaders/_concat_dataloader.py:101-110): step count, disjoint drop_last batches of the leading group, the other
    group REPLAYING the batches of its first pass, a fresh shuffle per epoch -- and the oracle / the product's sampler
    reproduce every step from the groups' visiting orders alone."""
    k = f"split/{ci}"
    B = int(_SPLIT[k + "/batch_size"])
    train = [_SPLIT[f"{k}/train{g}"] for g in range(2)]
    nb = [len(t) // B for t in train]
    assert int(_SPLIT[k + "/len"]) == max(nb)
    s = MinibatchSampler([len(t) for t in train], B, CPU, seed=0, group_indices_list=train)   # train_size 1: the split is the identity up to a permutation
    epochs = []
    for ep in range(2):
        steps = list(zip(unragged(_SPLIT, f"{k}/epoch{ep}/g0"), unragged(_SPLIT, f"{k}/epoch{ep}/g1")))
        assert len(steps) == max(nb)
        first = [[st[g] for st in steps[:nb[g]]] for g in range(2)]
        orders = [_orders_from_first_pass(train[g], first[g]) for g in range(2)]
        want = HS.concat_loader_steps(train, B, drop_last=True, orders=orders)
        assert len(want) == len(steps)
        for a, b in zip(want, steps):
            assert a[0].tolist() == b[0].tolist() and a[1].tolist() == b[1].tolist()
        # the product's sampler given the same visiting orders (its own training rows are a permutation of `train`)
        perms = [torch.as_tensor(train[g][orders[g]], dtype=torch.int32) for g in range(2)]
        s.batches_per_group, s.steps_per_epoch = nb, max(nb)
        got = [[r.tolist() for r in step] for step in s.epoch_from_permutations(perms)]
        assert got == [[b[0].tolist(), b[1].tolist()] for b in steps]
        epochs.append(np.concatenate([st[0] for st in steps]).tolist())
    assert epochs[0] != epochs[1]   # RandomSampler reshuffles when the loaders are iterated again


@pytest.mark.parametrize("ci", range(int(_LATENT["latent/n_cases"])))
def test_latent_steps_and_assembly_match_the_reference_run_get_latent_representation(ci):
    """``spVIPES.get_latent_representation`` of the reference (model/spvipes.py:424-650) on a recording module whose latents are
    functions of the cell's row number: the steps it ran, and the four arrays per group it returned."""
    k = f"latent/{ci}"
    mode, B = str(_LATENT[k + "/mode"]), int(_LATENT[k + "/batch_size"])
    dl = int(_LATENT[k + "/drop_last"])
    dl = None if dl < 0 else bool(dl)
    gi = [_LATENT[f"{k}/group{g}"] for g in range(2)]
    n0, n1 = (int(v) for v in _LATENT[k + "/n"])
    use_labels, use_plan, pair = {"label": (True, False, False), "paired": (False, True, True), "cluster": (False, True, False),
                                  "label_and_plan": (True, True, True)}[mode]
    want_mode = HS.loader_mode(use_labels, use_labels, use_plan, pair, dl)
    assert latent_loader_mode(use_labels, use_plan, pair, dl) == want_mode
    drop_last, cycling = want_mode
    ref_steps = list(zip(unragged(_LATENT, f"{k}/steps_g0"), unragged(_LATENT, f"{k}/steps_g1")))
    for fn in (HS.latent_steps, latent_steps):
        steps = fn([g.tolist() for g in gi], B, drop_last, cycling)
        assert len(steps) == len(ref_steps)
        for a, b in zip(steps, ref_steps):
            assert np.asarray(a[0]).tolist() == b[0].tolist() and np.asarray(a[1]).tolist() == b[1].tolist()
    if str(_LATENT[k + "/error"]):
        assert not ref_steps      # every batch dropped: the reference dies in torch.cat([]) (:630); so does np.concatenate([])
        with pytest.raises(ValueError):
            format_latent_results([], [], [], [], [], n0, n1)
        return
    col = _LATENT[k + "/obs_indices_column"]
    res = {kk: [] for kk in ("groups_1_latent", "groups_2_latent", "groups_1_latent_shared", "groups_2_latent_shared", "groups_2_original_indices")}
    for r0, r1 in ref_steps:      # the recording module's formulas (make_host_goldens.RecordingModule.inference)
        for g, r in ((0, r0.astype(np.float64)), (1, r1.astype(np.float64))):
            res[f"groups_{g + 1}_latent_shared"].append(np.stack([r, 2 * r + 1 + g, -r], 1))
            res[f"groups_{g + 1}_latent"].append(np.stack([3 * r + g, r * r], 1))
        res["groups_2_original_indices"].append(col[r1][:, None])
    want = HS.format_results(res, n0, n1)
    got = format_latent_results(res["groups_1_latent"], res["groups_2_latent"], res["groups_1_latent_shared"], res["groups_2_latent_shared"],
                                res["groups_2_original_indices"], n0, n1)
    for name in ("shared", "private", "shared_reordered", "private_reordered"):
        for g in (0, 1):
            np.testing.assert_array_equal(want[name][g], _LATENT[f"{k}/{name}{g}"])
            np.testing.assert_array_equal(got[name][g], _LATENT[f"{k}/{name}{g}"])


def test_train_kwargs_match_the_reference_run_training_mixin():
    """MultiGroupTrainingMixin.train with recording TrainingPlan / TrainRunner (model/base/training_mixin.py:89-123)."""
    from spvipes_amd.train import default_max_epochs
    cols = str(_MIXIN["mixin/columns"]).split(",")
    for row in _MIXIN["mixin/rows"]:
        r = dict(zip(cols, (int(v) for v in row)))
        if r["max_epochs_arg"] < 0:
            assert HS.default_max_epochs(r["n_obs"]) == r["max_epochs_used"] == default_max_epochs(r["n_obs"])
        else:
            assert r["max_epochs_used"] == r["max_epochs_arg"]
        assert r["splitter_batch_size"] == 4 and r["splitter_train_size_permille"] == 900   # batch_size reaches the loaders, train_size default 0.9
    # both warm-up arguments always reach the plan (update_dict, :93-101), user plan_kwargs survive beside them
    by_n = {int(r[0]): dict(zip(cols, (int(v) for v in r))) for r in _MIXIN["mixin/rows"]}
    assert by_n[33_333]["plan_n_steps_kl_warmup"] == 7 and by_n[33_333]["plan_n_epochs_kl_warmup"] == 12 and by_n[33_333]["plan_has_lr"] == 1
    assert by_n[400_000]["plan_n_epochs_kl_warmup"] == -1 and by_n[20_000]["early_stopping"] == 1 and by_n[50_000]["early_stopping"] == 0
