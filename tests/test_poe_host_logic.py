"""CPU tests of the device-agnostic PoE pairing/fusion logic (tests/_torch_poe.py, test infrastructure) against the
golden vectors produced by the reference, and of the module's parameter layout."""
import pytest
import torch

from tests import _torch_poe as P
from tests._golden import ALL_CASES, Golden


def _shared(g: Golden, grp: int):
    return {k: g.t(f"out/shared_{grp}/{k}") for k in ("logtheta_loc", "logtheta_logvar", "logtheta_scale")}


@pytest.mark.parametrize("case", ALL_CASES)
def test_poe_matches_reference(case):
    g = Golden(case)
    shared = {0: _shared(g, 0), 1: _shared(g, 1)}
    noise = g.noise()
    if g.mode == "label":
        out = P.label_based_poe(shared, {0: g.t("in/labels0"), 1: g.t("in/labels1")}, noise)
    else:
        plan = g.t("in/plan")
        idx = [torch.tensor(g.raw["in/idx0"], dtype=torch.float32).unsqueeze(1), torch.tensor(g.raw["in/idx1"], dtype=torch.float32).unsqueeze(1)]
        block = P.batch_transport_plan(plan, idx)
        if g.mode == "paired":
            out = P.paired_poe(shared, block, noise)
        else:
            out = P.cluster_based_poe(shared, block, [g.t("in/comp0"), g.t("in/comp1")], noise)
    for grp in range(2):
        assert list(out[grp].keys()) == ["logtheta_loc", "logtheta_logvar", "logtheta_scale", "logtheta_qz", "logtheta_log_z", "logtheta_theta"]
        for k in ("logtheta_loc", "logtheta_logvar", "logtheta_scale", "logtheta_log_z", "logtheta_theta"):
            torch.testing.assert_close(out[grp][k], g.t(f"out/poe_{grp}/{k}"), rtol=1e-5, atol=2e-6, msg=lambda m: f"{case} poe_{grp}/{k}: {m}")


def test_paired_poe_requires_equal_batches():
    s = lambda n: {"logtheta_loc": torch.zeros(n, 2), "logtheta_logvar": torch.zeros(n, 2), "logtheta_scale": torch.ones(n, 2)}
    with pytest.raises(AssertionError):
        P.paired_poe({0: s(3), 1: s(4)}, torch.zeros(3, 4), {})


def test_kl_closed_form():
    loc, scale = torch.randn(5, 3), torch.rand(5, 3) + 0.1
    td = torch.distributions
    want = td.kl_divergence(td.Normal(loc, scale), td.Normal(torch.zeros_like(loc), torch.ones_like(loc))).sum(1)
    torch.testing.assert_close(P.kl_normal_std(loc, scale), want)


def test_module_state_dict_has_reference_names():
    from spvipes_amd.module import spVIPESmodule

    g = Golden("label_train")
    m = spVIPESmodule({0: 24, 1: 32}, use_labels=True, n_hidden=g.H, n_dimensions_shared=g.n_s, n_dimensions_private=g.n_p)
    ref = g.state_dict()
    mine = m.state_dict()
    assert set(mine) == set(ref)
    for k in ref:
        assert tuple(mine[k].shape) == tuple(ref[k].shape), k
    m.load_state_dict(ref)  # strict


def test_module_refuses_cpu_minibatch():
    from spvipes_amd._abi import SpvError
    from spvipes_amd.module import spVIPESmodule

    m = spVIPESmodule({0: 8, 1: 8}, use_labels=True, n_hidden=8, n_dimensions_shared=4, n_dimensions_private=2)
    t = [{"X": torch.zeros(4, 16), "labels": torch.zeros(4, 1)}, {"X": torch.zeros(4, 16), "labels": torch.zeros(4, 1)}]
    with pytest.raises(SpvError):
        m(t)
