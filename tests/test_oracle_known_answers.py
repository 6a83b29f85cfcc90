"""Known-answer tests for the third-party (scvi-tools 0.20.0) arithmetic the reference calls but
does not contain, and for closed forms the reference itself states.  These pin the part of the
parity chain that the reference's own tests leave unpinned."""
import numpy as np
import pytest
import torch
from scipy import stats
from scipy.special import gammaln

from oracle import scvi_standins as S
from oracle import spvipes_oracle as O


def test_nb_mixture_matches_scipy_nbinom_on_integer_counts():
    rng = np.random.default_rng(0)
    B, G = 7, 11
    x = rng.poisson(3.0, size=(B, G)).astype(np.float64)
    mu1 = rng.uniform(0.05, 8.0, size=(B, G))
    mu2 = rng.uniform(0.05, 8.0, size=(B, G))
    theta = rng.uniform(0.2, 6.0, size=G)
    logits = rng.normal(size=(B, G))
    pi = 1.0 / (1.0 + np.exp(-logits))  # sigmoid(logits) weights component 1
    p1 = stats.nbinom.pmf(x, theta, theta / (theta + mu1))
    p2 = stats.nbinom.pmf(x, theta, theta / (theta + mu2))
    want = np.log(pi * p1 + (1 - pi) * p2)
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    got = O.log_mixture_nb(t(x), t(mu1), t(mu2), t(theta), t(logits)).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)  # eps=1e-8 terms only
    got_standin = S.NegativeBinomialMixture(t(mu1), t(mu2), t(theta), t(logits)).log_prob(t(x)).numpy()
    np.testing.assert_allclose(got_standin, got, rtol=0, atol=1e-12)


def test_nb_mixture_at_non_integer_x_matches_gamma_form():
    """The reference evaluates the NB at x = log1p(count) (spVIPESmodule.py:818-824): the formula
    must be the gamma-function continuation, not a pmf."""
    rng = np.random.default_rng(1)
    x = np.log1p(rng.poisson(2.0, size=(5, 9)).astype(np.float64))
    mu = rng.uniform(0.05, 4.0, size=(5, 9))
    theta = rng.uniform(0.3, 3.0, size=9)
    want = (gammaln(x + theta) - gammaln(theta) - gammaln(x + 1)
            + theta * np.log(theta / (theta + mu)) + x * np.log(mu / (theta + mu)))
    t = lambda a: torch.tensor(a, dtype=torch.float64)
    # both components equal => mixture == single NB whatever the logits
    got = O.log_mixture_nb(t(x), t(mu), t(mu), t(theta), t(rng.normal(size=(5, 9)))).numpy()
    np.testing.assert_allclose(got, want, rtol=1e-6, atol=1e-6)


def test_kl_matches_reference_closed_form_and_torch():
    """module/utils.py:4-15 get_kl: -0.5 * sum(1 + 2 log s - mu^2 - s^2)."""
    g = torch.Generator().manual_seed(0)
    mu, logsig = torch.randn(6, 5, generator=g, dtype=torch.float64), 0.3 * torch.randn(6, 5, generator=g, dtype=torch.float64)
    want = -0.5 * (1 + 2 * logsig - mu.pow(2) - (2 * logsig).exp()).sum(1)
    got = O.kl_normal_std(mu, logsig.exp())
    torch.testing.assert_close(got, want, rtol=1e-12, atol=1e-12)
    td = torch.distributions
    ref = td.kl_divergence(td.Normal(mu, logsig.exp()), td.Normal(torch.zeros_like(mu), torch.ones_like(mu))).sum(1)
    torch.testing.assert_close(got, ref, rtol=1e-12, atol=1e-12)


def test_fclayers_single_layer_layout():
    """fc_layers[0][0] is the Linear and fc_layers[0][1] the BatchNorm1d(eps=1e-3, momentum=0.01):
    the layout the reference indexes at spVIPESmodule.py:782-789."""
    fc = S.FCLayers(n_in=3, n_out=7, n_cat_list=[1], n_layers=1, use_activation=False, use_batch_norm=True,
                    use_layer_norm=False, bias=False, dropout_rate=0)
    lin, bn = fc.fc_layers[0][0], fc.fc_layers[0][1]
    assert isinstance(lin, torch.nn.Linear) and lin.bias is None and lin.weight.shape == (7, 3)
    assert isinstance(bn, torch.nn.BatchNorm1d) and bn.eps == 1e-3 and bn.momentum == 0.01
    assert len(fc.fc_layers[0]) == 2
    names = [k for k, _ in fc.state_dict().items()]
    assert "fc_layers.Layer 0.0.weight" in names and "fc_layers.Layer 0.1.running_var" in names
    trunk = S.FCLayers(n_in=9, n_out=256, n_cat_list=[1], n_layers=1, n_hidden=256, dropout_rate=0,
                       use_batch_norm=True, use_layer_norm=False)
    assert isinstance(trunk.fc_layers[0][2], torch.nn.ReLU) and trunk.fc_layers[0][0].bias is not None


def test_label_partner_rules():
    """A4a pairing rule: k-th same-label cell of the other minibatch; padding modes 1 and 2."""
    l0 = torch.tensor([2., 0., 2., 5., 2., 0.])
    l1 = torch.tensor([0., 2., 7., 2.])
    partner, mode = O.label_partner(l0, l1)
    assert partner.tolist() == [1, 0, 3, -1, -1, -1]
    assert mode.tolist() == [0, 0, 0, 2, 1, 1]
    partner, mode = O.label_partner(l1, l0)
    assert partner.tolist() == [1, 0, -1, 2] and mode.tolist() == [0, 0, 2, 0]


def test_poe_fuse_is_precision_weighted_with_unit_prior():
    loc, logvar = torch.tensor([[0.5]], dtype=torch.float64), torch.tensor([[0.2]], dtype=torch.float64)
    o_loc, o_logvar = torch.tensor([[-1.0]], dtype=torch.float64), torch.tensor([[-0.4]], dtype=torch.float64)
    v, ov = logvar.exp(), o_logvar.exp()
    j_loc, j_logvar = O._fuse(loc, logvar, 1 / ov, o_loc / ov)
    prec = 1 + 1 / v + 1 / ov
    torch.testing.assert_close(j_loc, (loc / v + o_loc / ov) / prec)
    torch.testing.assert_close(j_logvar, -prec.log())


@pytest.mark.parametrize("n_s,n_p", [(6, 3), (10, 5), (3, 5), (4, 4)])
def test_latent_slicing_quirk(n_s, n_p):
    """A6: decoder's z_shared = [private | poe[:n_s-n_p]] and z_private = poe[n_s-n_p:] when n_p <= n_s."""
    priv = torch.arange(2 * n_p, dtype=torch.float32).reshape(2, n_p)
    poe = 100 + torch.arange(2 * n_s, dtype=torch.float32).reshape(2, n_s)
    zp, zs = O.split_latents(priv, poe, n_s, n_p)
    assert zp.shape == (2, n_p) and zs.shape == (2, n_s)
    if n_p <= n_s:
        assert torch.equal(zs, torch.cat([priv, poe[:, : n_s - n_p]], 1))
        assert torch.equal(zp, poe[:, n_s - n_p:])
