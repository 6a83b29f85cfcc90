"""Duck-typed AnnData for the user-API tests: the fields of prepare_adatas' output that the model reads
(data/prepare_adatas.py:97-132).  Test infrastructure."""
import numpy as np


class DuckAnnData:
    """The fields of prepare_adatas' output that the model reads (data/prepare_adatas.py:97-132)."""

    def __init__(self, X, obs, uns):
        self.X, self.obs, self.uns, self.layers = X, obs, uns, {}
        self.n_obs, self.n_vars = X.shape


def make_duck(n=(300, 260), G=(96, 80), seed=0, with_plan=False):
    rng = np.random.default_rng(seed)
    X = np.zeros((n[0] + n[1], G[0] + G[1]), np.float32)
    X[: n[0], : G[0]] = rng.poisson(2.0, (n[0], G[0])) * (rng.random((n[0], G[0])) < 0.3)
    X[n[0]:, G[0]:] = rng.poisson(2.0, (n[1], G[1])) * (rng.random((n[1], G[1])) < 0.3)
    X[: n[0], 0] += 1
    X[n[0]:, G[0]] += 1
    obs = {"groups": np.array(["a"] * n[0] + ["b"] * n[1]), "indices": np.concatenate([np.arange(n[0]), np.arange(n[1])]).astype(np.int32),
           "cell_type": np.concatenate([rng.integers(0, 4, n[0]), rng.integers(1, 5, n[1])])}
    uns = {"groups_lengths": {0: G[0], 1: G[1]}, "groups_var_indices": [np.arange(G[0]), G[0] + np.arange(G[1])],
           "groups_obs_indices": [np.arange(n[0]), n[0] + np.arange(n[1])], "groups_obs_names": None, "groups_var_names": None}
    if with_plan:
        uns["plan"] = (rng.random(n) * (rng.random(n) < 0.1)).astype(np.float32)
    return DuckAnnData(X, obs, uns)
