"""Helpers shared by the parity tests: load the committed golden fixtures
(tests/golden/*.npz, produced by tests/golden/make_goldens.py from the reference itself)."""
import glob
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL_CASES = sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                   if not os.path.basename(p).startswith("host_"))   # host_*.npz: host-side semantics (tests/golden/make_host_goldens.py)
LOSS_CASES = [c for c in ALL_CASES if "infer" not in c]
INFER_CASES = [c for c in ALL_CASES if "infer" in c]


class Golden:
    def __init__(self, name):
        self.name = name
        z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
        self.raw = {k: z[k] for k in z.files}
        self.mode = str(self.raw["in/mode"])
        self.training = bool(self.raw["in/training"])
        self.dropout = float(self.raw["in/dropout"])
        self.kl_weight = float(self.raw["in/kl_weight"])
        self.H, self.n_s, self.n_p = (int(v) for v in self.raw["in/dims"])
        self.has_loss = "out/loss" in self.raw
        self.n_batch = int(self.raw["in/n_batch"]) if "in/n_batch" in self.raw else 0   # batch covariates (n_batch > 1 cases)

    def batch_kwargs(self):
        """forward_loss keyword arguments of the covariate cases"""
        if self.n_batch <= 1:
            return {}
        return {"n_batch": self.n_batch, "batch_index": [self.t("in/batch0", torch.int64), self.t("in/batch1", torch.int64)]}

    def one_hot(self, grp, dtype=torch.float32):
        if self.n_batch <= 1:
            return None
        return torch.nn.functional.one_hot(self.t(f"in/batch{grp}", torch.int64), self.n_batch).to(dtype)

    def t(self, key, dtype=torch.float32):
        return torch.tensor(self.raw[key]).to(dtype)

    def state_dict(self, dtype=torch.float32):
        sd = {}
        for k, v in self.raw.items():
            if k.startswith("sd/"):
                tv = torch.tensor(v)
                sd[k[3:]] = tv.to(dtype) if tv.is_floating_point() else tv
        return sd

    def counts(self, dtype=torch.float32):
        return [self.t("in/counts0", dtype), self.t("in/counts1", dtype)]

    def noise(self, dtype=torch.float32):
        return {k[6:]: self.t(k, dtype) for k in self.raw if k.startswith("noise/") and "n_discarded" not in k and "drop_" not in k}

    def dropout_masks(self, dtype=torch.float32):
        m = {k[len("noise/drop_"):]: self.t(k, dtype) for k in self.raw if k.startswith("noise/drop_")}
        return m or None

    def poe_inputs(self, dtype=torch.float32):
        kw = {}
        if self.mode == "label":
            kw["labels"] = [self.t("in/labels0", dtype), self.t("in/labels1", dtype)]
        else:
            plan = self.t("in/plan", dtype)
            i0, i1 = self.raw["in/idx0"], self.raw["in/idx1"]
            kw["plan_block"] = plan[i0][:, i1]
            if self.mode == "cluster":
                kw["components"] = [self.t("in/comp0", dtype), self.t("in/comp1", dtype)]
        return kw

    def grads(self):
        return {k[5:]: torch.tensor(v) for k, v in self.raw.items() if k.startswith("grad/")}


def host_golden(name):
    """tests/golden/host_<name>.npz: arrays recorded from the reference's own loaders / splitter / model methods
    (tests/golden/make_host_goldens.py)."""
    z = np.load(os.path.join(GOLDEN_DIR, f"host_{name}.npz"), allow_pickle=False)
    return {k: z[k] for k in z.files}


def unragged(raw, key):
    v, o = raw[key + "/values"], raw[key + "/offsets"]
    return [v[o[i]:o[i + 1]] for i in range(len(o) - 1)]
