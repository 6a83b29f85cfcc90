import ctypes as C, sys, torch
sys.path.insert(0, "/root/repo")
from spvipes_amd import _abi
from spvipes_amd.nn_ops import _lin_batch, _add_lin
dev = torch.device("cuda:0")
B, K, N = 4096, 16, 128
X = torch.ones(B, K, device=dev); W = torch.ones(N, K, device=dev)
outs = []
for seed in (1, 2):
    Ys = [torch.empty(B, N, device=dev) for _ in range(2)]
    b = _lin_batch(B, relu=True, drop_p=0.1, seed=seed)
    for i in range(2):
        _add_lin(b, N=N, K=K, W=_abi.ptr(W), X=_abi.ptr(X), ldx=K, Y=_abi.ptr(Ys[i]), ldy=N)
    _abi.call("spv_linear_fwd", C.byref(b), _abi.stream_ptr())
    torch.cuda.synchronize()
    outs += [(y > 0).float() for y in Ys]
for i, m in enumerate(outs):
    print("keep fraction", i, float(m.mean()), "row-min/max", float(m.mean(1).min()), float(m.mean(1).max()), "col-min/max", float(m.mean(0).min()), float(m.mean(0).max()))
import itertools
for (i, a), (j, c) in itertools.combinations(enumerate(outs), 2):
    print("corr", i, j, float(((a - a.mean()) * (c - c.mean())).mean() / (a.std() * c.std())))
a = outs[0]
print("neighbour corr cols", float(((a[:, :-1] - a.mean()) * (a[:, 1:] - a.mean())).mean() / a.var()), "rows", float(((a[:-1] - a.mean()) * (a[1:] - a.mean())).mean() / a.var()))
