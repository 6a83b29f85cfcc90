"""Probe (dev tool, GPU): time of spv_linear_fwd as a function of K, N, number of problems, relu / dropout (HIP events, 200 launches)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from spvipes_amd import _abi
from spvipes_amd.nn_ops import _lin_batch, _add_lin
dev = torch.device("cuda:0")
B = 4096
def run(K, N, nprob, relu=False, drop=0.0, ldx=None):
    ldx = ldx or K
    X = torch.randn(B, ldx, device=dev); W = [torch.randn(N, K, device=dev) for _ in range(nprob)]
    Y = [torch.empty(B, N, device=dev) for _ in range(nprob)]; bias = torch.zeros(N, device=dev)
    b = _lin_batch(B, relu=relu, drop_p=drop, seed=1)
    for i in range(nprob):
        _add_lin(b, N=N, K=K, W=_abi.ptr(W[i]), X=_abi.ptr(X), ldx=ldx, bias=_abi.ptr(bias), Y=_abi.ptr(Y[i]), ldy=N)
    for _ in range(5): _abi.call("spv_linear_fwd", C.byref(b), _abi.stream_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(200): _abi.call("spv_linear_fwd", C.byref(b), _abi.stream_ptr())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 200 * 1e3
for (K, N, nprob, relu, drop) in [(16, 32, 1, False, 0), (128, 32, 1, False, 0), (128, 128, 1, False, 0), (128, 128, 4, False, 0), (128, 128, 4, True, 0.1),
                                  (256, 256, 2, False, 0), (32, 256, 2, True, 0), (35, 256, 2, True, 0), (128, 32, 8, False, 0), (128, 25, 8, False, 0)]:
    print(f"K={K:4d} N={N:4d} nprob={nprob} relu={relu} drop={drop}: {run(K, N, nprob, relu, drop):6.1f} us per launch (back-to-back, incl. launch gaps)")
