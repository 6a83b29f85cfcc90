#!/usr/bin/env python3
"""What a DEPENDENT launch costs inside a replayed hipGraph on this stack (torch.cuda.CUDAGraph -> hipGraph), and what the small
kernels of the step cost beyond that (dev probe; GPU box).  Chains of N dependent launches are captured once and replayed; the figure is
(replay time of the N-chain - replay time of the 1-chain) / (N - 1), un-profiled, HIP events around 200 replays.

    python tools/probes/graph_chain_probe.py [N]
"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from spvipes_amd import _abi  # noqa: E402
from spvipes_amd.nn_ops import _add_lin, _lin_batch  # noqa: E402

dev = torch.device("cuda:0")
_abi.load()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 32
REP = 200


def replay_us(build, n):
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        build(1)
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    with torch.cuda.graph(g):
        build(n)
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(REP):
        g.replay()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3 / REP


def per_launch(name, build):
    t1, tn = replay_us(build, 1), replay_us(build, N)
    print(f"{name:72s} 1 launch {t1:7.2f} us   {N} launches {tn:8.2f} us   per dependent launch {(tn - t1) / (N - 1):6.2f} us", flush=True)


cnt = torch.zeros((), dtype=torch.int64, device=dev)


def bump(n):
    for _ in range(n):
        _abi.call("spv_counter_bump", _abi.ptr(cnt), _abi.stream_ptr())


per_launch("counter_bump (one thread, one atomic-free RMW)", bump)

src = torch.arange(4096, dtype=torch.int32, device=dev)
dsts = [torch.empty_like(src) for _ in range(2)]


def gather(n):
    for i in range(n):
        _abi.gather_u32([(src if i == 0 else dsts[(i + 1) % 2], None, dsts[i % 2])])


per_launch("gather_u32 4096 words (copy of the previous launch's output)", gather)


def two_streams(n):
    side = torch.cuda.Stream()
    for i in range(n):
        if i % 2:
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                _abi.call("spv_counter_bump", _abi.ptr(cnt), _abi.stream_ptr())
            torch.cuda.current_stream().wait_stream(side)
        else:
            _abi.call("spv_counter_bump", _abi.ptr(cnt), _abi.stream_ptr())


per_launch("counter_bump, every second launch on a side stream (fork + join)", two_streams)


def linear_chain(B, H, nprob):
    xs = [[torch.randn(B, H, device=dev) for _ in range(nprob)] for _ in range(2)]
    W = [torch.randn(H, H, device=dev) * 0.05 for _ in range(nprob)]
    bias = [torch.zeros(H, device=dev) for _ in range(nprob)]

    def build(n):
        for i in range(n):
            b = _lin_batch(B, relu=True)
            for p in range(nprob):
                _add_lin(b, N=H, K=H, W=_abi.ptr(W[p]), bias=_abi.ptr(bias[p]), X=_abi.ptr(xs[i % 2][p]), ldx=H, Y=_abi.ptr(xs[(i + 1) % 2][p]), ldy=H)
            _abi.call("spv_linear_fwd", C.byref(b), _abi.stream_ptr())
    return build


for B, H, nprob in ((128, 64, 4), (4096, 128, 1), (4096, 128, 4), (4096, 32, 4)):
    per_launch(f"spv_linear_fwd chain: {nprob} x [{B} x {H}] x [{H} x {H}]", linear_chain(B, H, nprob))
