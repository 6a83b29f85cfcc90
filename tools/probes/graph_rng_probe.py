"""Probe (dev tool, GPU): does torch.cuda.CUDAGraph.replay() launch its two generator-state fill kernels when the captured work
contains no torch RNG op?  Run under rocprofv3 --kernel-trace --stats and count FillFunctor<long> launches."""
import sys, torch
use_rng = len(sys.argv) > 1 and sys.argv[1] == "rng"
x = torch.ones(1 << 16, device="cuda")
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    y = x * 2
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    y = x * 2
    if use_rng:
        z = torch.randn(1024, device="cuda")
torch.cuda.synchronize()
for _ in range(100):
    g.replay()
torch.cuda.synchronize()
print("done", use_rng)
