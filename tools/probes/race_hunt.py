"""Dev probe: repeat one training step from identical state while another process keeps the GPU busy, and report any
parameter whose gradient is not bit-identical across repetitions (a stream race shows up as an intermittent difference).
usage: race_hunt.py [overlap 0|1] [precision] [reps]"""
import os, sys, subprocess, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

if len(sys.argv) > 1 and sys.argv[1] == "--noise":
    x = torch.randn(2048, 2048, device="cuda")
    t0 = time.time()
    while time.time() - t0 < float(sys.argv[2]):
        for _ in range(20):
            x = (x @ x).tanh()
        torch.cuda.synchronize()
    sys.exit(0)

from spvipes_amd.data import make_synthetic_group
from spvipes_amd.module import spVIPESmodule
from spvipes_amd.train import Trainer

overlap = bool(int(sys.argv[1])) if len(sys.argv) > 1 else True
precision = sys.argv[2] if len(sys.argv) > 2 else "fp32"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 150
dev = torch.device("cuda:0")
G = (600, 500)
groups = [make_synthetic_group(g, 1024, G[g], dev) for g in range(2)]
if os.environ.get("RACE_CLAMP"):   # keep every count inside the likelihood kernel's (count, gene) table: its fix-up path never runs
    for g_ in groups:
        g_.counts.X.clamp_(max=int(os.environ["RACE_CLAMP"]))
print("max count", [int(g_.counts.X.max()) for g_ in groups], "entries >= 64:", [int((g_.counts.X >= 64).sum()) for g_ in groups], flush=True)
torch.manual_seed(0)
module = spVIPESmodule({0: G[0], 1: G[1]}, use_labels=True, n_hidden=64, n_dimensions_shared=10, n_dimensions_private=5, dropout_rate=0.1,
                       precision=precision).to(dev)
tr = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups], lr=5e-3, overlap_allreduce=overlap)
module.train()
gen = torch.Generator().manual_seed(100)
rows = [torch.randperm(1024, generator=gen)[:256].to(torch.int32).to(dev) for _ in range(2)]
noise = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--noise", os.environ.get("RACE_NOISE_S", "60")]) for _ in range(2)]
time.sleep(8)   # (the noise makers have to import torch first)
first, nbad = None, 0
for it in range(reps):
    torch.manual_seed(1000)
    if getattr(module, "_rng_counter", None) is not None:   # the trainer's device generator: noise + dropout keyed by this counter
        module._rng_counter.fill_(50)
    else:
        if getattr(module, "_seed_dev", None) is None:
            module._seed_dev = torch.zeros((), dtype=torch.int64, device=dev)
        module._seed_dev.fill_(50)
    if os.environ.get("RACE_FWD_ONLY"):
        tr.fp.grad.zero_()
        _, _, lo = module(tr.minibatch(rows), loss_kwargs={"kl_weight": 1.0})
    else:
        lo = tr._forward_backward(rows, 1.0)
        if overlap:
            tr._backward_encoders()
    torch.cuda.synchronize()
    g = {n: p.grad.clone() for n, p in module.named_parameters()}
    for gi in (0, 1):   # every workspace buffer of the decoder / encoders as it stands after the step
        for key, t in module._workspace(gi, dev)._buf.items():
            g[f"<ws{gi}:{key[0]}{list(key[1])}>"] = t.clone()
    g["<loss>"] = lo.loss.detach().clone()
    for k_, v_ in lo.kl_local.items():
        g["<" + k_ + ">"] = v_.detach().clone()
    for k_, v_ in lo.reconstruction_loss.items():
        g["<" + k_ + ">"] = v_.detach().clone()
    if first is None:
        first = g
        continue
    def where(n):
        d = (g[n] != first[n]).flatten().nonzero().flatten()
        return f"{n} ({float((g[n].float() - first[n].float()).abs().max()):.2e}; {d.numel()} of {g[n].numel()} entries, flat idx {int(d.min())}..{int(d.max())})"
    bad = [where(n) for n in g if not torch.equal(g[n], first[n]) and (n.startswith("decoder") or n.startswith("<"))]
    if bad:
        nbad += 1
        if nbad <= 5:
            print(f"rep {it}: {len(bad)} parameters differ: " + ", ".join(bad), flush=True)
for n_ in noise:
    n_.terminate()
print(f"overlap={overlap} {precision}: {nbad} of {reps - 1} repetitions differed", flush=True)
