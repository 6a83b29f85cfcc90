"""Dev probe: rehearse the data-parallel step over RCCL on ONE GPU.  A one-rank "nccl" process group is a valid communicator
(all-reduce = copy), so pretending world = 2 drives exactly the code a multi-GPU rank runs: communicator start-up, the process
group's watchdog thread polling collectives while the step's hipGraphs are captured, async bucket all-reduces on RCCL's stream
between the two graph replays.  Numbers mean nothing (gradients get divided by 2); it must simply run, stay finite and match
the same steps without the process group in everything but that factor.
usage: rccl_one_rank.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 50
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

from spvipes_amd.data import MinibatchSampler, make_synthetic_group
from spvipes_amd.module import spVIPESmodule
from spvipes_amd.train import Trainer

torch.manual_seed(0)
cells, genes, B = 20000, 10000, 4096
groups = [make_synthetic_group(g, cells, genes, dev) for g in range(2)]
module = spVIPESmodule({0: genes, 1: genes}, use_labels=True, n_hidden=128, n_dimensions_shared=25, n_dimensions_private=10,
                       precision="bf16").to(dev)
tr = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups])
tr.world, tr.overlap = 2, True          # what a rank of a 2-GPU job sees
dist.broadcast(tr.fp.flat, src=0)
sampler = MinibatchSampler([cells, cells], B, dev, seed=0)
module.train()
def batches():
    while True:
        for rows in sampler.epoch():
            yield rows
it = batches()
for _ in range(3):
    tr.step(next(it), kl_weight=1.0)    # eager: collectives in flight right up to the capture
tr.capture(next(it))
assert tr.graph2 is not None
for _ in range(5):
    tr.step(next(it), kl_weight=1.0)
torch.cuda.synchronize(); dist.barrier()
t0 = time.perf_counter()
for _ in range(steps):
    last = tr.step(next(it), kl_weight=1.0)
torch.cuda.synchronize(); dist.barrier()
dt = (time.perf_counter() - t0) / steps
loss = float(last.loss.detach())
assert loss == loss and abs(loss) < 1e9, loss
print(f"one-rank RCCL rehearsal: {steps} graph steps with two async bucket all-reduces each, {dt * 1e3:.3f} ms/step, loss {loss:.1f}", flush=True)
dist.destroy_process_group()
