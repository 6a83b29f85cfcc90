"""dev probe: bench.py's sequence (serial-stream eager steps, then capture) -- which ingredient breaks the capture?"""
import os, subprocess, sys
code = r'''
import torch, sys
sys.path.insert(0, ".")
from spvipes_amd import _abi, ops
from spvipes_amd.data import make_synthetic_group, MinibatchSampler
from spvipes_amd.module import spVIPESmodule
from spvipes_amd.train import Trainer
dev = torch.device("cuda:0")
mode = sys.argv[1]
G, B, H = 10000, 4096, 128
groups = [make_synthetic_group(g, 8192, G, dev) for g in range(2)]
torch.manual_seed(0)
m = spVIPESmodule({0: G, 1: G}, use_labels=True, n_hidden=H, n_dimensions_shared=25, n_dimensions_private=10).to(dev)
t = Trainer(m, [g.counts for g in groups], labels=[g.labels for g in groups])
s = MinibatchSampler([8192, 8192], B, dev, seed=0)
m.train()
rows = next(iter(s.epoch()))
if "pre" in mode:
    t.step(rows, kl_weight=1.0)
if "serial" in mode:
    ops.SERIAL_STREAMS = True
    for _ in range(2): t.step(rows, kl_weight=1.0)
    torch.cuda.synchronize()
    if "prof" in mode:
        _abi.profile_start(["spv_dec_nb_fwd", "spv_enc_fc1_fwd"])
        for _ in range(3): t.step(rows, kl_weight=1.0)
        _abi.profile_stop()
    ops.SERIAL_STREAMS = False
t.capture(rows)
t.step(rows, kl_weight=1.0)
torch.cuda.synchronize()
print("capture ok")
'''
for mode in ("plain", "pre", "serial", "pre+serial", "serial+prof"):
    r = subprocess.run([sys.executable, "-c", code, mode], env=dict(os.environ), capture_output=True, text=True)
    err = [l for l in r.stderr.strip().splitlines() if "Error" in l or "error" in l]
    print(mode, "->", (r.stdout.strip().splitlines() or err[-1:] or ["?"])[-1][:160], flush=True)
