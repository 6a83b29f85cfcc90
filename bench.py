#!/usr/bin/env python3
"""Headline benchmark: training-step throughput (cells/s) of the 2-group PoE VAE on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment (WORLD_SIZE unset) the first form starts the second one itself as a child process,
one rank per GPU, forwards rank 0's JSON line and exits with the child's status; a WORLD_SIZE that differs from --gpus is an
error.  --gpus N > 1 without --config runs c3 (one rank's shard of the 2 x 200 000 x 20 000 configuration the scaling target
is quoted on).

One "step" = one optimisation step of the hot path on one synthetic minibatch per group
(forward through both encoders, label-based PoE, decoder + NB-mixture ELBO, backward, one gradient
all-reduce when N > 1, Adam).  Default workload (--config c2) = BASELINE.json configs[1]: 2 groups x
50 000 cells x 10 000 genes, n_shared 25, n_private 10, n_hidden 128, bf16 MFMA operands with fp32
accumulation; the configuration does not fix the minibatch, we use 4096 cells per group per step (the
batch size BASELINE.json names for its 8-GPU configuration).  Weak scaling: every rank holds its own
shard per group and draws its own minibatches; value = cells processed by ALL ranks / s.
The count matrices are resident in HBM (uint16) before the timed region.

Other workloads, one per BASELINE.json config, selected with --config:
    c1  2 x 2 000 x 2 000, B 128, H 64, 10/5 (the reference's CPU plumbing case)
    c3  one rank's shard of 2 x 200 000 x 20 000 on 8 GPUs: 25 000 cells x 20 000 genes per group, B 4096
    c4  3 groups x 100 000 x 15 000 on 4 GPUs: 25 000 cells per group per rank, H 256, cluster-matched PoE
        (throughput only: the reference cannot run 3 groups, data/prepare_adatas.py:94-95)
    c5  one rank's shard of 2 x 500 000 x 30 000 on 8 GPUs: 62 500 cells x 30 000 genes, paired PoE on a sparse plan, fp32
Explicit --cells / --genes / ... flags override the preset.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
MFMA_BF16_PEAK_TFLOPS = 2500.0  # dense bf16


CONFIGS = {
    "c1": dict(cells=2_000, genes=2_000, batch_size=128, n_hidden=64, n_shared=10, n_private=5, groups=2, poe="label", precision="bf16",
               what="BASELINE configs[0]"),
    "c2": dict(cells=50_000, genes=10_000, batch_size=4096, n_hidden=128, n_shared=25, n_private=10, groups=2, poe="label", precision="bf16",
               what="BASELINE configs[1]"),
    "c3": dict(cells=25_000, genes=20_000, batch_size=4096, n_hidden=128, n_shared=25, n_private=10, groups=2, poe="label", precision="bf16",
               what="BASELINE configs[2]: one rank's shard of 2 x 200 000 x 20 000 over 8 GPUs"),
    "c4": dict(cells=25_000, genes=15_000, batch_size=4096, n_hidden=256, n_shared=25, n_private=10, groups=3, poe="cluster", precision="bf16",
               what="BASELINE configs[3]: one rank's shard of 3 x 100 000 x 15 000 over 4 GPUs (throughput only)"),
    "c5": dict(cells=62_500, genes=30_000, batch_size=4096, n_hidden=128, n_shared=25, n_private=10, groups=2, poe="paired", precision="fp32",
               what="BASELINE configs[4]: one rank's shard of 2 x 500 000 x 30 000 over 8 GPUs"),
}


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)     # >= 50 timed steps after 10 warm-up
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--config", default=None, choices=sorted(CONFIGS),
                    help="BASELINE.json workload preset (see the module docstring); default: c2 on one GPU, c3 (the shard shape the 1 -> 8 "
                         "scaling target is quoted on) when --gpus > 1")
    ap.add_argument("--cells", type=int, default=None, help="cells per group per rank")
    ap.add_argument("--genes", type=int, default=None, help="genes per group")
    ap.add_argument("--batch-size", type=int, default=None, help="cells per group per step per rank")
    ap.add_argument("--n-hidden", type=int, default=None)
    ap.add_argument("--n-shared", type=int, default=None)
    ap.add_argument("--n-private", type=int, default=None)
    ap.add_argument("--groups", type=int, default=None, choices=[2, 3])
    ap.add_argument("--poe", default=None, choices=["label", "paired", "cluster"])
    ap.add_argument("--precision", default=None, choices=["bf16", "fp32"])
    ap.add_argument("--no-elbo-delta", action="store_true", help="skip the one-step ELBO comparison against the CPU oracle")
    ap.add_argument("--count-dtype", default="u16", choices=["u16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--allreduce", default="auto", choices=["auto", "single", "overlap"],
                    help="gradient all-reduce of a data-parallel job: one collective after the backward pass, or two buckets with the "
                         "decoder bucket overlapped with the encoder half of the backward pass (auto: overlap when WORLD_SIZE > 1)")
    ap.add_argument("--cpu-batch", type=int, default=None, help="cells per group of the CPU baseline sample (default: the timed batch size)")
    ap.add_argument("--cpu-steps", type=int, default=1, help="timed CPU steps after one untimed one (one step of 2 x 4096 x 10 000 is ~5-10 s of host time)")
    args = ap.parse_args(argv)
    if args.config is None:
        args.config = "c2" if args.gpus <= 1 else "c3"
    preset = CONFIGS[args.config]
    for k, v in preset.items():
        if k != "what" and getattr(args, k) is None:
            setattr(args, k, v)
    return args


def host_cpus():
    """(threads this process may really use, logical CPUs of the host, CPU model string).  The first is what the baseline runs on and
    reports as ``cores``: the scheduler affinity mask cut down by the cgroup CPU quota (a GPU box hands a 1-GPU job a share of a
    large host: os.cpu_count() threads on a 16-CPU quota would time the oversubscription, not the cores)."""
    logical = os.cpu_count() or 1
    usable = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else logical
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(f).read().split()
            if f.endswith("cpu.max"):
                if txt[0] != "max":
                    usable = min(usable, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    usable = min(usable, max(1, int(q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read()) + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    model = "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.lower().startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return usable, logical, model


def cpu_baseline(groups, args, plan=None):
    """The CPU oracle (oracle/spvipes_oracle.py, a port of the reference path pinned by goldens
    generated from the reference) timed on this box's host cores -- every core this process may use --
    forward + loss + backward + Adam on a bounded sample of the same workload: ``--cpu-steps`` steps at the TIMED batch size."""
    from oracle import spvipes_oracle as O

    n_threads, logical, cpu_model = host_cpus()
    torch.set_num_threads(n_threads)
    if args.cpu_batch is None:
        args.cpu_batch = args.batch_size
    Gs = [g.counts.G for g in groups]
    sd = O.init_state_dict(Gs, n_hidden=args.n_hidden, n_dimensions_shared=args.n_shared, n_dimensions_private=args.n_private, seed=0)
    names = O.param_names(sd)
    for k in names:
        sd[k].requires_grad_(True)
    opt = torch.optim.Adam([sd[k] for k in names], lr=1e-3, eps=0.01, weight_decay=1e-6)
    B = args.cpu_batch
    gen = torch.Generator().manual_seed(0)
    times = []
    for step in range(args.cpu_steps + 1):
        counts, labels, picked = [], [], []
        for g in groups:
            rows = torch.randint(0, g.counts.n_cells, (B,), generator=gen)
            X = g.counts.X[rows.to(g.counts.X.device)].cpu()
            X = torch.from_numpy(X.numpy().view(np.uint16).astype(np.float32)) if X.dtype == torch.int16 else X
            counts.append(X)
            labels.append(g.labels[rows.to(g.labels.device)].cpu())
            picked.append(rows.numpy())
        kw = {"labels": labels} if args.poe == "label" else {"plan_block": torch.from_numpy(plan[picked[0]][:, picked[1]].toarray().astype(np.float32))}
        if args.poe == "cluster":
            kw["components"] = labels
        noise = {f"enc_{i}_{k}": torch.randn(B, n, generator=gen) for i in range(2) for k, n in (("private", args.n_private), ("shared", args.n_shared))}
        noise.update({f"poe_{i}": torch.randn(B, args.n_shared, generator=gen) for i in range(2)})
        t0 = time.perf_counter()
        out = O.forward_loss(sd, counts, n_dimensions_shared=args.n_shared, n_dimensions_private=args.n_private, noise=noise,
                             mode=args.poe, training=True, **kw)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        if step > 0:  # first step pays allocator warm-up
            times.append(time.perf_counter() - t0)
    sec = float(np.mean(times))
    return {"value": 2 * B / sec, "unit": "cells/s", "cores": n_threads, "kind": "port", "cpu_model": cpu_model, "host_logical_cpus": logical,
            "torch_threads": torch.get_num_threads(),
            "sample": f"{args.cpu_steps} step(s) after one untimed step, 2 x {B} cells x {Gs[0]} genes = the timed minibatch shape "
                      f"(same synthetic data, {args.poe} PoE, fwd+loss+bwd+Adam), {sec:.2f} s/step"}



def _workload_tag(args) -> str:
    return f"{args.config} {args.precision} {args.count_dtype} B{args.batch_size} G{args.genes}"


def _pmc_table(args):
    """The committed per-kernel counter table of THIS run's workload (profiles/<LATEST>_<preset>_pmc_table.json, written by
    tools/pmc_workload.sh from five separate rocprofv3 --pmc passes of this command; bench.py itself cannot run under the profiler),
    or (None, None): a table is only used when its `workload` tag names this run's preset / precision / count dtype / batch / genes."""
    latest = os.path.join(ROOT, "profiles", "LATEST")   # tag of the newest evidence set (file names do not sort by time)
    if not os.path.exists(latest):
        return None, None
    name = f"{open(latest).read().strip()}_{args.config}_pmc_table.json"
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, None
    try:
        d = json.load(open(path))
    except (OSError, ValueError):
        return None, None
    return (d, "profiles/" + name) if d.get("workload") == _workload_tag(args) else (None, None)


def _pmc_fields(args):
    """HBM bytes per launch of the dominant kernel = FETCH_SIZE + WRITE_SIZE (KiB) and its VALU-busy fraction from the counter table of
    this run's workload; any other invocation gets nulls rather than another shape's counters.  The kernel's reads are 8 B / lane,
    outside the access widths the gfx950 FETCH_SIZE correction is calibrated for, so the raw counter is reported."""
    tab, src = _pmc_table(args)
    if tab is None:
        return None, None, None
    hit = [(e, 1.0) for k, e in tab["kernels"].items() if "dec_nb_kernel" in k and e.get("launches_per_step", 0) >= 1]
    # small steps: one grid for both groups (dec_nb_pair_kernel): a launch's bytes are two groups' -- the per-group share, like `achieved`
    hit += [(e, 0.5) for k, e in tab["kernels"].items() if "dec_nb_pair_kernel" in k and e.get("launches_per_step", 0) >= 1]
    if not hit or "FETCH_SIZE" not in hit[0][0] or "WRITE_SIZE" not in hit[0][0]:
        return None, None, None
    e, share = max(hit, key=lambda h: h[0]["launches_per_step"])
    return (e["FETCH_SIZE"] + e["WRITE_SIZE"]) * 1024.0 * share, src + " (raw FETCH_SIZE + WRITE_SIZE, KiB -> bytes)", e.get("valu_busy_frac")


# kernels of the decoder + likelihood chain (forward and backward) by name, and the FETCH_SIZE factor of each: x2 where the reads are
# 16 B / lane (LDS-DMA GEMMs, the one-pass backward: MI355X_MICROARCH.md, HBM section), raw elsewhere
_CHAIN = (("dec_nb_kernel", 1.0), ("dec_nb_pair_kernel", 1.0), ("dec_logits_dma_pair_kernel", 2.0), ("dec_lse_pair_kernel", 1.0), ("dec_lse_combine_pair_kernel", 1.0),
          ("dec_heads_bwd", 2.0), ("dec_softmax_bwd_kernel", 1.0), ("dec_gemm320_dma4", 2.0), ("dec_logits_dma_kernel", 2.0),
          ("dec_lse_kernel", 1.0), ("dec_lse_combine_kernel", 1.0), ("dec_heads_wgrad", 2.0), ("gemm_kernel<", 1.0))
_CHAIN_ENTRY_POINTS = ("spv_dec_nb_fwd", "spv_dec_logits", "spv_dec_lse", "spv_dec_softmax_bwd", "spv_dec_heads_bwd", "spv_gemm_bf16", "spv_dec_heads_wgrad",
                       "spv_dec_nb_fwd_grouped", "spv_dec_logits_grouped", "spv_dec_lse_grouped", "spv_dec_heads_bwd_grouped", "spv_gemm_bf16_grouped")


def _pmc_decoder_chain(args, n_groups=2, chain_ms_per_group_step=None):
    """[B, G] traffic of the whole decoder + likelihood chain of ONE group-step (likelihood, decoder backward, d A_m and d W_m GEMMs, logits GEMM,
    softmax statistics; in fp32 mode also the register-staged logits and regressor weight-gradient GEMMs) from the counter table of this
    run's workload, against SURVEY 8d's algorithmic bytes for that chain: 2 B G s_x (x read by the likelihood forward and backward) + 40
    bytes per decoder parameter.  With the chain's time (HIP events of this run, single stream) also the rate the counters imply against
    the HBM peak -- what BASELINE configs[4] asks for.  None unless a table of this workload is committed."""
    tab, src = _pmc_table(args)
    if tab is None:
        return None
    per, total = {}, 0.0
    for k, e in tab["kernels"].items():
        f = next((ff for pat, ff in _CHAIN if pat in k), None)
        if f is None or "FETCH_SIZE" not in e or "WRITE_SIZE" not in e:
            continue
        b = (e["FETCH_SIZE"] * f + e["WRITE_SIZE"]) * 1024.0 * e["launches_per_step"] / n_groups
        per[k[:60]] = b
        total += b
    if not per:
        return None
    B, G, n_s, n_p = args.batch_size, args.genes, args.n_shared, args.n_private
    sx = 2 if args.count_dtype == "u16" else 4
    p_dec = G * (n_p + 2) + G * (n_s + 2) + ((n_s + n_p) * 256 + 3 * 256) + G * (256 + n_s + n_p + 1) + G
    alg = 2.0 * B * G * sx + 40.0 * p_dec
    out = {"traffic_bytes_per_group_step": total, "algorithmic_bytes_per_group_step": alg, "ratio": total / alg, "per_kernel_bytes": per,
           "source": src + " (FETCH_SIZE x2 where reads are 16 B / lane, + WRITE_SIZE)"}
    if chain_ms_per_group_step:
        gbs = total / (chain_ms_per_group_step * 1e-3) / 1e9
        out.update({"chain_ms_per_group_step": chain_ms_per_group_step, "counter_GBs": gbs, "counter_frac_of_hbm_peak": gbs / HBM_PEAK_GBS,
                    "algorithmic_GBs": alg / (chain_ms_per_group_step * 1e-3) / 1e9})
    return out


def synthetic_plan(n0, n1, k=8, seed=2000):
    """the transport plan of the OT configs, defined implicitly as a permutation + k random neighbours per
    row (CSR, never densified)."""
    import scipy.sparse as sp
    rng = np.random.default_rng(seed)
    first = rng.permutation(n1)[:n0] if n1 >= n0 else rng.integers(0, n1, n0)
    cols = np.concatenate([first[:, None], rng.integers(0, n1, (n0, k))], axis=1).reshape(-1)
    rows = np.repeat(np.arange(n0), k + 1)
    vals = (rng.random(n0 * (k + 1)) + 0.05).astype(np.float32)
    m = sp.coo_matrix((vals, (rows, cols)), shape=(n0, n1)).tocsr()
    m.data = m.data.astype(np.float32)
    return m


def elbo_delta_gpu(trainer, module, groups, rows, args):
    """First half of the ELBO comparison (SURVEY.md 8d "ELBO delta"; module/spVIPESmodule.py:809-899): ONE eager training-mode
    step of this run's workload with injected noise, dropout off (the two implementations cannot share a dropout RNG stream),
    kl_weight 1, no optimiser step.  Returns what the CPU oracle needs to repeat it (run AFTER the timed region: its thread
    pool would otherwise still be spinning on the host cores while the timed steps are launched)."""
    B, n_p, n_s = args.batch_size, args.n_private, args.n_shared
    gen = torch.Generator().manual_seed(12345)
    noise = {f"enc_{g}_{k}": torch.randn(B, n, generator=gen) for g in range(2) for k, n in (("private", n_p), ("shared", n_s))}
    noise.update({f"poe_{g}": torch.randn(B, n_s, generator=gen) for g in range(2)})
    sd = {k: v.detach().cpu().clone() for k, v in module.state_dict().items()}
    p_drop = module.dropout_rate
    module.dropout_rate = 0.0
    try:
        lo = trainer.step(rows, kl_weight=1.0, noise={k: v.to(rows[0].device) for k, v in noise.items()}, optimizer_step=False)
        got = float(lo.loss.detach())
    finally:
        module.dropout_rate = p_drop
    counts = []
    for g in range(2):
        X = groups[g].counts.X[rows[g].long()].cpu()
        counts.append(torch.from_numpy(X.numpy().view(np.uint16).astype(np.float32)) if X.dtype == torch.int16 else X)
    return {"got": got, "sd": sd, "noise": noise, "counts": counts, "rows": [r.cpu().numpy() for r in rows],
            "labels": [groups[g].labels[rows[g].long()].cpu() for g in range(2)]}


def elbo_delta_oracle(st, args, plan):
    """Second half: the CPU oracle on the same parameters, minibatch and noise; |loss_build - loss_oracle| / |loss_oracle|."""
    from oracle import spvipes_oracle as O

    kw = {}
    if args.poe == "label":
        kw["labels"] = st["labels"]
    else:
        kw["plan_block"] = torch.from_numpy(plan[st["rows"][0]][:, st["rows"][1]].toarray().astype(np.float32))
        if args.poe == "cluster":
            kw["components"] = st["labels"]
    with torch.no_grad():
        want = float(O.forward_loss(st["sd"], st["counts"], n_dimensions_shared=args.n_shared, n_dimensions_private=args.n_private, noise=st["noise"],
                                    mode=args.poe, training=True, kl_weight=1.0, **kw)["loss"])
    got = st["got"]
    return {"value": abs(got - want) / abs(want), "loss_build": got, "loss_oracle": want,
            "what": f"one training-mode step of this workload (B {args.batch_size} x G {args.genes}, {args.poe} PoE, {args.precision}), same parameters / rows / noise, dropout off, kl_weight 1"}


def launcher_command(n: int, argv) -> list:
    """the one-rank-per-GPU launch of this script (the command the driver uses for N > 1), rendezvous on 127.0.0.1 at a free port"""
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__), *argv]


def visible_gpu_count():
    """GPUs this process would see, counted WITHOUT loading the HIP / HSA runtime (torch.cuda.device_count() may fall through to
    hipGetDeviceCount, which initialises it in the launcher parent): KFD topology nodes with SIMDs, cut down by
    HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.  None when sysfs says nothing (the ranks then fail loudly
    themselves and the launcher forwards their status)."""
    import glob
    n = 0
    nodes = glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties")
    if not nodes:
        return None
    for f in nodes:
        try:
            props = dict(l.split()[:2] for l in open(f).read().splitlines() if len(l.split()) >= 2)
        except OSError:
            return None
        n += int(props.get("simd_count", "0")) > 0
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launch_ranks(args, argv) -> int:
    """``python bench.py --gpus N`` (N > 1) outside a torchrun environment: start the N ranks as a CHILD process (never an exec:
    nothing in this process has touched the GPU, and nothing will -- the device count comes from sysfs), forward the child's output (rank 0 prints the JSON line) and
    return its exit status -- a plain `--gpus 8` can never end as a silent single-GPU number."""
    import subprocess
    cmd = launcher_command(args.gpus, argv)
    if os.environ.get("SPV_BENCH_LAUNCH_DRYRUN") == "1":   # (tests: show the command, start nothing)
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    n_dev = visible_gpu_count()
    if n_dev is not None and n_dev < args.gpus:
        print(f"bench.py: --gpus {args.gpus} but this node exposes {n_dev} GPU(s)", file=sys.stderr, flush=True)
        return 2
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    args = parse(argv)
    if "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:   # before anything touches the GPU
            raise SystemExit(launch_ranks(args, argv))
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (python bench.py --gpus N does it itself)")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the hot path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group("nccl", device_id=dev)

    from spvipes_amd import _abi
    from spvipes_amd.data import MinibatchSampler, make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer

    _abi.load()
    torch.manual_seed(0)
    NG = args.groups
    groups = [make_synthetic_group(g, args.cells, args.genes, dev, dtype=args.count_dtype, shard=rank) for g in range(NG)]   # every rank its own cells
    plan = None
    mkw = {}
    if args.poe == "label":
        mkw = dict(use_labels=True)
    else:
        if NG == 2:
            plan = synthetic_plan(args.cells, args.cells)
            mkw = dict(transport_plan=plan, pair_data=(args.poe == "paired"))
        else:   # N-group cluster matching: the experts are the per-component batch statistics, no pairwise plan is needed
            mkw = dict(transport_plan="components", pair_data=False, allow_more_groups=True, n_components=10)
    module = spVIPESmodule({g: args.genes for g in range(NG)}, n_hidden=args.n_hidden, n_dimensions_shared=args.n_shared,
                           n_dimensions_private=args.n_private, precision=args.precision, **mkw).to(dev)
    trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups] if args.poe == "label" else None,
                      components=[g.labels for g in groups] if args.poe == "cluster" else None,
                      overlap_allreduce=None if args.allreduce == "auto" else args.allreduce == "overlap")
    if world > 1:  # identical initial weights on every rank
        dist.broadcast(trainer.fp.flat, src=0)
        trainer.parameters_changed()
    sampler = MinibatchSampler([args.cells] * NG, args.batch_size, dev, seed=rank)
    module.train()

    def batches():
        while True:
            for rows in sampler.epoch():
                yield rows

    it = batches()
    prof_names = ["spv_dec_nb_fwd", "spv_dec_logits", "spv_enc_fc1_fwd", "spv_enc_fc1_wgrad", "spv_dec_lse", "spv_dec_softmax_bwd", "spv_dec_heads_bwd",
                  "spv_gemm_bf16", "spv_dec_heads_wgrad", "spv_adam_step", "spv_adam_step_images", "spv_enc_fc1_fwd_grouped", "spv_enc_fc1_bwd_grouped",
                  "spv_dec_nb_fwd_grouped", "spv_dec_logits_grouped", "spv_dec_lse_grouped", "spv_dec_heads_bwd_grouped", "spv_gemm_bf16_grouped"]
    use_graph = not args.no_graph
    delta = delta_state = None
    if rank == 0 and world == 1 and NG == 2 and not args.no_elbo_delta:
        try:
            delta_state = elbo_delta_gpu(trainer, module, groups, next(it), args)
        except Exception as e:  # a reported extra; never lose the throughput number over it
            delta = {"value": None, "what": f"failed: {e!r}"}
    prof = {}
    if use_graph:
        # A captured hipGraph cannot carry per-call events, so the per-kernel durations for the roofline line are
        # taken with HIP events on the launch stream in an eager pass of the same steps, immediately before the
        # graph is captured and timed (same kernels, same shapes, same data; rocprofv3 averages agree: profiles/).
        # In the timed (captured) step the two groups' chains run on two HIP streams and overlap; a kernel's own
        # duration is only defined when it runs alone, so this event pass keeps every launch on one stream
        # (ops.SERIAL_STREAMS).  `SPV_SERIAL_STREAMS=1 python bench.py ...` runs the whole benchmark that way: that is
        # the command of the rocprofv3 summary whose per-kernel averages these numbers agree with (profiles/README.md).
        from spvipes_amd import ops as _ops
        serial_before = _ops.SERIAL_STREAMS
        _ops.SERIAL_STREAMS = True
        for _ in range(2):
            trainer.step(next(it), kl_weight=1.0)
        torch.cuda.synchronize()
        _abi.profile_start(prof_names)
        for _ in range(max(3, min(args.steps, 10))):
            trainer.step(next(it), kl_weight=1.0)
        prof = _abi.profile_stop()
        _ops.SERIAL_STREAMS = serial_before
        trainer.capture(next(it))
    for _ in range(args.warmup):
        trainer.step(next(it), kl_weight=1.0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    if not use_graph:
        _abi.profile_start(prof_names)  # HIP events on the launch stream around the C-ABI calls, inside the timed region
    # per-step HIP events on the launch stream (no host synchronisation): the median step time SURVEY 8d defines
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    last = None
    ev[0].record()
    for i in range(args.steps):
        last = trainer.step(next(it), kl_weight=1.0)
        ev[i + 1].record()
    host_enqueue_ms = (time.perf_counter() - t0) * 1e3 / args.steps   # what the host spends issuing one step (close to ms_per_step = host bound)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if not use_graph:
        prof = _abi.profile_stop()
    step_ms = sorted(ev[i].elapsed_time(ev[i + 1]) for i in range(args.steps))
    median_ms = step_ms[len(step_ms) // 2]
    prof_steps = args.steps if not use_graph else max(3, min(args.steps, 10))
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    loss = float(last.loss)
    # exposed all-reduce time per step (outside the timed region): GPU time between the end of the backward pass and the
    # start of Adam, max over ranks
    exposed = None
    if world > 1:
        ex = []
        for _ in range(5):
            trainer.step(next(it), kl_weight=1.0)
            ex.append(trainer.last_allreduce_exposed_ms())
        tt = torch.tensor([float(np.median(ex))], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        exposed = float(tt)

    # Weak-scaling reference measured in the same job, in the form a ONE-GPU job runs: rank 0 builds a fresh trainer
    # with world = 1 -- one graph per step, no split backward pass, Adam as the graph's last node -- on its own shard and repeats the
    # warm-up + timed loop; the other ranks wait at the barrier.  value / this = the speed-up over ONE GPU on the SAME per-GPU
    # workload (the default N = 1 run of this script is preset c2, a different shape).
    n1_ref = n1_form = None
    if world > 1:
        torch.cuda.synchronize()
        dist.barrier()
        if rank == 0:
            t1gpu = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups] if args.poe == "label" else None,
                            components=[g.labels for g in groups] if args.poe == "cluster" else None, overlap_allreduce=False, world=1)
            if use_graph:
                t1gpu.capture(next(it))
            for _ in range(args.warmup):
                t1gpu.step(next(it), kl_weight=1.0)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(args.steps):
                t1gpu.step(next(it), kl_weight=1.0)
            torch.cuda.synchronize()
            n1_ref = NG * args.batch_size * args.steps / (time.perf_counter() - t1)
            n1_form = ("one hipGraph per step with Adam as its last node" if getattr(t1gpu, "_adam_in_graph", False) else
                       "one hipGraph per step, Adam launched after it" if use_graph else "eager launches") + ", no collective, no split backward pass"
        dist.barrier()

    if rank == 0 and delta_state is not None:
        try:
            delta = elbo_delta_oracle(delta_state, args, plan)
        except Exception as e:
            delta = {"value": None, "what": f"failed: {e!r}"}
    if rank == 0:
        B, G, H = args.batch_size, args.genes, args.n_hidden
        n_s, n_p = args.n_shared, args.n_private
        cells_per_step = NG * B * world
        value = cells_per_step * args.steps / elapsed
        sx = 2 if args.count_dtype == "u16" else 4
        # dominant kernel = the fused decoder + NB-mixture likelihood forward (per group launch):
        #   algorithmic bytes: the B x G counts once + mixture/regressor weights once (bf16 images)
        #   algorithmic flops: logits GEMM 2*B*G*(256+n_s+n_p+1) + the two regressor GEMMs
        nb_ms = prof.get("spv_dec_nb_fwd", [])
        nb_launch = "dec_nb_kernel (spv_dec_nb_fwd)"
        if not nb_ms and prof.get("spv_dec_nb_fwd_grouped"):   # small steps: ONE grid for both groups (ops.DEC_PAIR) = NG launches' worth of work
            nb_ms = [v / NG for v in prof["spv_dec_nb_fwd_grouped"]]
            nb_launch = f"dec_nb_pair_kernel (spv_dec_nb_fwd_grouped: one grid for {NG} groups; per-group share of its duration)"
        nb_avg = float(np.mean(nb_ms)) if nb_ms else float("nan")
        KM = 256 + n_s + n_p + 1
        nb_bytes = B * G * sx + G * (KM + 2 * (n_s + n_p + 2)) * 2 + B * KM * 2
        nb_flops = 2.0 * B * G * (KM + n_s + n_p + 2)
        per_kernel = {k: {"calls_per_step": len(v) / prof_steps, "avg_ms": float(np.mean(v))} for k, v in prof.items() if v}
        traffic, traffic_src, valu_busy = _pmc_fields(args)
        roof = {"kernel": nb_launch, "bound": "hbm", "achieved": nb_bytes / (nb_avg * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nb_bytes / (nb_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src,
                "limiter": "VALU / transcendental issue and per-wave memory latency, not HBM bandwidth (DESIGN.md section 4)",
                "valu_busy_frac": valu_busy,
                "avg_launch_ms": nb_avg, "algorithmic_bytes_per_launch": nb_bytes,
                "mfma_view": {"achieved_TFLOPs": nb_flops / (nb_avg * 1e-3) / 1e12, "peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS,
                              "frac": nb_flops / (nb_avg * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS},
                "traffic_GBs": (traffic / (nb_avg * 1e-3) / 1e9) if (traffic and nb_avg == nb_avg) else None,
                "decoder_chain": _pmc_decoder_chain(args, NG, (sum(sum(prof.get(k, [])) for k in _CHAIN_ENTRY_POINTS) / prof_steps / NG) or None),
                "per_entry_point": per_kernel}
        fc1_ms = prof.get("spv_enc_fc1_fwd", [])
        if not fc1_ms and prof.get("spv_enc_fc1_fwd_grouped"):   # one launch per kernel for all groups: per-group share of the entry point
            fc1_ms = [v / NG for v in prof["spv_enc_fc1_fwd_grouped"]]
        if fc1_ms:   # the encoder contraction SURVEY 8d prices against the MFMA roofline (whole entry point, epilogue included)
            fc1_avg, fc1_flops, fc1_bytes = float(np.mean(fc1_ms)), 2.0 * B * G * 2 * H, B * G * 2 + 2 * H * G * 2
            roof["encoder_fc1_view"] = {"bound": "mfma", "achieved": fc1_flops / (fc1_avg * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": fc1_flops / (fc1_avg * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "avg_launch_ms": fc1_avg,
                                        "hbm_co_bound_GBs": fc1_bytes / (fc1_avg * 1e-3) / 1e9}
        poe_txt = {"label": "label-based PoE", "paired": "paired-cells PoE on a sparse transport plan", "cluster": "cluster-matched PoE"}[args.poe]
        out = {
            "metric": "cells/sec/training-step (2-group PoE VAE)" if NG == 2 else f"cells/sec/training-step ({NG}-group PoE VAE)",
            "value": value, "unit": "cells/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "ms_per_step_median": median_ms, "host_enqueue_ms_per_step": host_enqueue_ms,
            "higher_is_better": True,
            # dtype, bf16 mode: bf16 decoder operands, f16 encoder fc1 operands, fp32 accumulation
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32(split-bf16)",
            "data": "synthetic",
            "config": {"workload": f"{NG} groups x {args.cells} cells x {G} genes per GPU, {poe_txt}, n_shared={n_s} n_private={n_p} "
                                   f"n_hidden={H}, batch {B} cells/group/step/GPU, counts resident as {args.count_dtype} ({CONFIGS[args.config]['what']})",
                       "preset": args.config,
                       "parallelism": f"dp{world}", "rccl_ranks": dist.get_world_size() if world > 1 else 1, "precision": args.precision, "launch": "hipGraph replay, 2 streams" if use_graph else "eager",
                       "allreduce": ("none (1 rank)" if world == 1 else "2 buckets, decoder bucket overlapped with the encoder backward"
                                     if trainer.overlap else "1 bucket after the backward pass (north_star form)"),
                       "allreduce_exposed_ms_per_step": exposed,
                       "one_gpu_same_workload_cells_per_s": n1_ref},   # rank 0 alone as a one-GPU job would run it (N > 1 runs only)
            "final_loss": loss,
            "elbo_delta": delta,
            "roofline": roof,
        }
        if world > 1:
            # N > 1 lines default to preset c3 (the shape BASELINE quotes the scaling target on) while the N = 1 line of the same script is
            # preset c2: a ratio of the two `value`s mixes workloads.  The one-GPU number of THIS line's per-GPU workload, measured in this
            # job (rank 0 alone, collectives off), is repeated here at top level so that it cannot be missed.
            out["n1_same_workload"] = {"value": n1_ref, "unit": "cells/s", "preset": args.config,
                                       "what": "rank 0 alone on the same per-GPU workload with a fresh world-1 trainer (" + str(n1_form) + "), "
                                               "measured in this job after the timed region; the data-parallel ranks run the two-graph split-backward step"}
        if world == 1 and not args.no_cpu_baseline and NG == 2:
            try:
                out["cpu_baseline"] = cpu_baseline(groups, args, plan)
            except Exception as e:  # the baseline is a reported extra; never lose the GPU number over it
                out["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": host_cpus()[0], "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
