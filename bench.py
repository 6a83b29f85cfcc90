#!/usr/bin/env python3
"""Headline benchmark: training-step throughput (cells/s) of the 2-group PoE VAE on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one optimisation step of the hot path on one synthetic minibatch per group
(forward through both encoders, label-based PoE, decoder + NB-mixture ELBO, backward, one gradient
all-reduce when N > 1, Adam).  Workload = BASELINE.json configs[1]: 2 groups x 50 000 cells x
10 000 genes, n_shared 25, n_private 10, n_hidden 128, bf16 MFMA operands with fp32 accumulation;
the configuration does not fix the minibatch, we use 4096 cells per group per step (the batch size
BASELINE.json names for its 8-GPU configuration).  Weak scaling: every rank holds its own
50 000-cell shard per group and draws its own minibatches; value = cells processed by ALL ranks / s.
The count matrices are resident in HBM (uint16) before the timed region.

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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--cells", type=int, default=50_000, help="cells per group per rank")
    ap.add_argument("--genes", type=int, default=10_000, help="genes per group")
    ap.add_argument("--batch-size", type=int, default=4096, help="cells per group per step per rank")
    ap.add_argument("--n-hidden", type=int, default=128)
    ap.add_argument("--n-shared", type=int, default=25)
    ap.add_argument("--n-private", type=int, default=10)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--count-dtype", default="u16", choices=["u16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true", help="launch every kernel eagerly instead of replaying a captured hipGraph")
    ap.add_argument("--allreduce", default="auto", choices=["auto", "single", "overlap"],
                    help="gradient all-reduce of a data-parallel job: one collective after the backward pass, or two buckets with the "
                         "decoder bucket overlapped with the encoder half of the backward pass (auto: overlap when WORLD_SIZE > 1)")
    ap.add_argument("--cpu-batch", type=int, default=256)
    ap.add_argument("--cpu-steps", type=int, default=2)
    return ap.parse_args()


def cpu_baseline(groups, args, n_threads):
    """The CPU oracle (oracle/spvipes_oracle.py, a port of the reference path pinned by goldens
    generated from the reference) timed on this box's host cores: forward + loss + backward + Adam
    on a bounded sample of the same workload."""
    from oracle import spvipes_oracle as O

    torch.set_num_threads(n_threads)
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
        counts, labels = [], []
        for g in groups:
            rows = torch.randint(0, g.counts.n_cells, (B,), generator=gen)
            X = g.counts.X[rows.to(g.counts.X.device)].cpu()
            X = torch.from_numpy(X.numpy().view(np.uint16).astype(np.float32)) if X.dtype == torch.int16 else X
            counts.append(X)
            labels.append(g.labels[rows.to(g.labels.device)].cpu())
        noise = {f"enc_{i}_{k}": torch.randn(B, n, generator=gen) for i in range(2) for k, n in (("private", args.n_private), ("shared", args.n_shared))}
        noise.update({f"poe_{i}": torch.randn(B, args.n_shared, generator=gen) for i in range(2)})
        t0 = time.perf_counter()
        out = O.forward_loss(sd, counts, n_dimensions_shared=args.n_shared, n_dimensions_private=args.n_private, noise=noise,
                             mode="label", labels=labels, training=True)
        opt.zero_grad()
        out["loss"].backward()
        opt.step()
        if step > 0:  # first step pays allocator warm-up
            times.append(time.perf_counter() - t0)
    sec = float(np.mean(times))
    return {"value": 2 * B / sec, "unit": "cells/s", "cores": n_threads, "kind": "port",
            "sample": f"{args.cpu_steps} steps of 2 x {B} cells x {Gs[0]} genes (same synthetic data, fwd+loss+bwd+Adam), {sec:.2f} s/step"}



def _pmc_traffic():
    """HBM bytes per launch of the dominant kernel = FETCH_SIZE + WRITE_SIZE (KiB) of the newest committed rocprofv3 --pmc
    passes under profiles/ (separate passes of this same command; bench.py itself cannot run under the profiler).  The
    kernel's reads are 8 B / lane: outside the access widths the gfx950 FETCH_SIZE correction is calibrated for, so the
    raw counter is reported."""
    import glob, re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_dec_nb_kernel.txt")))
    if not files:
        return None, None
    txt = open(files[-1]).read()
    f = re.search(r"FETCH_SIZE\s+n=\s*\d+\s+avg=\s*([0-9.]+)", txt)
    w = re.search(r"WRITE_SIZE\s+n=\s*\d+\s+avg=\s*([0-9.]+)", txt)
    if not (f and w):
        return None, None
    return (float(f.group(1)) + float(w.group(1))) * 1024.0, "profiles/" + os.path.basename(files[-1]) + " (raw FETCH_SIZE + WRITE_SIZE, KiB -> bytes)"


def _pmc_valu_busy():
    """Fraction of the dominant kernel's cycles in which a SIMD issues VALU work, from the same committed --pmc passes:
    SQ_ACTIVE_INST_VALU (quad-cycles summed over the 1024 SIMDs) * 4 / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs.  The
    likelihood kernel is bound by VALU / transcendental issue, not by HBM: this, not `frac`, is how close it runs to its
    own ceiling."""
    import glob, re
    files = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "profiles", "r*_pmc_dec_nb_kernel.txt")))
    if not files:
        return None
    txt = open(files[-1]).read()
    a = re.search(r"SQ_ACTIVE_INST_VALU\s+n=\s*\d+\s+avg=\s*([0-9.]+)", txt)
    g = re.search(r"GRBM_GUI_ACTIVE\s+n=\s*\d+\s+avg=\s*([0-9.]+)", txt)
    if not (a and g):
        return None
    return (float(a.group(1)) * 4.0 / 1024.0) / (float(g.group(1)) / 8.0)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
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
    groups = [make_synthetic_group(g, args.cells, args.genes, dev, dtype=args.count_dtype) for g in range(2)]
    module = spVIPESmodule({0: args.genes, 1: args.genes}, use_labels=True, n_hidden=args.n_hidden,
                           n_dimensions_shared=args.n_shared, n_dimensions_private=args.n_private, precision=args.precision).to(dev)
    trainer = Trainer(module, [g.counts for g in groups], labels=[g.labels for g in groups],
                      overlap_allreduce=None if args.allreduce == "auto" else args.allreduce == "overlap")
    if world > 1:  # identical initial weights on every rank
        dist.broadcast(trainer.fp.flat, src=0)
    sampler = MinibatchSampler([args.cells, args.cells], args.batch_size, dev, seed=rank)
    module.train()

    def batches():
        while True:
            for rows in sampler.epoch():
                yield rows

    it = batches()
    prof_names = ["spv_dec_nb_fwd", "spv_dec_logits", "spv_enc_fc1_fwd", "spv_enc_fc1_wgrad", "spv_dec_lse", "spv_dec_softmax_bwd",
                  "spv_gemm_bf16", "spv_adam_step"]
    use_graph = not args.no_graph
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
    t0 = time.perf_counter()
    last = None
    for _ in range(args.steps):
        last = trainer.step(next(it), kl_weight=1.0)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if not use_graph:
        prof = _abi.profile_stop()
    prof_steps = args.steps if not use_graph else max(3, min(args.steps, 10))
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t)
    loss = float(last.loss)

    if rank == 0:
        B, G, H = args.batch_size, args.genes, args.n_hidden
        n_s, n_p = args.n_shared, args.n_private
        cells_per_step = 2 * B * world
        value = cells_per_step * args.steps / elapsed
        sx = 2 if args.count_dtype == "u16" else 4
        # dominant kernel = the fused decoder + NB-mixture likelihood forward (per group launch):
        #   algorithmic bytes: the B x G counts once + mixture/regressor weights once (bf16 images)
        #   algorithmic flops: logits GEMM 2*B*G*(256+n_s+n_p+1) + the two regressor GEMMs
        nb_ms = prof.get("spv_dec_nb_fwd", [])
        nb_avg = float(np.mean(nb_ms)) if nb_ms else float("nan")
        KM = 256 + n_s + n_p + 1
        nb_bytes = B * G * sx + G * (KM + 2 * (n_s + n_p + 2)) * 2 + B * KM * 2
        nb_flops = 2.0 * B * G * (KM + n_s + n_p + 2)
        per_kernel = {k: {"calls_per_step": len(v) / prof_steps, "avg_ms": float(np.mean(v))} for k, v in prof.items() if v}
        traffic, traffic_src = _pmc_traffic()
        roof = {"kernel": "dec_nb_kernel (spv_dec_nb_fwd)", "bound": "hbm", "achieved": nb_bytes / (nb_avg * 1e-3) / 1e9,
                "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": nb_bytes / (nb_avg * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": traffic_src,
                "valu_busy_frac": _pmc_valu_busy(),
                "avg_launch_ms": nb_avg, "algorithmic_bytes_per_launch": nb_bytes,
                "mfma_view": {"achieved_TFLOPs": nb_flops / (nb_avg * 1e-3) / 1e12, "peak_TFLOPs": MFMA_BF16_PEAK_TFLOPS,
                              "frac": nb_flops / (nb_avg * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS},
                "per_entry_point": per_kernel}
        fc1_ms = prof.get("spv_enc_fc1_fwd", [])
        if fc1_ms:   # the encoder contraction SURVEY 8d prices against the MFMA roofline (entry point = split-K GEMM + bias/ReLU/slab-sum kernel)
            fc1_avg, fc1_flops, fc1_bytes = float(np.mean(fc1_ms)), 2.0 * B * G * 2 * H, B * G * 2 + 2 * H * G * 2
            roof["encoder_fc1_view"] = {"bound": "mfma", "achieved": fc1_flops / (fc1_avg * 1e-3) / 1e12, "peak": MFMA_BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                        "frac": fc1_flops / (fc1_avg * 1e-3) / 1e12 / MFMA_BF16_PEAK_TFLOPS, "avg_launch_ms": fc1_avg,
                                        "hbm_co_bound_GBs": fc1_bytes / (fc1_avg * 1e-3) / 1e9}
        out = {
            "metric": "cells/sec/training-step (2-group PoE VAE)", "value": value, "unit": "cells/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16" if args.precision == "bf16" else "f32(split-bf16)", "data": "synthetic",
            "config": {"workload": f"2 groups x {args.cells} cells x {G} genes per GPU, label-based PoE, n_shared={n_s} n_private={n_p} "
                                   f"n_hidden={H}, batch {B} cells/group/step/GPU, counts resident as {args.count_dtype} (BASELINE configs[1])",
                       "parallelism": f"dp{world}", "precision": args.precision, "launch": "hipGraph replay, 2 streams" if use_graph else "eager",
                       "allreduce": ("none (1 rank)" if world == 1 else "2 buckets, decoder bucket overlapped with the encoder backward"
                                     if trainer.overlap else "1 bucket after the backward pass")},
            "final_loss": loss,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(groups, args, min(os.cpu_count() or 1, 16))
            except Exception as e:  # the baseline is a reported extra; never lose the GPU number over it
                out["cpu_baseline"] = {"value": None, "unit": "cells/s", "cores": os.cpu_count(), "kind": "port", "sample": f"failed: {e!r}"}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
