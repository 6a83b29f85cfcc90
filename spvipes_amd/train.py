"""Training step driver: flat parameter / gradient buffers, HIP Adam, one RCCL all-reduce per step.

Stands in for what the reference delegates to scvi-tools + Lightning
(/root/reference/src/spVIPES/model/base/training_mixin.py:89-123): ``TrainingPlan`` = Adam(lr 1e-3,
eps 0.01, weight_decay 1e-6) with a linear KL warm-up (0 -> 1 over ``n_epochs_kl_warmup`` = 400
epochs; over ``n_steps_kl_warmup`` steps only when no epoch count is given), ``max_epochs = min(round(20000 / n_obs *
400), 400)``.

Data parallelism (the reference has none): one process per GPU, every rank draws its
own minibatch from its shard of the cells, gradients live in ONE flat fp32 buffer (decoder
parameters first, encoder parameters last) that is summed over the ranks with
``torch.distributed.all_reduce`` (backend "nccl" = RCCL over xGMI) in two contiguous buckets: the
decoder bucket is reduced while the encoder half of the backward pass still runs, the encoder bucket
right after it; the 1/world factor is folded into the Adam kernel.  BatchNorm statistics and PoE
pairing stay rank-local.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import os

import torch
import torch.distributed as dist

from . import _abi
from .data import MinibatchSampler
from .module import spVIPESmodule
from .ops import GroupCounts


def kl_weight_at(epoch: int, step: int, n_epochs_kl_warmup: Optional[int], n_steps_kl_warmup: Optional[int],
                 max_kl_weight: float = 1.0, min_kl_weight: float = 0.0) -> float:
    """KL weight of scvi-tools 0.20.0 ``TrainingPlan.kl_weight`` (``_compute_kl_weight`` in scvi/train/_trainingplans.py,
    the library version pinned by the reference's pyproject): the EPOCH criterion is checked first and the step criterion
    is used only when ``n_epochs_kl_warmup`` is None / 0 -- the reference's own docstring (training_mixin.py:67-68, "steps
    take precedence") describes the opposite of what the library it calls does; with its default n_epochs_kl_warmup=400 a
    user-supplied n_steps_kl_warmup is therefore ignored, and so it is here."""
    slope = max_kl_weight - min_kl_weight
    if n_epochs_kl_warmup:
        if epoch < n_epochs_kl_warmup:
            return min_kl_weight + slope * (epoch / n_epochs_kl_warmup)
    elif n_steps_kl_warmup:
        if step < n_steps_kl_warmup:
            return min_kl_weight + slope * (step / n_steps_kl_warmup)
    return max_kl_weight


def default_max_epochs(n_obs: int) -> int:
    """training_mixin.py:89-91."""
    return int(min(round((20000 / n_obs) * 400), 400))


def sync_float_buffers(module: torch.nn.Module, world: int) -> None:
    """Average every floating-point buffer of ``module`` (the BatchNorm running means / variances) over the ranks with ONE
    all-reduce.  Parameters are identical on all ranks by construction (same reduced gradient, same Adam step); BatchNorm running
    statistics are not -- every rank updates them from its own minibatches and they are in no gradient bucket -- so eval-mode
    forward passes (validation, ``get_latent_representation``) would otherwise see a different model on every rank."""
    if world <= 1:
        return
    bufs = [b for b in module.buffers() if b.is_floating_point() and b.numel()]
    if not bufs:
        return
    flat = torch.cat([b.detach().reshape(-1).to(torch.float32) for b in bufs])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    flat.div_(world)
    off = 0
    with torch.no_grad():
        for b in bufs:
            b.copy_(flat[off:off + b.numel()].view_as(b))
            off += b.numel()


class EarlyStopping:
    """scvi-tools / Lightning ``EarlyStopping`` on a minimised metric (patience, min_delta) whose decision is COLLECTIVE: with more
    than one rank the monitored value is averaged over the ranks before it is compared, so every rank sees the same number and
    leaves the epoch loop in the same epoch (a rank that stopped alone would leave its peers blocked in the next all-reduce)."""

    def __init__(self, patience: int, min_delta: float, world: int = 1, device=None):
        self.patience, self.min_delta, self.world, self.device = int(patience), float(min_delta), int(world), device
        self.best, self.bad_epochs = float("inf"), 0

    def reduced(self, value: float) -> float:
        if self.world <= 1:
            return float(value)
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return float(t) / self.world

    def should_stop(self, value: float) -> bool:
        v = self.reduced(value)
        if v < self.best - self.min_delta:
            self.best, self.bad_epochs = v, 0
            return False
        self.bad_epochs += 1
        return self.bad_epochs >= self.patience


class FlatParams:
    """Re-homes every trainable parameter of a module into one contiguous fp32 buffer (and its
    gradient into a second one), so that the all-reduce and the optimiser each touch one array."""

    def __init__(self, module: torch.nn.Module, late=None):
        """``late(name) -> bool`` marks parameters whose gradients are produced last in the backward pass; they are
        placed at the end of the buffers and ``self.split`` is the offset of the first of them (bucket boundary of
        the data-parallel all-reduce)."""
        named = [(n, p) for n, p in module.named_parameters() if p.requires_grad]
        if not named:
            raise ValueError("module has no trainable parameters")
        early = [p for n, p in named if late is None or not late(n)]
        params = early + [p for n, p in named if late is not None and late(n)]
        dev = params[0].device
        sizes = [(p.numel() + 3) // 4 * 4 for p in params]  # keep every view 16-byte aligned
        total = sum(sizes)
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(total, dtype=torch.float32, device=dev)
        self.split = sum(sizes[:len(early)])
        off = 0
        for p, n in zip(params, sizes):
            view = self.flat[off:off + p.numel()].view_as(p)
            view.copy_(p.data)
            p.data = view
            p.grad = self.grad[off:off + p.numel()].view_as(p)
            off += n
        self.params, self.numel = params, total

    def zero_grad(self):
        self.grad.zero_()


class HipAdam:
    """torch.optim.Adam semantics on a FlatParams buffer, one kernel launch (spv_adam_step_images).  The number of steps taken
    lives on the DEVICE (``t_dev``): the kernel forms the bias corrections 1 - beta^t itself, so no argument of the launch changes
    from step to step and the launch can be captured into the step's hipGraph (``Trainer.capture``, single-rank jobs)."""

    def __init__(self, fp: FlatParams, lr=1e-3, betas=(0.9, 0.999), eps=0.01, weight_decay=1e-6):
        self.fp, self.lr, self.betas, self.eps, self.wd = fp, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(fp.flat)
        self.v = torch.zeros_like(fp.flat)
        self.t = 0   # host copy of the step count (bookkeeping only: the kernel reads t_dev)
        self.t_dev = torch.zeros((), dtype=torch.int64, device=fp.flat.device)

    def step(self, grad_scale: float = 1.0, images=None, counter=None):
        """``images``: optional ctypes array of SpvAdamImage -- 16-bit operand images the kernel rewrites from the updated values;
        ``counter``: optional 0-dim int64 device tensor the kernel increments (the step counter the module's noise is keyed by)"""
        self.t += 1
        self.launch(grad_scale, images, counter)

    def launch(self, grad_scale: float = 1.0, images=None, counter=None):
        """the two launches of one step (Adam, then the device step count + 1); captured as they are by ``Trainer.capture``"""
        b1, b2 = self.betas
        n_img = len(images) if images is not None else 0
        _abi.call("spv_adam_step_images", _abi.ptr(self.fp.flat), _abi.ptr(self.fp.grad), _abi.ptr(self.m), _abi.ptr(self.v),
                  self.fp.numel, self.lr, b1, b2, self.eps, self.wd, 1.0, 1.0,
                  grad_scale, images if n_img else None, n_img, _abi.ptr(counter), _abi.ptr(self.t_dev), float(b1), float(b2), _abi.stream_ptr())
        _abi.call("spv_counter_bump", _abi.ptr(self.t_dev), _abi.stream_ptr())


class Trainer:
    """Runs optimisation steps of a spVIPESmodule on device-resident count matrices."""

    def __init__(self, module: spVIPESmodule, counts: Sequence[GroupCounts], labels: Optional[Sequence[torch.Tensor]] = None,
                 components: Optional[Sequence[torch.Tensor]] = None, lr: float = 1e-3, eps: float = 0.01,
                 weight_decay: float = 1e-6, n_epochs_kl_warmup: Optional[int] = 400, n_steps_kl_warmup: Optional[int] = None,
                 overlap_allreduce: Optional[bool] = None, batch_codes: Optional[Sequence[torch.Tensor]] = None, world: Optional[int] = None):
        """``world``: number of ranks whose gradients are summed (default: the size of the initialised process group; ``world=1``
        inside a data-parallel job gives the trainer a single-GPU job would build -- bench.py's one-GPU reference on rank 0)."""
        self.module, self.counts, self.labels, self.components = module, list(counts), labels, components
        # batch covariates (module built with n_batch > 1): one integer code per cell of every group, gathered per minibatch
        self.batch_codes = None
        if getattr(module, "n_cov", 0):
            if batch_codes is None or len(batch_codes) != len(self.counts):
                raise ValueError(f"the module was built with n_batch = {module.n_batch}: pass batch_codes (one code per cell and group)")
            self.batch_codes = [b.flatten().to(device=counts[0].X.device, dtype=torch.int32).contiguous() for b in batch_codes]
            for g, b in enumerate(self.batch_codes):   # (the fc1 epilogue indexes its covariate table with them: validated once, here)
                if b.numel() and not bool(((b >= 0) & (b < module.n_cov)).all()):
                    raise ValueError(f"batch codes of group {g} must lie in [0, n_batch = {module.n_batch})")
        for what, codes in (("labels", labels), ("cluster components", components)):
            if codes is None:   # (the pairing kernels index per-code tables: validated once, here, not per step)
                continue
            for g, l in enumerate(codes):
                lf = l.flatten().to(torch.float32)
                if lf.numel() and not bool(((lf >= 0) & (lf < _abi.POE_LMAX) & (lf == lf.floor())).all()):
                    raise ValueError(f"{what} of group {g} must be integral codes in [0, {_abi.POE_LMAX})")
                n_comp = getattr(module, "n_components", None)
                if what == "cluster components" and module.n_groups > 2 and n_comp is not None and lf.numel() and float(lf.max()) >= n_comp:
                    raise ValueError(f"cluster components of group {g} must be codes in [0, n_components = {n_comp})")
        self.device = counts[0].X.device
        self.fp = FlatParams(module, late=lambda name: name.startswith("encoder_"))
        self.opt = HipAdam(self.fp, lr=lr, eps=eps, weight_decay=weight_decay)
        self.n_epochs_kl_warmup, self.n_steps_kl_warmup = n_epochs_kl_warmup, n_steps_kl_warmup
        self._pg_world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
        self.world = self._pg_world if world is None else int(world)
        # two gradient buckets with the backward pass split between them (default: whenever there is more than one rank)
        self.overlap = (self.world > 1) if overlap_allreduce is None else bool(overlap_allreduce)
        self.global_step, self.epoch = 0, 0
        if self.DEVICE_RNG:
            # the module draws its noise (and keys its dropout masks) from a counter-based generator on the device: key from torch's
            # seed + rank, counter incremented by the Adam launch -- see spv_randn; a replayed graph then needs no host generator state
            rank = dist.get_rank() if self.world > 1 else 0   # ranks share the key and start their counters 2^40 apart
            module.enable_device_rng(self.device, key=(int(torch.initial_seed()) * 0x9E3779B97F4A7C15 + 1) & 0xFFFFFFFFFFFFFFFF, counter0=rank << 40)
        self.graph = self.graph2 = None
        self.last_outputs = None
        self.history: Dict[str, List[float]] = {"train_loss": [], "elbo_train": [], "reconstruction_loss_train": [], "kl_local_train": []}

    # ---- bf16 weight images kept in step by the optimiser -------------------------------------------------------------------
    # The step starts by repacking the fc1 weights of every encoder and [W_m | b_m] of every decoder into bf16 operand images
    # (6 spv_pack_bf16 launches, ~45 us of HBM time at C2, the first 30 us of them ahead of the fc1 GEMMs).  In bf16 mode the Adam
    # kernel writes those images itself from the values it has just computed (spv_adam_step_images, +29 MB of stores next to its
    # 330 MB), the workspaces are told so (ops.Workspace.fresh) and the forward pass skips the packs -- also inside the captured
    # graph, which is captured with the images fresh.  Any torch-side write to one of the parameters (load_state_dict, broadcast)
    # changes its version counter: the token no longer matches and ``_ensure_images`` repacks before the next step.
    IMAGES_BY_ADAM = os.environ.get("SPV_ADAM_IMAGES", "1") != "0"
    # Every kernel that produces a parameter gradient OVERWRITES its slice of the flat gradient buffer (the "gradient sink": one
    # contribution per parameter per step, slab sums written with accumulate off), so the 47 MB fill at the start of a step (9 us at
    # C2, at the head of the critical path) is not needed; the buffer is zero from construction, alignment padding stays zero.
    # tests/test_gpu_train_and_model.py::test_every_gradient_element_is_overwritten_by_a_step pins that property (NaN-filled buffer).
    ZERO_GRADS_EACH_STEP = os.environ.get("SPV_ZERO_GRADS", "0") != "0"
    DEVICE_RNG = os.environ.get("SPV_DEVICE_RNG", "1") != "0"
    MULTI_GATHER = os.environ.get("SPV_MULTI_GATHER", "1") != "0"   # the groups' label gathers / row-index copies as one launch each (spv_gather_u32)
    ADAM_IN_GRAPH = os.environ.get("SPV_ADAM_IN_GRAPH", "1") != "0"  # single rank: the Adam launch is the captured graph's last node

    def _image_specs(self):
        """[(workspace, key, image tensor, [(parameter, rows_off, col_off)], token parameters)] for every packed weight image"""
        from . import ops
        from .dec_ops import KMP

        m = self.module
        out = []
        for g in range(m.n_groups):
            ws = m._workspace(g, self.device)
            ep, es = m.encoders[g]["private"], m.encoders[g]["shared"]
            H, G = ep.fc1.weight.shape
            N1 = 2 * H
            bn = 32 if N1 <= 32 else (128 if N1 <= 128 else 256)
            img = ops._bf16_image(ws, "fc1_W", ops.round_up(N1, bn), ops.round_up(G, 64), False)[0]
            out.append((ws, "fc1_W", img, [(ep.fc1.weight, 0, 0), (es.fc1.weight, H, 0)], (ep.fc1.weight, es.fc1.weight)))   # (f16 words of W * ops.fc1_w_scale())
            dec = m.decoders[g]
            Wm, bm = dec.mixture.linear.weight, dec.mixture.linear.bias
            img = ops._bf16_image(ws, "dec_Wm", ops.round_up(G, 256), KMP, False)[0]
            out.append((ws, "dec_Wm", img, [(Wm, 0, 0), (bm, 0, Wm.shape[1])], (Wm, bm)))
        return out

    def _build_image_plan(self):
        if not self.IMAGES_BY_ADAM or self.module.nsplit != 1 or getattr(self.module, "n_cov", 0):
            # (covariate columns: the images are not the parameters' own column layout -- the forward pass packs them every step)
            self._img_specs, self._img_plan = [], None
            return
        from . import ops
        specs = self._image_specs()
        offs = {}
        for p_ in self.fp.params:
            offs[id(p_)] = (p_.data_ptr() - self.fp.flat.data_ptr()) // 4
        descs = []
        for ws, key, img, parts, _tok in specs:
            for par, row_off, col_off in parts:
                if id(par) not in offs:   # (a frozen parameter: nothing to keep in step -- the plain pack path stays)
                    self._img_specs, self._img_plan = [], None
                    return
                d = _abi.SpvAdamImage()
                d.begin, d.count = offs[id(par)], par.numel()
                d.cols = par.shape[1] if par.dim() == 2 else 1
                d.row_off, d.col_off, d.ld, d.dst = row_off, col_off, img.shape[1], _abi.ptr(img)
                d.fmt, d.scale = (_abi.IMAGE_F16, ops.fc1_w_scale()) if key == "fc1_W" else (_abi.IMAGE_BF16, 1.0)
                descs.append(d)
        if not descs or len(descs) > _abi.ADAM_MAX_IMAGES:
            self._img_specs, self._img_plan = [], None
            return
        self._img_specs, self._img_plan = specs, (_abi.SpvAdamImage * len(descs))(*descs)

    def _images_are_fresh(self) -> bool:
        from .ops import image_token
        if getattr(self, "_flat_version", None) != self.fp.flat._version:   # a torch-side write to the flat buffer itself
            return False
        return all(ws.fresh.get(key) == image_token(*tok) for ws, key, _img, _parts, tok in self._img_specs)

    def _mark_images_fresh(self) -> None:
        from .ops import image_token
        for ws, key, _img, _parts, tok in self._img_specs:
            ws.fresh[key] = image_token(*tok)
        self._flat_version = self.fp.flat._version

    def parameters_changed(self) -> None:
        """Tell the trainer that parameter values were written behind torch's back (a collective into ``fp.flat``, a raw kernel):
        the packed bf16 weight images are rebuilt before the next step."""
        for ws, key, _img, _parts, _tok in getattr(self, "_img_specs", []):
            ws.fresh.pop(key, None)

    def _ensure_images(self) -> None:
        """(re)pack every image from the current parameters unless the workspaces already hold them"""
        from . import ops
        if getattr(self, "_img_plan", "unset") == "unset":
            self._build_image_plan()
        if self._img_plan is None or self._images_are_fresh():
            return
        for ws, key, img, parts, _tok in self._img_specs:
            if key == "fc1_W":
                (wp, _, _), (wsh, H, _) = parts
                ops._pack_fc1_weights(wp, wsh, img, None, H)
            else:
                (Wm, _, _), (bm, _, _) = parts
                ops._pack(Wm, img, None, extra_col=bm)
        self._mark_images_fresh()

    def minibatch(self, rows: Sequence[torch.Tensor]):
        """tensors_by_group in the resident layout for the given per-group row indices."""
        out = []
        need_idx = bool(getattr(self.module, "use_transport_plan", False))
        if self.labels is not None and getattr(self, "_labels_f32", None) is None:
            self._labels_f32 = [l.flatten().to(torch.float32).contiguous() for l in self.labels]  # once: the PoE kernel reads fp32 codes
        lab = None
        if self.labels is not None:   # labels[rows] of every group in ONE launch (spv_gather_u32) when the indices are int32 device tensors
            if self.MULTI_GATHER and all(r.dtype == torch.int32 and r.is_cuda and r.is_contiguous() for r in rows):
                lab = [torch.empty(r.numel(), dtype=torch.float32, device=r.device) for r in rows]
                _abi.gather_u32([(self._labels_f32[g], r, lab[g]) for g, r in enumerate(rows)])
            else:
                lab = [self._labels_f32[g].index_select(0, r) for g, r in enumerate(rows)]
        for g, r in enumerate(rows):
            d = {"counts": self.counts[g], "rows": r, "groups": None, "batch": None}
            if self.batch_codes is not None:
                d["batch"] = self.batch_codes[g].index_select(0, r)
            d["indices"] = r.to(torch.float32).unsqueeze(1) if need_idx else None
            if lab is not None:
                d["labels"] = lab[g].unsqueeze(1)
            if self.components is not None:
                d["processed_transport_labels"] = self.components[g][r.long()].unsqueeze(1)
            out.append(d)
        return tuple(out)

    def _forward_backward(self, rows, kl_weight, noise=None):
        """forward + loss + backward.  With ``self.overlap`` only the decoder half of the backward pass runs here (down
        to the tensors that cross from the encoders into the decoder); ``_backward_encoders`` finishes it.
        ``noise``: injected standard-normal draws (parity tests; see spVIPESmodule.inference)."""
        from . import nn_ops, ops

        if self.ZERO_GRADS_EACH_STEP:
            self.fp.grad.zero_()
        self.module.split_backward = self.overlap   # (only for this call: a plain module(...) elsewhere keeps one backward pass)
        try:
            inf, gen, lo = self.module(self.minibatch(rows), inference_kwargs=({"noise": noise} if noise is not None else None),
                                       loss_kwargs={"kl_weight": kl_weight})
            self.last_outputs = (inf, gen)   # (inference_outputs, generative_outputs) of the latest step
        finally:
            self.module.split_backward = False
        nn_ops.GRAD_SINK = True  # small-layer gradients land directly in the flat buffer (see nn_ops.grad_out)
        ops.DEFER_JOIN = True    # side-stream gradient GEMMs are joined here, after the whole backward pass
        try:
            one = getattr(self, "_one", None)   # the backward seed d loss / d loss = 1, kept: autograd would otherwise launch a fill kernel per step
            if one is None or one.device != lo.loss.device:
                one = self._one = torch.ones((), dtype=torch.float32, device=lo.loss.device)
            cut = self.module._cut
            if cut is not None:
                # split backward pass: the gradients of the tensors that cross from the encoders into the decoder are TAKEN from the
                # engine (autograd.grad), not accumulated into leaf .grad fields -- accumulation clones a gradient that reaches several
                # leaves (the four KL vectors share one), and inside a captured step a clone is a memcpy NODE; like the memset nodes
                # (profiles/r03_graph_edges_*.json) those are kept off the captured path.  Parameter gradients need no
                # accumulation either: with the gradient sink every kernel writes its slice of the flat buffer itself.
                # With batch covariates two decoder weights reach the fused op through a torch-side node (module.CovariateColumns)
                # whose only descendant is the parameter itself: an engine asked for the cut tensors alone never runs it and the
                # parameter's slice of the gradient buffer keeps last step's values.  Asking for the decoder-side parameters as well
                # makes every such node part of the pass; their kernels still write the gradient sink (the engine gets None back).
                extra = ([p for n, p in self.module.named_parameters() if p.requires_grad and not n.startswith("encoder_")]
                         if getattr(self.module, "n_cov", 0) else [])
                grads = torch.autograd.grad(lo.loss, list(cut[1]) + extra, grad_outputs=one, allow_unused=True)
                self._cut_grads = grads[:len(cut[1])]
                for p, g in zip(extra, grads[len(cut[1]):]):
                    if g is not None:   # (a node that could not use the sink handed its gradient to the engine instead)
                        p.grad.copy_(g)
            else:
                lo.loss.backward(gradient=one)
        finally:
            nn_ops.GRAD_SINK = False
            ops.DEFER_JOIN = False
            ops.join_pending()
            ops.join_all_side_streams(self.device)
        return lo

    def _backward_encoders(self):
        """second half of a split backward pass: PoE, encoder tails, fc1 (gradient bucket ``fp.grad[fp.split:]``)"""
        from . import nn_ops, ops

        outs, _cut = self.module._cut
        pairs = [(o, g) for o, g in zip(outs, self._cut_grads) if g is not None and o.requires_grad]
        nn_ops.GRAD_SINK = True
        ops.DEFER_JOIN = True
        try:
            torch.autograd.backward([o for o, _ in pairs], [g for _, g in pairs])
        finally:
            nn_ops.GRAD_SINK = False
            ops.DEFER_JOIN = False
            ops.join_pending()
            ops.join_all_side_streams(self.device)
        self.module._cut = self._cut_grads = None

    def capture(self, rows: Sequence[torch.Tensor], warmup: int = 3) -> None:
        """Capture forward + loss + backward of one step into a hipGraph (``torch.cuda.CUDAGraph``): the ~100
        launches of a step replay as one submission.  Everything that changes between steps lives in device
        memory: the row indices (copied into static buffers), the KL weight (0-dim tensor) and the step counter that keys the
        dropout masks and the noise (module._rng_counter; torch's graph-safe generator when DEVICE_RNG is off).  The all-reduce and Adam stay outside; in a
        data-parallel job the step is captured as TWO graphs (decoder half / encoder half of the backward pass, same
        memory pool) so that the first gradient bucket is on the wire while the second graph runs."""
        from . import ops

        if self.graph is not None:
            return
        self._ensure_images()   # the graph is captured WITHOUT the weight-image packs (see IMAGES_BY_ADAM)
        self._static_rows = [r.clone() for r in rows]
        self._klw = torch.ones((), dtype=torch.float32, device=self.device)
        self._klw_host = 1.0
        side = torch.cuda.Stream(device=self.device)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):  # allocates every workspace buffer before the capture
                self._forward_backward(self._static_rows, self._klw)
                if self.overlap:
                    self._backward_encoders()
        torch.cuda.current_stream().wait_stream(side)
        # In a data-parallel job the process group's watchdog thread polls the events of collectives still in flight; a
        # capture in the default "global" error mode makes such a call from ANOTHER thread an error (HIP refuses event
        # queries while any global-mode capture is open).  Drain the device first and capture in thread-local mode.
        mode = {}
        if self._pg_world > 1:
            torch.cuda.synchronize(self.device)
            mode = {"capture_error_mode": "thread_local"}
        # SPV_GRAPH_KEEP=1 (tools/graph_dump.py): keep the captured hipGraph_t alive beside its executable, so that its nodes and
        # edges can be read back (CUDAGraph.raw_cuda_graph) -- what the replay honours is then on record, not inferred
        keep = os.environ.get("SPV_GRAPH_KEEP") == "1"
        g = torch.cuda.CUDAGraph(keep_graph=True) if keep else torch.cuda.CUDAGraph()
        # single-rank jobs: the optimiser launch is the LAST node of the graph (nothing has to happen between the backward pass and
        # Adam; every per-step value it needs is device resident: HipAdam.t_dev, module._rng_counter) -- the replay then ends with the
        # parameters updated instead of leaving a host launch (and its ~15-20 us of idle GPU) behind it.  Data-parallel jobs keep
        # Adam outside: the all-reduce comes between.
        self._adam_in_graph = bool(self.ADAM_IN_GRAPH and self.world == 1 and not self.overlap)
        with torch.cuda.graph(g, **mode):
            self._static_lo = self._forward_backward(self._static_rows, self._klw)
            if self._adam_in_graph:
                self.opt.launch(grad_scale=1.0, images=self._img_plan, counter=getattr(self.module, "_rng_counter", None))
        if keep:
            g.instantiate()
        self.graph = g
        if self.overlap:
            g2 = torch.cuda.CUDAGraph(keep_graph=True) if keep else torch.cuda.CUDAGraph()
            with torch.cuda.graph(g2, pool=g.pool(), **mode):
                self._backward_encoders()
            if keep:
                g2.instantiate()
            self.graph2 = g2

    def step(self, rows: Sequence[torch.Tensor], kl_weight: Optional[float] = None, noise=None, optimizer_step: bool = True):
        """forward + loss + backward + (all-reduce) + Adam for one minibatch; returns the LossOutput.  ``noise`` (eager steps
        only) injects the step's standard-normal draws; ``optimizer_step=False`` stops before Adam, leaving the reduced
        gradients in ``self.fp.grad`` (parity tests)."""
        if noise is not None and self.graph is not None:
            raise ValueError("injected noise needs an eager step (the captured graph draws its own)")
        if kl_weight is None:
            kl_weight = kl_weight_at(self.epoch, self.global_step, self.n_epochs_kl_warmup, self.n_steps_kl_warmup)
        self._ensure_images()
        if self.graph is not None:
            if self.MULTI_GATHER and all(r.dtype == torch.int32 and r.is_cuda and r.is_contiguous() and r.numel() == s.numel() for s, r in zip(self._static_rows, rows)):
                _abi.gather_u32([(r, None, s) for s, r in zip(self._static_rows, rows)])   # one launch for all groups' row indices
            else:
                for s, r in zip(self._static_rows, rows):
                    s.copy_(r)
            if float(kl_weight) != self._klw_host:  # (changes once per epoch during the warm-up, then never)
                self._klw.fill_(float(kl_weight))
                self._klw_host = float(kl_weight)
            self.graph.replay()
            lo = self._static_lo  # tensors are overwritten by the next replay
        else:
            lo = self._forward_backward(rows, kl_weight, noise)
        timed = self.world > 1
        if timed:   # GPU-time bracket of what the collectives leave exposed after the backward pass (bench.py reports it)
            if getattr(self, "_ev_ar", None) is None:
                self._ev_ar = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        if self.overlap:
            # bucket 1 (decoders) is summed over the ranks while the encoder half of the backward pass runs
            split = self.fp.split
            w1 = dist.all_reduce(self.fp.grad[:split], op=dist.ReduceOp.SUM, async_op=True) if self.world > 1 else None
            if self.graph is not None:
                self.graph2.replay()
            else:
                self._backward_encoders()
            if self.world > 1:
                self._ev_ar[0].record()
                w2 = dist.all_reduce(self.fp.grad[split:], op=dist.ReduceOp.SUM, async_op=True)
                w1.wait()
                w2.wait()
                self._ev_ar[1].record()
        elif self.world > 1:
            self._ev_ar[0].record()
            dist.all_reduce(self.fp.grad, op=dist.ReduceOp.SUM)  # ONE collective for the whole flat buffer (the north_star form)
            self._ev_ar[1].record()
        if self.graph is not None and getattr(self, "_adam_in_graph", False):
            if not optimizer_step:
                raise ValueError("this trainer's captured graph contains the optimiser step: use an eager step for optimizer_step=False")
            self.opt.t += 1   # (the launch itself was part of the replay)
            self.global_step += 1
        elif optimizer_step:
            plan = getattr(self, "_img_plan", None)
            if plan is not None and not self._images_are_fresh():
                plan = None   # (somebody wrote to a parameter since the images were made: plain Adam, and the images are rebuilt next step)
                self.parameters_changed()
            self.opt.step(grad_scale=1.0 / self.world, images=plan, counter=getattr(self.module, "_rng_counter", None))
            self.global_step += 1
        return lo

    def last_allreduce_exposed_ms(self) -> float:
        """GPU time of the latest step between the end of the backward pass and the point where Adam may start (what the
        gradient all-reduce leaves exposed).  Synchronises; 0 on a single rank."""
        if self.world <= 1 or getattr(self, "_ev_ar", None) is None:
            return 0.0
        self._ev_ar[1].synchronize()
        return float(self._ev_ar[0].elapsed_time(self._ev_ar[1]))

    @torch.no_grad()
    def validation_metrics(self, sampler: MinibatchSampler):
        """Mean loss / reconstruction / KL over the held-out validation cells (module in eval mode, kl_weight 1, full batches
        in the reference's loader order: the validation loader is built like the training one, drop_last=True,
        data/_multi_datasplitter.py:81-98).  None when a group has no full validation batch.  In a data-parallel job every
        rank evaluates the same cells (validation rows are not sharded) on the same model (``fit`` averages the BatchNorm running
        statistics over the ranks first); the draws still differ per rank (rank-offset counters), which is why ``fit`` takes the
        early-stopping decision on the rank-averaged metric (``EarlyStopping``)."""
        B = sampler.batch_size
        nb = [len(v) // B for v in sampler.val_idx]
        if not nb or min(nb) == 0:
            return None
        was_training = self.module.training
        self.module.eval()
        val = [torch.as_tensor(v, dtype=torch.int32, device=self.device) for v in sampler.val_idx]
        acc = None
        for step in range(max(nb)):
            rows = [v[(step % n) * B:(step % n + 1) * B].contiguous() for v, n in zip(val, nb)]
            _, _, lo = self.module(self.minibatch(rows), loss_kwargs={"kl_weight": 1.0})
            rec = lo.reconstruction_loss_mean
            cur = torch.stack([lo.loss, rec, lo.loss - rec])
            acc = cur if acc is None else acc + cur
        self.module.train(was_training)
        tot, rec, kl = (float(v) / max(nb) for v in acc.cpu())
        return {"elbo_validation": rec + kl, "reconstruction_loss_validation": rec, "kl_local_validation": kl, "validation_loss": tot}

    def fit(self, sampler: MinibatchSampler, max_epochs: int, log_every: int = 0, use_graph: bool = True, early_stopping: bool = False,
            early_stopping_patience: int = 45, early_stopping_min_delta: float = 0.0, check_val_every_n_epoch: Optional[int] = None):
        """``max_epochs`` passes over the sampler.  The step is captured into a hipGraph at the first minibatch (the
        sampler yields fixed-size batches); per-step metrics are accumulated ON DEVICE and read back once per epoch, so
        logging costs one tiny launch per logged step and no host synchronisation.

        Validation / early stopping follow scvi-tools' TrainRunner defaults (what the reference's ``train`` forwards to,
        training_mixin.py:112-123): with held-out cells the validation metrics are evaluated every
        ``check_val_every_n_epoch`` epochs (every epoch when early stopping is on, else never unless asked), and
        ``early_stopping`` monitors ``elbo_validation`` (mode min, patience 45, min_delta 0)."""
        if check_val_every_n_epoch is None:
            check_val_every_n_epoch = 1 if early_stopping else 0
        if early_stopping and (not sampler.val_idx or min(len(v) // sampler.batch_size for v in sampler.val_idx) == 0):
            raise ValueError("early_stopping needs at least one full validation batch per group (train_size < 1 / validation_size > 0)")
        for k in ("elbo_validation", "reconstruction_loss_validation", "kl_local_validation", "validation_loss"):
            self.history.setdefault(k, [])
        stopper = EarlyStopping(early_stopping_patience, early_stopping_min_delta, self.world, self.device) if early_stopping else None
        self.module.train()
        for ep in range(max_epochs):
            self.epoch = ep
            acc = None  # device: [sum loss, sum reconstruction, sum kl, n]
            for rows in sampler.epoch():
                if use_graph and self.graph is None:
                    self.capture(rows)
                klw = kl_weight_at(self.epoch, self.global_step, self.n_epochs_kl_warmup, self.n_steps_kl_warmup)
                lo = self.step(rows, kl_weight=klw)
                if log_every and (self.global_step % log_every == 0):
                    loss = lo.loss.detach()
                    rec = lo.reconstruction_loss_mean if lo.reconstruction_loss_mean is not None else sum(v.mean() for v in lo.reconstruction_loss.values())
                    if klw > 1e-12:
                        kl = (loss - rec) / klw   # loss = rec + kl_weight * mean_b(sum of the KL terms)
                    else:
                        kl = sum(v.detach().mean() for v in lo.kl_local.values())
                    cur = torch.stack([loss, rec.detach(), kl, torch.ones_like(loss)])
                    acc = cur if acc is None else acc + cur
            if acc is not None:
                tot, rec, kl, n = (float(v) for v in acc.cpu())
                self.history["train_loss"].append(tot / n)
                self.history["reconstruction_loss_train"].append(rec / n)
                self.history["kl_local_train"].append(kl / n)
                self.history["elbo_train"].append((rec + kl) / n)
            if check_val_every_n_epoch and (ep + 1) % check_val_every_n_epoch == 0:
                sync_float_buffers(self.module, self.world)   # every rank validates the same model
                vm = self.validation_metrics(sampler)
                if vm is not None:   # (the same on every rank: it depends on the split sizes only)
                    for k, v in vm.items():
                        self.history[k].append(v)
                    if stopper is not None and stopper.should_stop(vm["elbo_validation"]):   # collective: all ranks leave together
                        self.stopped_epoch = ep
                        break
        sync_float_buffers(self.module, self.world)   # ranks end with one model, BatchNorm running statistics included
        self.module.eval()  # scvi's TrainRunner leaves the module in eval mode
        return self.history
