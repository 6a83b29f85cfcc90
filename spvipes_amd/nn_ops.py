"""Autograd wrappers over the small dense HIP primitives (csrc/spv_small.h): the [B, <=256] layers of the
step -- both encoders' tails of both groups, and the decoder trunk -- run as a handful of batched
launches with a hand-written backward instead of hundreds of eager torch kernels.

Reference arithmetic replaced (file:line into /root/reference/src/spVIPES):
    EncoderTails   nn/networks.py:120-129 (fc2+relu, dropout, mu / logvar heads with BatchNorm1d, rsample,
                   softmax) and the private KL of module/spVIPESmodule.py:841-854
    TrunkBN        nn/networks.py:322-323 (sigmoid_decoder: Linear -> BatchNorm1d(eps 1e-3, momentum 0.01) -> ReLU)
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence

import torch

from . import _abi
from ._abi import SpvBnBatch, SpvLinearBatch, SpvSampleBatch, ptr, stream_ptr


def _lin_batch(B: int, relu: bool = False, drop_p: float = 0.0, seed=0, accumulate: bool = False) -> SpvLinearBatch:
    """``seed``: python int, or a device int64 tensor (read by the kernel at run time: hipGraph-safe)."""
    b = SpvLinearBatch()
    b.nprob, b.B, b.relu, b.drop_p, b.accumulate = 0, B, int(relu), float(drop_p), int(accumulate)
    if isinstance(seed, torch.Tensor):
        b.seed, b.seed_ptr = 0, seed.data_ptr()
    else:
        b.seed, b.seed_ptr = int(seed), None
    return b


def _add_lin(b: SpvLinearBatch, *, N: int, K: int, W=None, X=None, ldx=0, bias=None, Y=None, ldy=0, dY=None, lddy=0, dX=None, lddx=0,
             dW=None, db=None, keep=None, W2=None, n_w2=0) -> None:
    q = b.p[b.nprob]
    q.X, q.ldx, q.W, q.bias, q.Y, q.ldy, q.keep, q.W2, q.n_w2 = X, ldx, W, bias, Y, ldy, keep, W2, n_w2
    q.dY, q.lddy, q.dX, q.lddx, q.dW, q.db, q.N, q.K = dY, lddy, dX, lddx, dW, db, N, K
    b.nprob += 1


# When True (set by train.Trainer around its step), parameter gradients of the small layers are written
# straight into the parameters' pre-allocated, zeroed ``.grad`` views of the flat gradient buffer and the
# autograd node returns None for them: every parameter receives exactly one contribution per step, so a
# store equals an accumulation, and ~60 tiny "grad += g" kernels per step disappear.
GRAD_SINK = False


def grad_out(param: torch.Tensor):
    """(tensor the kernel writes, value the autograd node returns) for one parameter's gradient."""
    g = param.grad if param.is_leaf else None   # (a derived tensor -- e.g. module.CovariateColumns' output -- has no gradient buffer of its own)
    if GRAD_SINK and g is not None and g.is_contiguous() and g.dtype == torch.float32:
        return g, None
    t = torch.empty_like(param, dtype=torch.float32)
    return t, t


def _wgrad(b: SpvLinearBatch, ws) -> None:
    """spv_linear_wgrad with its slice-partial workspace (16 batch slices, summed in fixed order).
    (Measured: moving these parameter-gradient launches to the side stream of the mixture-weight GEMMs makes the step
    slower -- they are too small to be worth the cross-stream edges.)"""
    nmax = max(b.p[i].N for i in range(b.nprob))
    kmax = max(b.p[i].K for i in range(b.nprob))
    n = b.nprob * 16 * nmax * (kmax + 1)
    part = ws.get("wgrad_part", (n,), torch.float32)
    _abi.call("spv_linear_wgrad", C.byref(b), ptr(part), n, stream_ptr())


def _fptr(t: torch.Tensor, col: int = 0) -> int:
    """device address of t[0, col] for a row-major fp32 matrix (or of t[col] for a vector)."""
    return t.data_ptr() + 4 * col


class EncoderSpec:
    """Parameters and buffers of one reference ``Encoder`` (nn/networks.py:47-83) after fc1."""

    def __init__(self, enc, h1_group: int, h1_col: int):
        self.enc, self.h1_group, self.h1_col = enc, h1_group, h1_col
        self.n = enc.n_topics

    def params(self) -> List[torch.Tensor]:
        e = self.enc
        return [e.fc2.weight, e.fc2.bias, e.mu_encoder[0].weight, e.mu_encoder[0].bias, e.lvar_encoder[0].weight, e.lvar_encoder[0].bias,
                e.mu_encoder[1].weight, e.mu_encoder[1].bias, e.lvar_encoder[1].weight, e.lvar_encoder[1].bias]


N_ENC_PARAMS = 10


class EncoderTails(torch.autograd.Function):
    """All encoders' tails in 4 forward / 8 backward launches (fc2, heads, BatchNorm statistics, [finalise + normalise + sample + KL];
    backward: [sampling + BatchNorm partial sums], [finalise + apply], heads wgrad + reduce, 2 x heads dgrad, fc2 wgrad + reduce, fc2 dgrad).

    inputs : h1 per group ([B, 2H]: private | shared halves), then per encoder its 10 parameters
    outputs: per encoder (loc, logvar, scale, log_z, theta, kl)  -- theta is not differentiable here
    """

    @staticmethod
    def forward(ctx, specs: Sequence[EncoderSpec], eps: Sequence[torch.Tensor], training: bool, drop_p: float, seed: int, ws, masks, *tensors):
        """``masks``: None, or one optional [B, H] keep-mask (1 = keep) per encoder replacing the counter-based dropout draw."""
        ctx.set_materialize_grads(False)  # outputs nobody differentiates (scale, unused loc/logvar) arrive as None, not as zero fills
        n_groups = max(s.h1_group for s in specs) + 1
        h1 = tensors[:n_groups]
        par = [tensors[n_groups + i * N_ENC_PARAMS: n_groups + (i + 1) * N_ENC_PARAMS] for i in range(len(specs))]
        B, H = h1[0].shape[0], specs[0].enc.fc2.weight.shape[0]
        dev = h1[0].device
        for t in h1:
            if t.dtype != torch.float32 or not t.is_contiguous():
                raise _abi.SpvError("EncoderTails expects contiguous fp32 h1")
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        h2 = [new(B, H) for _ in specs]
        pre = [new(B, 2 * s.n) for s in specs]
        post = [new(B, 2 * s.n) for s in specs]
        stats = [new(2, s.n, 2) for s in specs]
        nblk = -(-B // _abi.BN_ROWS)
        dp = float(drop_p) if training else 0.0
        # 1. fc2 + relu (+ dropout)
        b = _lin_batch(B, relu=True, drop_p=dp, seed=seed)
        keep_masks = []
        for i, (s, p, y) in enumerate(zip(specs, par, h2)):
            mk = None
            if masks is not None and masks[i] is not None and dp > 0:
                mk = masks[i].to(dev, torch.float32).contiguous()
                if mk.shape != (B, H):
                    raise _abi.SpvError("dropout mask must be [B, n_hidden]")
                keep_masks.append(mk)
            _add_lin(b, N=H, K=H, W=ptr(p[0]), bias=ptr(p[1]), X=_fptr(h1[s.h1_group], s.h1_col), ldx=h1[s.h1_group].shape[1], Y=ptr(y), ldy=H, keep=ptr(mk))
        _abi.call("spv_linear_fwd", C.byref(b), stream_ptr())
        # 2. mu / logvar heads into the two halves of `pre`
        b = _lin_batch(B)
        for s, p, x, y in zip(specs, par, h2, pre):
            _add_lin(b, N=s.n, K=H, W=ptr(p[2]), bias=ptr(p[3]), X=ptr(x), ldx=H, Y=_fptr(y, 0), ldy=2 * s.n)
            _add_lin(b, N=s.n, K=H, W=ptr(p[4]), bias=ptr(p[5]), X=ptr(x), ldx=H, Y=_fptr(y, s.n), ldy=2 * s.n)
        _abi.call("spv_linear_fwd", C.byref(b), stream_ptr())
        # 3. BatchNorm1d (torch defaults eps 1e-5, momentum 0.1: nn/networks.py:76,82)
        bn = SpvBnBatch()
        bn.nprob, bn.B, bn.training, bn.relu, bn.eps, bn.momentum = 0, B, int(training), 0, 1e-5, 0.1
        parts = []
        for s, p, x, y, st in zip(specs, par, pre, post, stats):
            for half, (g, be, mod) in enumerate(((p[6], p[7], s.enc.mu_encoder[1]), (p[8], p[9], s.enc.lvar_encoder[1]))):
                part = ws.get(f"tail_bn_part_{len(parts)}", (nblk, s.n, 2), torch.float32)
                parts.append(part)
                q = bn.p[bn.nprob]
                q.X, q.ldx, q.Y, q.ldy = _fptr(x, half * s.n), 2 * s.n, _fptr(y, half * s.n), 2 * s.n
                q.gamma, q.beta, q.running_mean, q.running_var = ptr(g), ptr(be), ptr(mod.running_mean), ptr(mod.running_var)
                q.stats, q.part, q.N = ptr(st[half]), ptr(part), s.n
                bn.nprob += 1
        # 4. scale, reparameterised draw, softmax, KL -- in the same launch as the BatchNorm apply (spv_enc_heads_fwd: batch statistics,
        #    then ONE kernel that finalises them, normalises, samples and forms the KL)
        scale = [new(B, s.n) for s in specs]
        logz = [new(B, s.n) for s in specs]
        theta = [new(B, s.n) for s in specs]
        kl = [new(B) for _ in specs]
        sb = SpvSampleBatch()
        sb.nprob, sb.B = len(specs), B
        for i, s in enumerate(specs):
            q = sb.p[i]
            q.post, q.n, q.eps, q.scale, q.logz, q.theta, q.kl = ptr(post[i]), s.n, ptr(eps[i].contiguous()), ptr(scale[i]), ptr(logz[i]), ptr(theta[i]), ptr(kl[i])
        _abi.call("spv_enc_heads_fwd", C.byref(bn), C.byref(sb), stream_ptr())
        ctx.specs, ctx.training, ctx.dp, ctx.seed, ctx.ws, ctx.n_groups, ctx.B, ctx.H = specs, training, dp, seed, ws, n_groups, B, H
        ctx.eps = [e.contiguous() for e in eps]
        ctx.save_for_backward(*tensors, *h2, *pre, *post, *stats, *scale)
        outs = []
        for i, s in enumerate(specs):
            outs += [post[i][:, :s.n], post[i][:, s.n:], scale[i], logz[i], theta[i], kl[i]]
            ctx.mark_non_differentiable(theta[i])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g):
        specs, B, H, ws, n_groups = ctx.specs, ctx.B, ctx.H, ctx.ws, ctx.n_groups
        E = len(specs)
        saved = ctx.saved_tensors
        n_in = n_groups + E * N_ENC_PARAMS
        tensors = saved[:n_in]
        h1 = tensors[:n_groups]
        par = [tensors[n_groups + i * N_ENC_PARAMS: n_groups + (i + 1) * N_ENC_PARAMS] for i in range(E)]
        h2, pre, post, stats, scale = (saved[n_in + k * E: n_in + (k + 1) * E] for k in range(5))
        dev = h1[0].device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        cont = lambda t: None if t is None else t.contiguous()
        # 1. sampling / KL backward -> d_post
        d_post = [new(B, 2 * s.n) for s in specs]
        sb = SpvSampleBatch()
        sb.nprob, sb.B = E, B
        keep = []
        for i, s in enumerate(specs):
            gl, gv, gs, gz, _gt, gk = g[6 * i: 6 * i + 6]
            # g_loc / g_logvar usually arrive as the two column halves of one [B][2n] buffer (PoELabel.backward): read them in place
            g_ld = 0
            strided = [t for t in (gl, gv) if t is not None and not t.is_contiguous()]
            if strided and all(t is None or (t.dtype == torch.float32 and t.stride(1) == 1 and t.stride(0) == strided[0].stride(0)) for t in (gl, gv)):
                g_ld = strided[0].stride(0)
            else:
                gl, gv = cont(gl), cont(gv)
            gs, gz, gk = cont(gs), cont(gz), cont(gk)
            keep += [gl, gv, gs, gz, gk]
            q = sb.p[i]
            q.post, q.n, q.eps, q.scale = ptr(post[i]), s.n, ptr(ctx.eps[i]), ptr(scale[i])
            q.g_loc, q.g_logvar, q.g_scale, q.g_logz, q.g_kl, q.d_post, q.g_ld = ptr(gl), ptr(gv), ptr(gs), ptr(gz), ptr(gk), ptr(d_post[i]), g_ld
        fused = bool(ctx.training)   # spv_enc_heads_bwd: sampling backward + BatchNorm partial sums in one kernel, finalise + apply in a second
        if not fused:
            _abi.call("spv_enc_sample_bwd", C.byref(sb), stream_ptr())
        # 2. BatchNorm backward -> d_pre, d gamma / beta
        d_pre = [new(B, 2 * s.n) for s in specs]
        # parameter gradients: (write target, autograd return value) per parameter of each encoder
        pg = [[grad_out(par[i][j]) for j in range(N_ENC_PARAMS)] for i in range(E)]
        d_gb = [[pg[i][6][0], pg[i][7][0], pg[i][8][0], pg[i][9][0]] for i in range(E)]  # d gamma_mu, d beta_mu, d gamma_lv, d beta_lv
        nblk = -(-B // _abi.BN_ROWS)
        bn = SpvBnBatch()
        bn.nprob, bn.B, bn.training, bn.relu, bn.eps, bn.momentum = 0, B, int(ctx.training), 0, 1e-5, 0.1
        k = 0
        for i, (s, p) in enumerate(zip(specs, par)):
            for half, (gam, be, mod) in enumerate(((p[6], p[7], s.enc.mu_encoder[1]), (p[8], p[9], s.enc.lvar_encoder[1]))):
                q = bn.p[bn.nprob]
                q.X, q.ldx, q.Y, q.ldy = _fptr(pre[i], half * s.n), 2 * s.n, _fptr(post[i], half * s.n), 2 * s.n
                q.gamma, q.beta, q.running_mean, q.running_var = ptr(gam), ptr(be), ptr(mod.running_mean), ptr(mod.running_var)
                q.stats, q.part, q.N = ptr(stats[i][half]), ptr(ws.get(f"tail_bn_part_{k}", (nblk, s.n, 2), torch.float32)), s.n
                q.dY, q.lddy, q.dX, q.lddx = _fptr(d_post[i], half * s.n), 2 * s.n, _fptr(d_pre[i], half * s.n), 2 * s.n
                q.dgamma, q.dbeta = ptr(d_gb[i][2 * half]), ptr(d_gb[i][2 * half + 1])
                bn.nprob += 1
                k += 1
        if fused:
            _abi.call("spv_enc_heads_bwd", C.byref(bn), C.byref(sb), stream_ptr())
        else:
            _abi.call("spv_bn_bwd", C.byref(bn), stream_ptr())
        # 3. head weight gradients
        dWmu, dbmu, dWlv, dblv = ([pg[i][2][0] for i in range(E)], [pg[i][3][0] for i in range(E)], [pg[i][4][0] for i in range(E)], [pg[i][5][0] for i in range(E)])
        b = _lin_batch(B)
        for i, s in enumerate(specs):
            _add_lin(b, N=s.n, K=H, W=ptr(par[i][2]), X=ptr(h2[i]), ldx=H, dY=_fptr(d_pre[i], 0), lddy=2 * s.n, dW=ptr(dWmu[i]), db=ptr(dbmu[i]))
            _add_lin(b, N=s.n, K=H, W=ptr(par[i][4]), X=ptr(h2[i]), ldx=H, dY=_fptr(d_pre[i], s.n), lddy=2 * s.n, dW=ptr(dWlv[i]), db=ptr(dblv[i]))
        from . import ops as _ops
        wsw = ws if not isinstance(ws, (list, tuple)) else ws[0]
        _wgrad(b, wsw)
        # 4. d h2 = d_pre_mu Wmu + d_pre_lv Wlv
        #    one launch: the contraction runs over the 2n columns of d_pre, rows 0..n-1 of the weight from Wmu, rows n.. from Wlv
        dh2 = [new(B, H) for _ in specs]
        b = _lin_batch(B)
        for i, s in enumerate(specs):
            _add_lin(b, N=2 * s.n, K=H, W=ptr(par[i][2]), W2=ptr(par[i][4]), n_w2=s.n, dY=ptr(d_pre[i]), lddy=2 * s.n, dX=ptr(dh2[i]), lddx=H)
        _abi.call("spv_linear_dgrad", C.byref(b), stream_ptr())
        # 5./6. fc2 backward (relu + dropout mask recovered from the saved h2 > 0)
        dW2, db2 = [pg[i][0][0] for i in range(E)], [pg[i][1][0] for i in range(E)]
        # every column block of h1 that an encoder reads gets written by the dgrad below; zero only what none covers
        covered = [sorted((s.h1_col, s.h1_col + H) for s in specs if s.h1_group == k) for k in range(n_groups)]
        full = [bool(c) and c[0][0] == 0 and c[-1][1] == h1[k].shape[1] and all(c[i][1] == c[i + 1][0] for i in range(len(c) - 1))
                for k, c in enumerate(covered)]
        dh1 = [torch.empty_like(t) if full[k] else torch.zeros_like(t) for k, t in enumerate(h1)]
        bw = _lin_batch(B, relu=True, drop_p=ctx.dp)
        bd = _lin_batch(B, relu=True, drop_p=ctx.dp)
        for i, s in enumerate(specs):
            x, ldx = _fptr(h1[s.h1_group], s.h1_col), h1[s.h1_group].shape[1]
            _add_lin(bw, N=H, K=H, W=ptr(par[i][0]), X=x, ldx=ldx, Y=ptr(h2[i]), ldy=H, dY=ptr(dh2[i]), lddy=H, dW=ptr(dW2[i]), db=ptr(db2[i]))
            _add_lin(bd, N=H, K=H, W=ptr(par[i][0]), Y=ptr(h2[i]), ldy=H, dY=ptr(dh2[i]), lddy=H, dX=_fptr(dh1[s.h1_group], s.h1_col), lddx=ldx)
        _wgrad(bw, wsw)
        _abi.call("spv_linear_dgrad", C.byref(bd), stream_ptr())
        grads: List[Optional[torch.Tensor]] = list(dh1)
        for i in range(E):
            grads += [pg[i][j][1] for j in range(N_ENC_PARAMS)]
        return (None, None, None, None, None, None, None, *grads)


# ------------------------------------------------------------------------------------------------
# label-based Product of Experts
# ------------------------------------------------------------------------------------------------
def _loc_logvar_block(loc: torch.Tensor, logvar: torch.Tensor):
    """(base pointer, row pitch) of a [B][>=2n] block holding loc | logvar side by side; copies if the two
    are not already adjacent views of one buffer (they are when they come out of EncoderTails)."""
    n = loc.shape[1]
    if (loc.dtype == torch.float32 and loc.stride(1) == 1 and logvar.stride(1) == 1 and loc.stride(0) == logvar.stride(0)
            and logvar.data_ptr() == loc.data_ptr() + 4 * n and loc.stride(0) >= 2 * n):
        return loc, loc.data_ptr(), loc.stride(0)
    blk = torch.cat([loc, logvar], dim=1).contiguous()
    return blk, blk.data_ptr(), 2 * n


def label_partners(labels: Sequence[torch.Tensor], ws):
    """Ranking of both minibatches' cells within their labels (spv_poe_rank: no dependence on the encoders, so ``module.inference``
    may launch it beside the fc1 GEMMs) and the buffers the fusion kernel then fills / reads: returns
    (partner, mode, lab, order, rank, tables); partner / mode are written by spv_poe_fuse_fwd, which does the per-cell lookup itself."""
    dev = labels[0].device
    lab = [l.flatten().contiguous().float() for l in labels]
    Bs = [lab[0].numel(), lab[1].numel()]
    i32 = lambda name, k: ws.get(name, (k,), torch.int32)
    order = [i32(f"poe_order{g}", Bs[g]) for g in range(2)]
    rank = [i32(f"poe_rank{g}", Bs[g]) for g in range(2)]
    partner = [torch.empty(Bs[g], dtype=torch.int32, device=dev) for g in range(2)]
    mode = [torch.empty(Bs[g], dtype=torch.int32, device=dev) for g in range(2)]
    err = ws.get("poe_err", (1,), torch.int32, zero=True)
    tables = ws.get("poe_tables", (2, 2, 1024), torch.int32)
    _abi.call("spv_poe_rank", ptr(lab[0]), ptr(lab[1]), Bs[0], Bs[1], ptr(order[0]), ptr(order[1]), ptr(rank[0]), ptr(rank[1]),
              ptr(tables), ptr(err), stream_ptr())
    return partner, mode, lab, order, rank, tables


class PoELabel(torch.autograd.Function):
    """Both groups' label-based PoE in 2 forward launches (ranking; lookup + fusion / draw / KL) and 2 backward launches (fill, scatter).
    inputs : loc0, logvar0, loc1, logvar1 (shared-encoder statistics); outputs per group
    (loc*, logvar*, scale*, log_z, theta, kl) -- theta not differentiable.  ``pre``: the result of ``label_partners`` when
    the caller has already launched the pairing."""

    @staticmethod
    def forward(ctx, labels: Sequence[torch.Tensor], eps: Sequence[torch.Tensor], ws, pre, loc0, logvar0, loc1, logvar1):
        from ._abi import SpvPoeArgs
        ctx.set_materialize_grads(False)
        dev = loc0.device
        n = loc0.shape[1]
        Bs = [loc0.shape[0], loc1.shape[0]]
        partner, mode, lab, order, rank, tables = pre if pre is not None else label_partners(labels, ws)
        blocks = [_loc_logvar_block(loc0, logvar0), _loc_logvar_block(loc1, logvar1)]
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        out = {k: [new(Bs[g], n) for g in range(2)] for k in ("loc", "logvar", "scale", "logz", "theta")}
        kl = [new(Bs[g]) for g in range(2)]
        eps = [e.contiguous() for e in eps]
        a = SpvPoeArgs()
        a.n = n
        for g in range(2):
            a.stats[g], a.ld[g], a.partner[g], a.mode[g], a.eps[g], a.B[g] = blocks[g][1], blocks[g][2], ptr(partner[g]), ptr(mode[g]), ptr(eps[g]), Bs[g]
            a.loc[g], a.logvar[g], a.scale[g], a.logz[g], a.theta[g] = (ptr(out[k][g]) for k in ("loc", "logvar", "scale", "logz", "theta"))
            a.kl[g] = ptr(kl[g])
            a.lab[g], a.order[g], a.rank[g] = ptr(lab[g]), ptr(order[g]), ptr(rank[g])   # the fusion kernel looks the partners up itself
        a.tables = ptr(tables)
        _abi.call("spv_poe_fuse_fwd", C.byref(a), stream_ptr())
        ctx.blocks, ctx.eps, ctx.partner, ctx.mode, ctx.n, ctx.Bs = blocks, eps, partner, mode, n, Bs
        ctx.lab = lab
        ctx.save_for_backward(out["loc"][0], out["loc"][1], out["scale"][0], out["scale"][1])
        res = []
        for g in range(2):
            res += [out["loc"][g], out["logvar"][g], out["scale"][g], out["logz"][g], out["theta"][g], kl[g]]
            ctx.mark_non_differentiable(out["theta"][g])
        return tuple(res)

    @staticmethod
    def backward(ctx, *g):
        from ._abi import SpvPoeArgs
        loc = ctx.saved_tensors[0:2]
        scale = ctx.saved_tensors[2:4]
        n, Bs = ctx.n, ctx.Bs
        dev = loc[0].device
        cont = lambda t: None if t is None else t.contiguous()
        a = SpvPoeArgs()
        a.n = n
        # both groups' gradient blocks in ONE allocation: spv_poe_fuse_bwd then zeroes them with a single memset
        lds = [ctx.blocks[k][2] for k in range(2)]
        dbuf = torch.empty(Bs[0] * lds[0] + Bs[1] * lds[1], dtype=torch.float32, device=dev)
        d = [dbuf[:Bs[0] * lds[0]].view(Bs[0], lds[0]), dbuf[Bs[0] * lds[0]:].view(Bs[1], lds[1])]
        keep = []
        for k in range(2):
            gl, gv, gs, gz, _gt, gk = (cont(t) for t in g[6 * k: 6 * k + 6])
            keep += [gl, gv, gs, gz, gk]
            a.stats[k], a.ld[k], a.partner[k], a.mode[k], a.eps[k], a.B[k] = ctx.blocks[k][1], ctx.blocks[k][2], ptr(ctx.partner[k]), ptr(ctx.mode[k]), ptr(ctx.eps[k]), Bs[k]
            a.loc[k], a.scale[k] = ptr(loc[k]), ptr(scale[k])
            a.g_loc[k], a.g_logvar[k], a.g_scale[k], a.g_logz[k], a.g_kl[k] = ptr(gl), ptr(gv), ptr(gs), ptr(gz), ptr(gk)
            a.lab[k] = ptr(ctx.lab[k])   # (a flag for the backward kernel: label pairs are one to one)
        for k in range(2):  # (d_stats uses the same pitch as stats, which may be views of a wider buffer)
            a.d_stats[k] = ptr(d[k])
        _abi.call("spv_poe_fuse_bwd", C.byref(a), stream_ptr())
        return (None, None, None, None, d[0][:, :n], d[0][:, n:2 * n], d[1][:, :n], d[1][:, n:2 * n])


class PoEPaired(torch.autograd.Function):
    """Paired PoE (module/spVIPESmodule.py:511-571) on a sparse transport plan: every cell is fused with the arg max of its
    plan row (group 0) / column (group 1) inside the minibatch pair.  2 + 1 forward launches, 1 backward launch.
    inputs : loc0, logvar0, loc1, logvar1; outputs per group (loc*, logvar*, scale*, log_z, theta, kl, qscale)
    -- theta and qscale (= scale*.clamp(min = 1e-6), the scale of the reference's Normal) are not differentiable."""

    @staticmethod
    def forward(ctx, plan, idx: Sequence[torch.Tensor], eps: Sequence[torch.Tensor], ws, loc0, logvar0, loc1, logvar1):
        from ._abi import SpvPoeArgs
        ctx.set_materialize_grads(False)
        dev = loc0.device
        n = loc0.shape[1]
        Bs = [loc0.shape[0], loc1.shape[0]]
        if Bs[0] != Bs[1]:
            raise AssertionError("Paired PoE requires equal number of cells from both groups")
        idx = [i.flatten().to(torch.int32).contiguous() for i in idx]
        plan.bind_minibatch(idx[0], idx[1])
        partner = [torch.empty(Bs[g], dtype=torch.int32, device=dev) for g in range(2)]
        mode = [ws.get(f"poe_mode0_{g}", (Bs[g],), torch.int32, zero=True) for g in range(2)]  # every cell has a partner
        ps = plan.c_struct()
        _abi.call("spv_plan_argmax", C.byref(ps), ptr(idx[0]), ptr(idx[1]), ptr(plan.inv0), ptr(plan.inv1), Bs[0], Bs[1], ptr(partner[0]),
                  ptr(partner[1]), stream_ptr())
        blocks = [_loc_logvar_block(loc0, logvar0), _loc_logvar_block(loc1, logvar1)]
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        out = {k: [new(Bs[g], n) for g in range(2)] for k in ("loc", "logvar", "scale", "logz", "theta")}
        kl = [new(Bs[g]) for g in range(2)]
        eps = [e.contiguous() for e in eps]
        a = SpvPoeArgs()
        a.n, a.clamp_scale, a.lone_passthrough = n, 1, 0
        for g in range(2):
            a.stats[g], a.ld[g], a.partner[g], a.mode[g], a.eps[g], a.B[g] = blocks[g][1], blocks[g][2], ptr(partner[g]), ptr(mode[g]), ptr(eps[g]), Bs[g]
            a.loc[g], a.logvar[g], a.scale[g], a.logz[g], a.theta[g] = (ptr(out[k][g]) for k in ("loc", "logvar", "scale", "logz", "theta"))
            a.kl[g] = ptr(kl[g])
        _abi.call("spv_poe_fuse_fwd", C.byref(a), stream_ptr())
        ctx.blocks, ctx.eps, ctx.partner, ctx.mode, ctx.n, ctx.Bs = blocks, eps, partner, mode, n, Bs
        ctx.save_for_backward(out["loc"][0], out["loc"][1], out["scale"][0], out["scale"][1])
        res = []
        for g in range(2):
            qscale = out["scale"][g].clamp(min=1e-6)
            res += [out["loc"][g], out["logvar"][g], out["scale"][g], out["logz"][g], out["theta"][g], kl[g], qscale]
            ctx.mark_non_differentiable(out["theta"][g], qscale)
        return tuple(res)

    @staticmethod
    def backward(ctx, *g):
        from ._abi import SpvPoeArgs
        loc, scale = ctx.saved_tensors[0:2], ctx.saved_tensors[2:4]
        n, Bs = ctx.n, ctx.Bs
        dev = loc[0].device
        cont = lambda t: None if t is None else t.contiguous()
        a = SpvPoeArgs()
        a.n, a.clamp_scale, a.lone_passthrough = n, 1, 0
        keep, d = [], []
        for k in range(2):
            gl, gv, gs, gz, _gt, gk, _gq = (cont(t) for t in g[7 * k: 7 * k + 7])
            keep += [gl, gv, gs, gz, gk]
            a.stats[k], a.ld[k], a.partner[k], a.mode[k], a.eps[k], a.B[k] = ctx.blocks[k][1], ctx.blocks[k][2], ptr(ctx.partner[k]), ptr(ctx.mode[k]), ptr(ctx.eps[k]), Bs[k]
            a.loc[k], a.scale[k] = ptr(loc[k]), ptr(scale[k])
            a.g_loc[k], a.g_logvar[k], a.g_scale[k], a.g_logz[k], a.g_kl[k] = ptr(gl), ptr(gv), ptr(gs), ptr(gz), ptr(gk)
            d.append(torch.empty(Bs[k], ctx.blocks[k][2], dtype=torch.float32, device=dev))  # zeroed by spv_poe_fuse_bwd
            a.d_stats[k] = ptr(d[k])
        _abi.call("spv_poe_fuse_bwd", C.byref(a), stream_ptr())
        return (None, None, None, None, d[0][:, :n], d[0][:, n:2 * n], d[1][:, :n], d[1][:, n:2 * n])


class PoECluster(torch.autograd.Function):
    """Cluster-matched PoE (module/spVIPESmodule.py:184-280) on a sparse transport plan: plan-weighted experts inside
    each component (spv_plan_expert_fwd), rank-within-component pairing (spv_poe_partner on the component codes) and the
    _poe2 fusion with its padding rule; a component present in one minibatch only keeps its encoder statistics.
    inputs : loc0, logvar0, loc1, logvar1; outputs per group (loc*, logvar*, scale*, log_z, theta, kl, qscale)."""

    @staticmethod
    def forward(ctx, plan, idx: Sequence[torch.Tensor], comps: Sequence[torch.Tensor], eps: Sequence[torch.Tensor], ws, loc0, logvar0, loc1, logvar1):
        from ._abi import SpvPlanExpertArgs, SpvPoeArgs
        ctx.set_materialize_grads(False)
        dev = loc0.device
        n = loc0.shape[1]
        Bs = [loc0.shape[0], loc1.shape[0]]
        if Bs[0] != Bs[1]:
            raise RuntimeError("cluster-based PoE needs equally sized minibatches (the reference indexes one group's statistics with the other's mask)")
        B = Bs[0]
        idx = [i.flatten().to(torch.int32).contiguous() for i in idx]
        comp = [c.flatten().contiguous().float() for c in comps]
        plan.bind_minibatch(idx[0], idx[1])
        i32 = lambda name, k: ws.get(name, (k,), torch.int32)
        order = [i32(f"poe_order{g}", B) for g in range(2)]
        rank = [i32(f"poe_rank{g}", B) for g in range(2)]
        partner = [torch.empty(B, dtype=torch.int32, device=dev) for g in range(2)]
        mode = [torch.empty(B, dtype=torch.int32, device=dev) for g in range(2)]
        err = ws.get("poe_err", (1,), torch.int32, zero=True)
        tables = ws.get("poe_tables", (2, 2, 1024), torch.int32)
        _abi.call("spv_poe_partner", ptr(comp[0]), ptr(comp[1]), B, B, ptr(order[0]), ptr(order[1]), ptr(rank[0]), ptr(rank[1]),
                  ptr(tables), ptr(partner[0]), ptr(mode[0]), ptr(partner[1]), ptr(mode[1]), ptr(err), stream_ptr())
        blocks = [_loc_logvar_block(loc0, logvar0), _loc_logvar_block(loc1, logvar1)]
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        expert = [new(B, 2 * n) for _ in range(2)]
        rowsum = [new(B) for _ in range(2)]
        ea = SpvPlanExpertArgs()
        ea.plan, ea.B, ea.n = plan.c_struct(), B, n
        inv = [plan.inv0, plan.inv1]
        for g in range(2):
            ea.idx[g], ea.inv[g], ea.comp[g], ea.stats[g], ea.ld[g] = ptr(idx[g]), ptr(inv[g]), ptr(comp[g]), blocks[g][1], blocks[g][2]
            ea.expert[g], ea.ld_expert[g], ea.rowsum[g] = ptr(expert[g]), 2 * n, ptr(rowsum[g])
        _abi.call("spv_plan_expert_fwd", C.byref(ea), stream_ptr())
        out = {k: [new(B, n) for g in range(2)] for k in ("loc", "logvar", "scale", "logz", "theta")}
        kl = [new(B) for g in range(2)]
        eps = [e.contiguous() for e in eps]
        a = SpvPoeArgs()
        a.n, a.clamp_scale, a.lone_passthrough = n, 1, 1
        for g in range(2):
            a.stats[g], a.ld[g], a.partner[g], a.mode[g], a.eps[g], a.B[g] = blocks[g][1], blocks[g][2], ptr(partner[g]), ptr(mode[g]), ptr(eps[g]), B
            a.expert[g], a.ld_expert[g] = ptr(expert[g]), 2 * n
            a.loc[g], a.logvar[g], a.scale[g], a.logz[g], a.theta[g] = (ptr(out[k][g]) for k in ("loc", "logvar", "scale", "logz", "theta"))
            a.kl[g] = ptr(kl[g])
        _abi.call("spv_poe_fuse_fwd", C.byref(a), stream_ptr())
        # the inverse maps of the plan are rebuilt by the next minibatch: keep what the backward needs of them
        # (copied by a kernel: a tensor .clone() would put a memcpy node into the captured step, DESIGN.md §4 "memset node")
        kept = [torch.empty_like(plan.inv0), torch.empty_like(plan.inv1)]
        _abi.gather_u32([(plan.inv0, None, kept[0]), (plan.inv1, None, kept[1])])
        ctx.plan, ctx.idx, ctx.comp, ctx.inv = plan, idx, comp, kept
        ctx.blocks, ctx.eps, ctx.partner, ctx.mode, ctx.n, ctx.B = blocks, eps, partner, mode, n, B
        ctx.expert, ctx.rowsum = expert, rowsum
        ctx.save_for_backward(out["loc"][0], out["loc"][1], out["scale"][0], out["scale"][1])
        res = []
        for g in range(2):
            qscale = out["scale"][g].clamp(min=1e-6)
            res += [out["loc"][g], out["logvar"][g], out["scale"][g], out["logz"][g], out["theta"][g], kl[g], qscale]
            ctx.mark_non_differentiable(out["theta"][g], qscale)
        return tuple(res)

    @staticmethod
    def backward(ctx, *g):
        from ._abi import SpvPlanExpertArgs, SpvPoeArgs
        loc, scale = ctx.saved_tensors[0:2], ctx.saved_tensors[2:4]
        n, B = ctx.n, ctx.B
        dev = loc[0].device
        cont = lambda t: None if t is None else t.contiguous()
        a = SpvPoeArgs()
        a.n, a.clamp_scale, a.lone_passthrough = n, 1, 1
        keep, d, de = [], [], []
        for k in range(2):
            gl, gv, gs, gz, _gt, gk, _gq = (cont(t) for t in g[7 * k: 7 * k + 7])
            keep += [gl, gv, gs, gz, gk]
            a.stats[k], a.ld[k], a.partner[k], a.mode[k], a.eps[k], a.B[k] = ctx.blocks[k][1], ctx.blocks[k][2], ptr(ctx.partner[k]), ptr(ctx.mode[k]), ptr(ctx.eps[k]), B
            a.expert[k], a.ld_expert[k] = ptr(ctx.expert[k]), 2 * n
            a.loc[k], a.scale[k] = ptr(loc[k]), ptr(scale[k])
            a.g_loc[k], a.g_logvar[k], a.g_scale[k], a.g_logz[k], a.g_kl[k] = ptr(gl), ptr(gv), ptr(gs), ptr(gz), ptr(gk)
            d.append(torch.empty(B, ctx.blocks[k][2], dtype=torch.float32, device=dev))   # both zeroed by spv_poe_fuse_bwd
            de.append(torch.empty(B, 2 * n, dtype=torch.float32, device=dev))
            a.d_stats[k], a.d_expert[k] = ptr(d[k]), ptr(de[k])
        _abi.call("spv_poe_fuse_bwd", C.byref(a), stream_ptr())
        ea = SpvPlanExpertArgs()
        ea.plan, ea.B, ea.n = ctx.plan.c_struct(), B, n
        for k in range(2):
            ea.idx[k], ea.inv[k], ea.comp[k], ea.stats[k], ea.ld[k] = ptr(ctx.idx[k]), ptr(ctx.inv[k]), ptr(ctx.comp[k]), ctx.blocks[k][1], ctx.blocks[k][2]
            ea.expert[k], ea.ld_expert[k], ea.rowsum[k] = ptr(ctx.expert[k]), 2 * n, ptr(ctx.rowsum[k])
            ea.d_expert[k], ea.d_stats[k] = ptr(de[k]), ptr(d[k])
        _abi.call("spv_plan_expert_bwd", C.byref(ea), stream_ptr())
        return (None, None, None, None, None, d[0][:, :n], d[0][:, n:2 * n], d[1][:, :n], d[1][:, n:2 * n])



class PoEComponents(torch.autograd.Function):
    """N-group cluster-matched PoE (csrc/spv_poe_n.h; BASELINE config 4, throughput only: the reference stops at two groups).
    inputs : per group (loc_g, logvar_g) of the shared encoders; ``comp``: per-group component codes [B] (fp32, integral in
    [0, n_comp)); outputs per group (loc*, logvar*, scale*, log_z, theta, kl, qscale) -- theta and qscale not differentiable."""

    @staticmethod
    def forward(ctx, comp: Sequence[torch.Tensor], n_comp: int, eps: Sequence[torch.Tensor], ws, *stats):
        from ._abi import POE_COMP_CMAX, POE_COMP_SEG, SPV_POE_MAXG, SpvPoeCompArgs
        ctx.set_materialize_grads(False)
        NG = len(comp)
        if not (2 <= NG <= SPV_POE_MAXG) or len(stats) != 2 * NG:
            raise _abi.SpvError(f"PoEComponents supports 2..{SPV_POE_MAXG} groups")
        if n_comp > POE_COMP_CMAX:
            raise _abi.SpvError(f"PoEComponents supports at most {POE_COMP_CMAX} components")
        dev = stats[0].device
        n = stats[0].shape[1]
        Bs = [stats[2 * g].shape[0] for g in range(NG)]
        blocks = [_loc_logvar_block(stats[2 * g], stats[2 * g + 1]) for g in range(NG)]
        comp = [c.flatten().contiguous().float() for c in comp]
        eps = [e.contiguous() for e in eps]
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        W = 2 * n + 1
        part = ws.get("poen_part", (NG, POE_COMP_SEG, n_comp, W), torch.float32)
        mean = new(NG, n_comp, W)
        out = {k: [new(Bs[g], n) for g in range(NG)] for k in ("loc", "logvar", "scale", "logz", "theta")}
        kl = [new(Bs[g]) for g in range(NG)]
        a = SpvPoeCompArgs()
        a.ngroups, a.n, a.ncomp, a.part, a.mean = NG, n, n_comp, ptr(part), ptr(mean)
        for g in range(NG):
            a.B[g], a.stats[g], a.ld[g], a.comp[g], a.eps[g] = Bs[g], blocks[g][1], blocks[g][2], ptr(comp[g]), ptr(eps[g])
            a.loc[g], a.logvar[g], a.scale[g], a.logz[g], a.theta[g] = (ptr(out[k][g]) for k in ("loc", "logvar", "scale", "logz", "theta"))
            a.kl[g] = ptr(kl[g])
        _abi.call("spv_poe_comp_fwd", C.byref(a), stream_ptr())
        ctx.blocks, ctx.comp, ctx.eps, ctx.n, ctx.Bs, ctx.NG, ctx.n_comp, ctx.ws, ctx.mean = blocks, comp, eps, n, Bs, NG, n_comp, ws, mean
        ctx.save_for_backward(*out["loc"], *out["scale"])
        res = []
        for g in range(NG):
            qscale = out["scale"][g].clamp(min=1e-6)
            res += [out["loc"][g], out["logvar"][g], out["scale"][g], out["logz"][g], out["theta"][g], kl[g], qscale]
            ctx.mark_non_differentiable(out["theta"][g], qscale)
        return tuple(res)

    @staticmethod
    def backward(ctx, *g):
        from ._abi import POE_COMP_SEG, SpvPoeCompArgs
        NG, n, Bs = ctx.NG, ctx.n, ctx.Bs
        loc, scale = ctx.saved_tensors[:NG], ctx.saved_tensors[NG:2 * NG]
        dev = loc[0].device
        cont = lambda t: None if t is None else t.contiguous()
        a = SpvPoeCompArgs()
        W = 2 * n + 1
        part = ctx.ws.get("poen_part", (NG, POE_COMP_SEG, ctx.n_comp, W), torch.float32)
        a.ngroups, a.n, a.ncomp, a.part, a.mean = NG, n, ctx.n_comp, ptr(part), ptr(ctx.mean)
        keep, d = [], []
        for k in range(NG):
            gl, gv, gs, gz, _gt, gk, _gq = (cont(t) for t in g[7 * k: 7 * k + 7])
            dk = torch.empty((Bs[k], ctx.blocks[k][2]), dtype=torch.float32, device=dev)
            dpn = torch.empty((Bs[k], 2 * n), dtype=torch.float32, device=dev)
            keep += [gl, gv, gs, gz, gk, dpn]
            d.append(dk)
            a.B[k], a.stats[k], a.ld[k], a.comp[k], a.eps[k] = Bs[k], ctx.blocks[k][1], ctx.blocks[k][2], ptr(ctx.comp[k]), ptr(ctx.eps[k])
            a.loc[k], a.scale[k] = ptr(loc[k]), ptr(scale[k])
            a.g_loc[k], a.g_logvar[k], a.g_scale[k], a.g_logz[k], a.g_kl[k] = ptr(gl), ptr(gv), ptr(gs), ptr(gz), ptr(gk)
            a.dpn[k], a.d_stats[k] = ptr(dpn), ptr(dk)
        _abi.call("spv_poe_comp_bwd", C.byref(a), stream_ptr())
        grads = []
        for k in range(NG):
            grads += [d[k][:, :n], d[k][:, n:2 * n]]
        return (None, None, None, None, *grads)
