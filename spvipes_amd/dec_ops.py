"""The whole decoder side of the step for BOTH groups as one autograd node: latent slicing, BatchNorm
folding of the two factor regressors, mixing trunk, operand packing, logits GEMM, softmax statistics,
NB-mixture likelihood -- and the hand-written backward of all of it.

Reference arithmetic replaced (file:line into /root/reference/src/spVIPES):
    module/spVIPESmodule.py:733-759 (generative), nn/networks.py:314-325 (LinearDecoderSPVIPE.forward),
    module/spVIPESmodule.py:817-824 (reconstruction term of loss)

Launches per step (2 groups): 10 small batched kernels + per group {5 packs, tables, logits GEMM, lse x2,
likelihood} forward; backward per group {softmax fix, 6 GEMMs} + 8 small batched kernels.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Sequence

import torch

from . import _abi
from ._abi import (DEC_CELLS_PER_WG, DEC_KP, DEC_KPS, DEC_KS, NB_CMAX, TRUNK_KMAX, SpvBnBatch, SpvDecGroup, SpvDecParams, SpvFoldBatch, SpvGemmArgs, SpvReduceBatch, SpvTrunkBatch, SpvZsplitArgs,
                   ptr, round_up, stream_ptr)
from .nn_ops import _add_lin, _fptr, _lin_batch, _wgrad, grad_out
from . import ops as _ops
from .ops import N_HIDDEN_MIX, GroupCounts, Workspace, _bf16_image, _gemm_slabs, _gene_splits, _nb_cell_tiles, _nb_splits, _pack, fork, group_streams, join

N_DEC_PARAMS = 13  # Wp, gamma_p, beta_p, Ws, gamma_s, beta_s, Wa, ba, gamma_a, beta_a, Wm, bm, px_r
KMP = 320


def decoder_params(dec, px_r) -> List[torch.Tensor]:
    fp, fs, tr, mx = dec.factor_regressor_private, dec.factor_regressor_shared, dec.sigmoid_decoder, dec.mixture
    return [fp.linear.weight, fp.bn.weight, fp.bn.bias, fs.linear.weight, fs.bn.weight, fs.bn.bias,
            tr.linear.weight, tr.linear.bias, tr.bn.weight, tr.bn.bias, mx.linear.weight, mx.linear.bias, px_r]


def _add_red(b: SpvReduceBatch, src: torch.Tensor, nslabs: int, slab_stride: int, ld_src: int, rows: int, cols: int, dst: torch.Tensor,
             ld_dst: int, *, col_off: int = 0, dst_col: int = 0, accumulate: bool = False, alpha=None, exp_scale=None) -> None:
    """one problem of spv_reduce_slabs: dst[r][dst_col + c] (+)= alpha * exp(exp_scale[c]) * sum_s src[s][r][col_off + c]"""
    q = b.p[b.nprob]
    q.src, q.slab_stride, q.ld_src, q.nslabs, q.col_off, q.rows, q.cols = ptr(src), slab_stride, ld_src, nslabs, col_off, rows, cols
    q.dst, q.ld_dst, q.accumulate = dst.data_ptr() + 4 * dst_col, ld_dst, int(accumulate)
    q.alpha, q.exp_scale = ptr(alpha), ptr(exp_scale)
    b.nprob += 1


def _run_red(b: SpvReduceBatch) -> None:
    if b.nprob:
        _abi.call("spv_reduce_slabs", C.byref(b), stream_ptr())


class _DecoderBwd:
    """One group's decoder backward launches, as separately schedulable stages (none depends on the upstream gradient of
    the loss: that scalar is applied later by spv_reduce_slabs):
      softmax()  the softmax fix in place on tP / tS; in bf16 mode it also yields the latent gradient of the two rate heads
      gemm_d()   d A_m = dL W_m (contraction over genes): feeds the trunk backward          -- needs only dL
      gemm_bc()  d [W'_p | c_p], d [W'_s | c_s] (contractions over cells): feed the BatchNorm-fold backward -- after softmax()
      gemm_a()   d W_m = dL^T A_m: feeds only the optimiser                                   -- needs only dL
      gemm_ef()  fp32 mode only: the latent gradient of the rate heads as two GEMMs           -- after softmax()
    (Measured: issuing these from the forward pass right after the group's likelihood kernel, staggered so that the other
    group's VALU-bound likelihood kernel runs beside these memory-bound kernels, changes nothing.)"""

    def __init__(self, g: int, P, S, Wps_g, wsg, B: int, Bp: int, G: int, Gp: int, nsplit: int, grads_f32: bool, pair: bool = False):
        self.g, self.P, self.S, self.Wps_g, self.wsg = g, P, S, Wps_g, wsg
        self.B, self.Bp, self.G, self.Gp, self.nsplit, self.grads_f32 = B, Bp, G, Gp, nsplit, grads_f32
        self.fused_dz = bool(_ops.FUSED_DZ and (not grads_f32 or _ops.FUSED_DZ_F32))
        self.dz_part = wsg.get("dec_dz_part", (P.gene_splits, Bp, DEC_KPS), torch.float32) if self.fused_dz else None
        self.T = T = Gp // 32
        bt = Bp // 32
        # K splits (fp32 slabs, reduced in order by spv_reduce_slabs).  The 320-column GEMMs (d A_m, d W_m) get ~1 workgroup
        # per CU: every extra split adds a [M, 320] fp32 slab to write and to re-read (at C2 8 -> 4 and 4 -> 2 splits:
        # -0.05 ms per step, same-box A/B); the narrow ones (16 / 32 columns) stream their [B,G] operand with ~2 per CU
        self.ksp_m, self.ksp_n = max(1, min(T // 8, -(-256 // max(Bp // 64, 1)))), max(1, min(T // 8, -(-512 // max(Bp // 128, 1))))
        self.csp_m, self.csp_n = max(1, min(bt // 8, -(-256 // max(Gp // 64, 1)))), max(1, min(bt // 8, -(-512 // max(Gp // 128, 1))))
        # bf16 mode: one streaming kernel for both regressor weight gradients (csrc/spv_dec_gemm.h)
        self.heads_dma = bool(_ops.HEADS_DMA and nsplit == 1 and not grads_f32 and Bp % 64 == 0)
        if self.heads_dma:
            mt, kt = -(-G // 128), Bp // 64   # 128-gene workgroup tiles, two workgroups per CU
            self.csp_n = max(1, min(512 // mt if mt <= 512 else 1, max(kt // 4, 1)))
        # bf16 mode: the regressor weight gradients come out of the softmax-fix pass itself (spv_dec_heads_bwd), one partial slab per
        # 128-cell workgroup row
        self.fused_heads = bool(_ops.FUSED_HEADS and not _ops.DZ_ONLY and self.fused_dz and self.heads_dma and Bp % DEC_CELLS_PER_WG == 0)
        # "fp32" mode (split-bf16 gradient words, round 4): the same one-pass kernel on the hi / lo planes -- no in-place fix of the four
        # planes and no register-staged weight-gradient GEMMs re-reading them
        if _ops.FUSED_HEADS_F32 and grads_f32 and nsplit == 3 and not _ops.DZ_ONLY and self.fused_dz and Bp % DEC_CELLS_PER_WG == 0:
            self.fused_heads = True
        if self.fused_heads:
            self.csp_n = Bp // DEC_CELLS_PER_WG
            self.dw_part = wsg.get("dec_dW48", (self.csp_n, Gp, DEC_KPS), torch.float32)   # [d W'_p | d W'_s] rows over the padded genes
        # bf16 mode: the two 320-column GEMMs run the LDS-DMA 128 x 320 kernels, which want their own split counts
        # (d A_m of the two groups run side by side on the group streams; d W_m one after the other on the side stream)
        self.ksp_m = self._splits(False, B, G, self.ksp_m, 128 if (pair and _ops.DEC_PAIR_SPLITS) else 256)
        # the d A_m GEMM may sum its split-K slabs inside the launch (LDS-DMA kernels only)
        self.can_fix_d = bool((_ops.DA_FIXUP == "1" or (_ops.DA_FIXUP == "fp32" and nsplit == 3)) and _abi.load().spv_gemm_bf16_uses_dma(0, B, KMP, G, nsplit, self.T, KMP))
        self.csp_m = self._splits(True, G, Bp, self.csp_m)

    def _operand(self, key: str):
        """(hi, lo) operand image of dL / tP / tS: the bf16 tile array itself, or its hi / lo planes in fp32 mode"""
        if not self.grads_f32:
            return self.S[key], None
        t = self.S[key]   # [2 Bp][Gp]: hi plane, lo plane
        return t[:self.Bp], t[self.Bp:]

    def softmax(self):
        if self.fused_heads:
            _abi.call("spv_dec_heads_bwd", C.byref(self.P), ptr(self.S["Tp"]), ptr(self.S["Ts"]), ptr(self.dz_part), ptr(self.dw_part), stream_ptr())
            return
        _abi.call("spv_dec_softmax_bwd", C.byref(self.P), ptr(self.S["Tp"]), ptr(self.S["Ts"]), ptr(self.dz_part), stream_ptr())

    def dz_only(self):
        """read-only pass: the latent gradient of the rate heads (into dz_part), tP / tS left uncorrected"""
        _abi.call("spv_dec_dz", C.byref(self.P), ptr(self.S["Tp"]), ptr(self.S["Ts"]), ptr(self.dz_part), stream_ptr())

    def softmax_fix(self):
        """the in-place correction of tP / tS alone (what gemm_bc consumes)"""
        _abi.call("spv_dec_softmax_bwd", C.byref(self.P), ptr(self.S["Tp"]), ptr(self.S["Ts"]), None, stream_ptr())

    @staticmethod
    def _dma_splits(M: int, K: int, cus: int = 256) -> int:
        """K splits for the LDS-DMA 128 x 320 kernels (csrc/spv_dec_gemm.h): the count that minimises
        rounds-of-``cus``-workgroups x (64-deep steps per split + ~4 steps of prologue / slab store).  ``cus`` = 128 for a launch that
        runs beside its twin of the other group on a second stream (one workgroup per CU: the pair then shares ONE round)."""
        mt, kt = -(-M // 128), -(-K // 64)
        best, best_cost = 1, None
        for s_ in range(1, 17):
            if s_ > 1 and kt // s_ < 4:
                break
            cost = -(-(mt * s_) // cus) * (-(-kt // s_) + 4)
            if best_cost is None or cost < best_cost:
                best, best_cost = s_, cost
        return best

    def _splits(self, a_kmajor: bool, M: int, K: int, default: int, cus: int = 256) -> int:
        if _abi.load().spv_gemm_bf16_uses_dma(int(a_kmajor), M, KMP, K, self.nsplit, self.T, KMP):
            return self._dma_splits(M, K, cus)
        return default

    def gemm_d(self, fix=None):
        """``fix``: (alpha, dAm [B][n_m], d_zcat [B][nt], n_m, nt) -- the slabs are summed inside the launch (spv_gemm_bf16_fix) straight into the
        two consumers of d A_m; returns None then (nothing is left for spv_reduce_slabs)"""
        (Wm_hi, Wm_lo), (dL_hi, dL_lo) = self.S["Wm"], self._operand("dL")
        if fix is not None and self.can_fix_d:
            alpha, dAm, d_zcat, n_m, nt = fix
            M, N, K = self.B, KMP, self.G
            out = self.wsg.get("dec_dAm", (self.ksp_m, M, N), torch.float32)
            fx = _abi.SpvGemmFixup()
            fx.counters = ptr(self.wsg.get("dec_dAm_cnt", (-(-M // 128),), torch.int32, zero=True))
            fx.alpha, fx.dst0, fx.ld0, fx.n0 = ptr(alpha), ptr(dAm), n_m, n_m
            fx.dst1, fx.ld1, fx.c1, fx.n1 = ptr(d_zcat), nt, n_m, nt
            _abi.call("spv_gemm_bf16_fix", 0, ptr(dL_hi), ptr(dL_lo), self.Gp, ptr(Wm_hi), ptr(Wm_lo) if Wm_lo is not None else None, KMP, ptr(out), N, M, N, K,
                      self.nsplit, self.ksp_m, M * N, self.T, C.byref(fx), stream_ptr())
            return None
        return _gemm_slabs(False, dL_hi, dL_lo, self.Gp, Wm_hi, Wm_lo, KMP, self.B, KMP, self.G, self.nsplit,
                           self.ksp_m, self.wsg, "dec_dAm", a_tiles=self.T)

    def gemm_bc(self):
        if self.fused_heads:   # already produced by softmax() (spv_dec_heads_bwd): hand the partial slabs to the reduction
            return self.dw_part, self.dw_part   # (one 48-column slab: the reductions below pick their columns)
        (Aps_hi, Aps_lo), (tP_hi, tP_lo), (tS_hi, tS_lo) = self.S["Aps"], self._operand("tP"), self._operand("tS")
        if self.heads_dma:   # bf16 mode: both heads in one LDS-DMA streaming pass (csrc/spv_dec_gemm.h)
            b = self.wsg.get("dec_dWp", (self.csp_n, self.G, DEC_KP), torch.float32)
            c = self.wsg.get("dec_dWs", (self.csp_n, self.G, DEC_KS), torch.float32)
            _abi.call("spv_dec_heads_wgrad", ptr(tP_hi), ptr(tS_hi), self.T, ptr(Aps_hi), self.G, self.Bp, self.csp_n, ptr(b), ptr(c), stream_ptr())
            return b, c
        b = _gemm_slabs(True, tP_hi, tP_lo, self.Gp, Aps_hi, Aps_lo, DEC_KPS, self.G, DEC_KP, self.Bp, self.nsplit, self.csp_n, self.wsg, "dec_dWp", a_tiles=self.T)
        c = _gemm_slabs(True, tS_hi, tS_lo, self.Gp, Aps_hi, Aps_lo, DEC_KPS, self.G, DEC_KS, self.Bp, self.nsplit, self.csp_n, self.wsg, "dec_dWs",
                        b_col_off=DEC_KP, a_tiles=self.T)
        return b, c

    def gemm_a(self):
        (Am_hi, Am_lo), (dL_hi, dL_lo) = self.S["Am"], self._operand("dL")
        return _gemm_slabs(True, dL_hi, dL_lo, self.Gp, Am_hi, Am_lo, KMP, self.G, KMP, self.Bp, self.nsplit,
                           self.csp_m, self.wsg, "dec_dWm", a_tiles=self.T)

    def gemm_args(self, which: str, a: "SpvGemmArgs") -> torch.Tensor:
        """fill ``a`` with the spv_gemm_bf16 arguments of gemm_d ("d") / gemm_a ("a") for spv_gemm_bf16_grouped; returns the slab tensor"""
        (dL_hi, dL_lo) = self._operand("dL")
        if which == "d":
            (B_hi, B_lo), kmaj, M, K, splits, name = self.S["Wm"], 0, self.B, self.G, self.ksp_m, "dec_dAm"
        else:
            (B_hi, B_lo), kmaj, M, K, splits, name = self.S["Am"], 1, self.G, self.Bp, self.csp_m, "dec_dWm"
        out = self.wsg.get(name, (splits, M, KMP), torch.float32)
        a.a_kmajor, a.A_hi, a.A_lo, a.lda = kmaj, ptr(dL_hi), ptr(dL_lo), self.Gp
        a.B_hi, a.B_lo, a.ldb, a.C, a.ldc = ptr(B_hi), ptr(B_lo) if B_lo is not None else None, KMP, ptr(out), KMP
        a.M, a.N, a.K, a.nsplit, a.splits, a.a_tiles, a.slab_stride = M, KMP, K, self.nsplit, splits, self.T, M * KMP
        return out

    def gemm_ef(self):
        (tP_hi, tP_lo), (tS_hi, tS_lo) = self._operand("tP"), self._operand("tS")
        e = _gemm_slabs(False, tP_hi, tP_lo, self.Gp, self.Wps_g[0], self.Wps_g[1], DEC_KPS, self.B, DEC_KP, self.G, self.nsplit, self.ksp_n, self.wsg, "dec_dAp", a_tiles=self.T)
        f = _gemm_slabs(False, tS_hi, tS_lo, self.Gp, self.Wps_g[0], self.Wps_g[1], DEC_KPS, self.B, DEC_KS, self.G, self.nsplit, self.ksp_n, self.wsg, "dec_dAs",
                        b_col_off=DEC_KP, a_tiles=self.T)
        return e, f


class DecoderFused(torch.autograd.Function):
    """inputs : per group (private_log_z [B,n_p], poe_log_z [B,n_s]), per group its 13 parameters, then n_kl KL
             vectors [B] (module/spVIPESmodule.py:841-868) that the loss adds with weight kl_weight / B
    outputs: (loss, sum_g sum_b w_b rec_gb, rec_0 [B][, rec_1 [B]])   -- all but loss detached; one or two groups per call
    ``w_pad``: the per-cell weights of the reconstruction term, length >= Bp, zero beyond B.
    ``klw``  : 0-dim fp32 device tensor (read at run time: a captured graph sees later updates)."""

    @staticmethod
    def forward(ctx, counts: Sequence[GroupCounts], rows, B: int, decoders, library: Sequence[torch.Tensor], w_pad: torch.Tensor,
                training: bool, nsplit: int, ws: Sequence[Workspace], klw: torch.Tensor, n_kl: int, *tensors):
        NG = len(counts)   # one or two groups per call (the batched kernels carry two slots; spVIPESmodule.loss chunks more)
        if NG not in (1, 2):
            raise _abi.SpvError("DecoderFused takes one or two groups per call")
        ctx.set_materialize_grads(False)
        lat = [(tensors[2 * g], tensors[2 * g + 1]) for g in range(NG)]
        par = [tensors[2 * NG + g * N_DEC_PARAMS: 2 * NG + (g + 1) * N_DEC_PARAMS] for g in range(NG)]
        kls = [t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float() for t in tensors[2 * NG + NG * N_DEC_PARAMS:]]
        if len(kls) != n_kl or n_kl > 4:
            raise _abi.SpvError("DecoderFused: expected n_kl <= 4 KL vectors after the parameters")
        dev = lat[0][0].device
        n_p, n_s = lat[0][0].shape[1], lat[0][1].shape[1]
        nt = n_p + n_s
        if n_p + 1 > DEC_KP or n_s + 1 > DEC_KS:
            raise _abi.SpvError(f"decoder kernels support n_private <= {DEC_KP - 1} and n_shared <= {DEC_KS - 1}")
        n_m = par[0][6].shape[0]
        KM = n_m + nt + 1
        if KM > KMP:
            raise _abi.SpvError("mixture input wider than 320 columns is not supported")
        need_grad = any(ctx.needs_input_grad)  # (grad mode is off inside Function.forward)
        Bp = round_up(B, DEC_CELLS_PER_WG)
        Gs = [c.G for c in counts]
        Gps = [round_up(G, 256) for G in Gs]  # 256: the logits GEMM runs 256-gene workgroup tiles
        mlo = nsplit == 3
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        cont = lambda t: t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float()
        # ---- 1. latent slicing: zcat = [z_private | z_shared] --------------------------------------
        zcat = [new(B, nt) for _ in range(NG)]
        za = SpvZsplitArgs()
        za.B, za.n_p, za.n_s, za.ngroups = B, n_p, n_s, NG
        keep = []
        # ... and, from the same call, the decoder's packed bf16 operand images of these latents
        Am_img = [_bf16_image(ws[g], "dec_Am", Bp, KMP, mlo) for g in range(NG)]
        Aps_img = [_bf16_image(ws[g], "dec_Aps", Bp, DEC_KPS, True) for g in range(NG)]
        fused_pack = bool(_ops.FUSED_PACK)
        za.ld_am, za.am_col, za.am_cols, za.Bp = KMP, n_m, KMP - n_m, Bp
        for g in range(NG):
            pz, qz = cont(lat[g][0]), cont(lat[g][1])
            keep += [pz, qz]
            za.priv[g], za.poe[g], za.zcat[g] = ptr(pz), ptr(qz), ptr(zcat[g])
            if fused_pack:
                za.am_hi[g], za.am_lo[g], za.aps_hi[g], za.aps_lo[g] = ptr(Am_img[g][0]), ptr(Am_img[g][1]), ptr(Aps_img[g][0]), ptr(Aps_img[g][1])
        _abi.call("spv_zsplit_fwd", C.byref(za), stream_ptr())
        # ---- 2. batch statistics of z (column sums and z^T z) for the BatchNorm fold ---------------
        zsum = [[new(n_p), new(n_s)] for _ in range(NG)]
        zz = [[new(n_p, n_p), new(n_s, n_s)] for _ in range(NG)]
        # the mixing trunk's BatchNorm is folded too (spv_trunk_fold_fwd): it needs the statistics of ALL of zcat's columns, cross terms included
        trunk_fold = bool(_ops.TRUNK_FOLD and nt <= TRUNK_KMAX and n_m <= 256)
        zsum_t = [new(nt) for _ in range(NG)] if trunk_fold else None
        zz_t = [new(nt, nt) for _ in range(NG)] if trunk_fold else None
        if training:
            b = _lin_batch(B)
            for g in range(NG):
                for k, (off, n) in enumerate(((0, n_p), (n_p, n_s))):
                    _add_lin(b, N=n, K=n, W=ptr(zz[g][k]), X=_fptr(zcat[g], off), ldx=nt, dY=_fptr(zcat[g], off), lddy=nt, dW=ptr(zz[g][k]), db=ptr(zsum[g][k]))
                if trunk_fold:
                    _add_lin(b, N=nt, K=nt, W=ptr(zz_t[g]), X=ptr(zcat[g]), ldx=nt, dY=ptr(zcat[g]), lddy=nt, dW=ptr(zz_t[g]), db=ptr(zsum_t[g]))
            _wgrad(b, ws if not isinstance(ws, (list, tuple)) else ws[0])
        # ---- 3. fold the regressors' BatchNorm into the packed [Gp][48] operand image ---------------
        Wps = [_bf16_image(ws[g], "dec_Wps", Gps[g], DEC_KPS, True) for g in range(NG)]
        fstat = [[new(Gs[g], 2), new(Gs[g], 2)] for g in range(NG)]
        fb = SpvFoldBatch()
        fb.nprob, fb.B, fb.training, fb.eps, fb.momentum = 0, B, int(training), 1e-3, 0.01
        for g in range(NG):
            fp, fs = decoders[g].factor_regressor_private, decoders[g].factor_regressor_shared
            for k, (reg, W, gam, bet, off, slot, n, zoff) in enumerate(((fp, par[g][0], par[g][1], par[g][2], 0, DEC_KP, n_p, 0),
                                                                        (fs, par[g][3], par[g][4], par[g][5], DEC_KP, DEC_KS, n_s, n_p))):
                q = fb.p[fb.nprob]
                q.W, q.gamma, q.beta, q.running_mean, q.running_var = ptr(W), ptr(gam), ptr(bet), ptr(reg.bn.running_mean), ptr(reg.bn.running_var)
                q.zsum, q.zz, q.z, q.ldz, q.stat = ptr(zsum[g][k]), ptr(zz[g][k]), _fptr(zcat[g], zoff), nt, ptr(fstat[g][k])
                q.img_hi, q.img_lo, q.ld_img, q.col_off, q.slot = ptr(Wps[g][0]), ptr(Wps[g][1]), DEC_KPS, off, slot
                q.G, q.Gp, q.K = Gs[g], Gps[g], n
                fb.nprob += 1
        pair = _ops.dec_pair_for(B, NG)   # both groups' per-group launches as one grid per kernel (ops.DEC_PAIR)
        # The fold, the mixture-weight images and the (count, gene) tables depend on nothing the trunk (step 4) computes:
        # they run beside it on a side stream and are joined before the per-group section.
        side = group_streams(dev, 3)[2] if _ops.OVERLAP_SMALL else torch.cuda.current_stream(dev)
        if _ops.OVERLAP_SMALL:
            side.wait_stream(torch.cuda.current_stream(dev))
        Wm_img, tabs = [], []
        with torch.cuda.stream(side):
            _abi.call("spv_bn_fold_fwd", C.byref(fb), stream_ptr())
            for g in range(NG):
                Wm_hi, Wm_lo = _bf16_image(ws[g], "dec_Wm", Gps[g], KMP, mlo)
                if Wm_lo is not None or ws[g].fresh.get("dec_Wm") != _ops.image_token(par[g][10], par[g][11]):   # (else: Adam has just rewritten it)
                    _pack(cont(par[g][10]), Wm_hi, Wm_lo, extra_col=cont(par[g][11]))
                gene_tab = ws[g].get("dec_gene_tab", (Gps[g], 4), torch.float32)
                cnt_tab = ws[g].get("dec_cnt_tab", (NB_CMAX, Gps[g], 2), torch.float32)
                if not pair:
                    _abi.call("spv_dec_tables", ptr(cont(par[g][12])), Gs[g], Gps[g], ptr(gene_tab), ptr(cnt_tab), stream_ptr())
                Wm_img.append((Wm_hi, Wm_lo))
                tabs.append((gene_tab, cnt_tab))
            if pair:   # one grid for both groups' tables
                tg = (SpvDecGroup * NG)()
                px_keep = [cont(par[g][12]) for g in range(NG)]
                for g in range(NG):
                    tg[g].px_r, tg[g].p.G, tg[g].p.Gp, tg[g].p.gene_tab, tg[g].p.cnt_tab = ptr(px_keep[g]), Gs[g], Gps[g], ptr(tabs[g][0]), ptr(tabs[g][1])
                _abi.call("spv_dec_tables_grouped", tg, NG, stream_ptr())
        # ---- 4. mixing trunk: m = relu(BN(zcat Wa^T + ba)) -------------------------------------------
        m = [new(B, n_m) for _ in range(NG)]
        tstat = [new(n_m, 2) for _ in range(NG)]
        Wf = cf = pre_a = None
        if trunk_fold:
            # the BatchNorm folded into the layer (batch mean / variance of a linear map of z from zbar and cov(z)): one affine map + rectifier,
            # written by spv_linear_fwd straight into m AND into the logits operand image -- no statistics / finalise / normalise launches
            Wf, cf = [new(n_m, nt) for _ in range(NG)], [new(n_m) for _ in range(NG)]
            tb_ = SpvTrunkBatch()
            tb_.nprob, tb_.B, tb_.training, tb_.eps, tb_.momentum = NG, B, int(training), 1e-3, 0.01
            for g in range(NG):
                bnm = decoders[g].sigmoid_decoder.bn
                q = tb_.p[g]
                q.W, q.bias, q.gamma, q.beta = ptr(cont(par[g][6])), ptr(par[g][7]), ptr(par[g][8]), ptr(par[g][9])
                q.running_mean, q.running_var = ptr(bnm.running_mean), ptr(bnm.running_var)
                q.zsum, q.zz, q.Wf, q.cf, q.stat, q.N, q.K = ptr(zsum_t[g]), ptr(zz_t[g]), ptr(Wf[g]), ptr(cf[g]), ptr(tstat[g]), n_m, nt
            _abi.call("spv_trunk_fold_fwd", C.byref(tb_), stream_ptr())
            b = _lin_batch(B, relu=True)
            for g in range(NG):
                _add_lin(b, N=n_m, K=nt, W=ptr(Wf[g]), bias=ptr(cf[g]), X=ptr(zcat[g]), ldx=nt, Y=ptr(m[g]), ldy=n_m)
                if fused_pack:  # m straight into the logits operand image
                    q = b.p[g]
                    q.img_hi, q.img_lo, q.ld_img, q.img_rows = ptr(Am_img[g][0]), ptr(Am_img[g][1]), KMP, Bp
            _abi.call("spv_linear_fwd", C.byref(b), stream_ptr())
        else:
            pre_a = [new(B, n_m) for _ in range(NG)]
            b = _lin_batch(B)
            for g in range(NG):
                _add_lin(b, N=n_m, K=nt, W=ptr(par[g][6]), bias=ptr(par[g][7]), X=ptr(zcat[g]), ldx=nt, Y=ptr(pre_a[g]), ldy=n_m)
            _abi.call("spv_linear_fwd", C.byref(b), stream_ptr())
            nblk = -(-B // _abi.BN_ROWS)
            bn = SpvBnBatch()
            bn.nprob, bn.B, bn.training, bn.relu, bn.eps, bn.momentum = NG, B, int(training), 1, 1e-3, 0.01
            for g in range(NG):
                tb = decoders[g].sigmoid_decoder.bn
                q = bn.p[g]
                q.X, q.ldx, q.Y, q.ldy, q.gamma, q.beta = ptr(pre_a[g]), n_m, ptr(m[g]), n_m, ptr(par[g][8]), ptr(par[g][9])
                q.running_mean, q.running_var, q.stats, q.N = ptr(tb.running_mean), ptr(tb.running_var), ptr(tstat[g]), n_m
                q.part = ptr(ws[g].get("trunk_bn_part", (nblk, n_m, 2), torch.float32))
                if fused_pack:  # m straight into the logits operand image
                    q.img_hi, q.img_lo, q.ld_img, q.img_rows = ptr(Am_img[g][0]), ptr(Am_img[g][1]), KMP, Bp
            _abi.call("spv_bn_fwd", C.byref(bn), stream_ptr())
        # ---- 5. per group: operand images, tables, logits GEMM, softmax statistics, likelihood ------
        if w_pad.numel() < Bp or w_pad.dtype != torch.float32 or not w_pad.is_contiguous():
            raise _abi.SpvError("DecoderFused: w_pad must be contiguous fp32 of length >= round_up(B, 128)")
        grads_f32 = bool(need_grad and mlo)
        # bf16 mode: one bf16 array per gradient; fp32 mode: a bf16 hi plane followed by a bf16 lo plane ([2 Bp][Gp] rows):
        # the two operand images the split-bf16 GEMMs consume, written directly by the likelihood kernel
        gdt, gname, gplanes = (torch.int16, "split", 2) if grads_f32 else (torch.int16, "bf16", 1)
        P, saved_g = [], []
        rec = [new(B) for _ in range(NG)]  # (allocated before the fork: every temporary belongs to the main stream)
        red = SpvReduceBatch()
        red.nprob = 0
        if _ops.OVERLAP_SMALL:
            torch.cuda.current_stream(dev).wait_stream(side)
        streams = group_streams(dev, NG) if (_ops.FWD_GROUP_STREAMS and not pair) else [torch.cuda.current_stream(dev)] * NG
        fork(streams)
        dgrp = (SpvDecGroup * NG)() if pair else None
        lib_keep = []
        for g in range(NG):
          with torch.cuda.stream(streams[g]):
              G, Gp, wsg = Gs[g], Gps[g], ws[g]
              (Wm_hi, Wm_lo), (gene_tab, cnt_tab) = Wm_img[g], tabs[g]
              (Am_hi, Am_lo), (Aps_hi, Aps_lo) = Am_img[g], Aps_img[g]
              if not fused_pack:
                  _pack(m[g], Am_hi, Am_lo, dst_col_off=0, cslot=n_m)
                  _pack(zcat[g], Am_hi, Am_lo, extra_one=True, dst_col_off=n_m, cslot=KMP - n_m)
                  _pack(zcat[g][:, :n_p], Aps_hi, Aps_lo, extra_one=True, dst_col_off=0, cslot=DEC_KP)
                  _pack(zcat[g][:, n_p:], Aps_hi, Aps_lo, extra_one=True, dst_col_off=DEC_KP, cslot=DEC_KS)
              logits = wsg.get("dec_logits_" + ("f32" if mlo else "f16"), (Bp, Gp), torch.float32 if mlo else torch.float16)
              lse_first = bool(_ops.STAGGER and g % 2 == 1)  # group 1 runs its (VALU-bound) softmax statistics beside group 0's logits GEMM
              if not lse_first and not pair:
                  _abi.call("spv_dec_logits", ptr(Am_hi), ptr(Am_lo), ptr(Wm_hi), ptr(Wm_lo), KMP, Bp, Gp, nsplit, ptr(logits), int(mlo), stream_ptr())
              splits, per = _gene_splits(Bp, Gp)
              nbs, nbper = _nb_splits(Gp, Bp)
              vec = lambda nme: wsg.get(nme, (Bp,), torch.float32)
              part = lambda nme: wsg.get(nme, (splits, Bp), torch.float32)
              nbpart = lambda nme: wsg.get(nme, (nbs, Bp), torch.float32)
              if need_grad:
                  dL = wsg.get("dec_dL_" + gname, (gplanes * Bp, Gp), gdt, zero=True)
                  tP = wsg.get("dec_tP_" + gname, (gplanes * Bp, Gp), gdt, zero=True)
                  tS = wsg.get("dec_tS_" + gname, (gplanes * Bp, Gp), gdt, zero=True)
                  dth = wsg.get("dec_dtheta", (Bp // 64, Gp), torch.float32, zero=True)
                  Tp, Ts = wsg.get("dec_Tp", (Bp,), torch.float32), wsg.get("dec_Ts", (Bp,), torch.float32)
              else:
                  dL = tP = tS = dth = Tp = Ts = None
              lse_p, lse_s, a_p, a_s = vec("dec_lse_p"), vec("dec_lse_s"), vec("dec_a_p"), vec("dec_a_s")
              cst = counts[g].c_struct(rows[g])
              p = SpvDecParams(
                  X=cst.X, ldx=cst.ld, rows=cst.rows, col_off=cst.col_off, count_is_u16=int(cst.dtype == _abi.SPV_COUNT_U16),
                  B=B, G=G, Bp=Bp, Gp=Gp, logits=ptr(logits), n_gene_tiles=Gp // 32, logits_f32=int(mlo),
                  Wps_hi=ptr(Wps[g][0]), Wps_lo=ptr(Wps[g][1]), Aps_hi=ptr(Aps_hi), Aps_lo=ptr(Aps_lo),
                  gene_tab=ptr(gene_tab), cnt_tab=ptr(cnt_tab), a_p=ptr(a_p), a_s=ptr(a_s), lse_p=ptr(lse_p), lse_s=ptr(lse_s),
                  w_row=ptr(w_pad), gene_splits=splits, genes_per_split=per,
                  part_max_p=ptr(part("dec_pmp")), part_sum_p=ptr(part("dec_psp")), part_max_s=ptr(part("dec_pms")), part_sum_s=ptr(part("dec_pss")),
                  rec_part=ptr(nbpart("dec_rec")), tp_part=ptr(nbpart("dec_tp")), ts_part=ptr(nbpart("dec_ts")),
                  dtheta_part=ptr(dth), dL=ptr(dL), tP=ptr(tP), tS=ptr(tS), grads_f32=int(grads_f32), nb_splits=nbs, nb_genes_per_split=nbper,
                  nb_cell_tiles=_nb_cell_tiles(Bp, Gp),
              )
              if pair:
                  lib_keep.append(cont(library[g].flatten()))
                  q = dgrp[g]
                  q.p, q.library = p, ptr(lib_keep[-1])
                  q.Am_hi, q.Am_lo, q.Wm_hi, q.Wm_lo, q.K, q.nsplit = ptr(Am_hi), ptr(Am_lo), ptr(Wm_hi), ptr(Wm_lo), KMP, nsplit
              else:
                  _abi.call("spv_dec_lse", C.byref(p), ptr(cont(library[g].flatten())), stream_ptr())
                  if lse_first:
                      _abi.call("spv_dec_logits", ptr(Am_hi), ptr(Am_lo), ptr(Wm_hi), ptr(Wm_lo), KMP, Bp, Gp, nsplit, ptr(logits), int(mlo), stream_ptr())
                  _abi.call("spv_dec_nb_fwd", C.byref(p), int(need_grad), stream_ptr())
              r = rec[g]
              _add_red(red, nbpart("dec_rec"), nbs, Bp, Bp, 1, B, r, B)
              if need_grad:
                  _add_red(red, nbpart("dec_tp"), nbs, Bp, Bp, 1, Bp, Tp, Bp)
                  _add_red(red, nbpart("dec_ts"), nbs, Bp, Bp, 1, Bp, Ts, Bp)
              P.append(p)
              if need_grad:
                  saved_g.append(dict(Wm=(Wm_hi, Wm_lo), Am=(Am_hi, Am_lo), Aps=(Aps_hi, Aps_lo), dL=dL, tP=tP, tS=tS, dth=dth, Tp=Tp, Ts=Ts))
        join(streams)
        if pair:   # logits GEMM, softmax statistics and likelihood of both groups: one grid each
            _abi.call("spv_dec_logits_grouped", dgrp, NG, stream_ptr())
            _abi.call("spv_dec_lse_grouped", dgrp, NG, stream_ptr())
            _abi.call("spv_dec_nb_fwd_grouped", dgrp, NG, int(need_grad), stream_ptr())
        _run_red(red)
        loss, rec_sum, gkl = new(()), new(()), new(B)
        klp = (C.c_void_p * 4)(*[ptr(k) for k in kls], *([None] * (4 - n_kl)))
        _abi.call("spv_loss_assemble", ptr(rec[0]), ptr(rec[1]) if NG > 1 else None, ptr(w_pad), klp, n_kl, B, ptr(klw), ptr(loss), ptr(rec_sum), ptr(gkl), stream_ptr())
        if need_grad:
            ctx.P, ctx.saved_g, ctx.ws, ctx.decoders, ctx.training, ctx.nsplit = P, saved_g, ws, decoders, training, nsplit
            ctx.dims = (B, Bp, Gs, Gps, n_p, n_s, n_m, KM)
            ctx.NG = NG
            ctx.grads_f32, ctx.done, ctx.keep = grads_f32, False, keep
            ctx.Wps, ctx.fb = Wps, fb
            ctx.small = (zcat, zsum, zz, fstat, pre_a, m, tstat)
            ctx.trunk = (Wf, cf, zsum_t, zz_t) if trunk_fold else None
            ctx.gkl, ctx.n_kl = gkl, n_kl
            ctx.save_for_backward(*tensors[:2 * NG + NG * N_DEC_PARAMS])
        ctx.mark_non_differentiable(rec_sum, *rec)
        return (loss, rec_sum, *rec)

    @staticmethod
    def backward(ctx, g_loss, *_g_rec):
        if ctx.done:
            raise _abi.SpvError("DecoderFused.backward may run once per forward (gradient buffers are consumed in place)")
        ctx.done = True
        NG = ctx.NG
        tensors = ctx.saved_tensors
        par = [tensors[2 * NG + g * N_DEC_PARAMS: 2 * NG + (g + 1) * N_DEC_PARAMS] for g in range(NG)]
        B, Bp, Gs, Gps, n_p, n_s, n_m, KM = ctx.dims
        nt = n_p + n_s
        ws, nsplit, training = ctx.ws, ctx.nsplit, ctx.training
        zcat, zsum, zz, fstat, pre_a, m, tstat = ctx.small
        dev = zcat[0].device
        new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        if g_loss is None:
            raise _abi.SpvError("DecoderFused.backward: the loss output received no gradient")
        g_loss = g_loss if (g_loss.dtype == torch.float32 and g_loss.is_contiguous()) else g_loss.float().contiguous()
        pg = [[grad_out(par[g][j]) for j in range(N_DEC_PARAMS)] for g in range(NG)]  # (kernel target, autograd return) per parameter
        red, red2 = SpvReduceBatch(), SpvReduceBatch()
        red.nprob = red2.nprob = 0
        dWp, dWs = [new(Gs[g], DEC_KP) for g in range(NG)], [new(Gs[g], DEC_KS) for g in range(NG)]
        dAm, d_zcat = [new(B, n_m) for _ in range(NG)], [new(B, nt) for _ in range(NG)]
        # ---- schedule ---------------------------------------------------------------------------------------------
        # group g's softmax fix and d A_m GEMM on group stream g; the regressor weight-gradient GEMMs (first needed by the
        # BatchNorm-fold backward, which sits behind the ~15 small launches of the trunk backward) and then the
        # mixture-weight GEMMs (which feed only the optimiser) on the side stream, beside the trunk / PoE / encoder-tail
        # backward chain; the side stream is joined at the end of the backward pass (train.Trainer, DEFER_JOIN) or before
        # returning.  Same-box A/B: SPV_DEFER_BC 1.90 -> 1.82 ms.  (Measured and left out: running the softmax fix BESIDE
        # the d A_m GEMM + trunk backward instead of ahead of them, +0.01 ms.)
        cur = torch.cuda.current_stream(dev)
        streams = group_streams(dev, NG) if _ops.BWD_GROUP_STREAMS else [cur] * NG
        side = group_streams(dev, 3)[2] if _ops.DEFER_WM else cur
        stages = [_DecoderBwd(g, ctx.P[g], ctx.saved_g[g], ctx.Wps[g], ws[g], B, Bp, Gs[g], Gps[g], nsplit, ctx.grads_f32, pair=(NG == 2 and _ops.BWD_GROUP_STREAMS)) for g in range(NG)]
        d_slabs, bc_slabs, ef_slabs = [None] * NG, [None] * NG, [None] * NG
        # bf16 mode with the regressor GEMMs deferred to the side stream: the critical chain only needs the latent gradient, which
        # the READ-ONLY pass spv_dec_dz gives (168 MB read per group instead of 336 MB read + written); the in-place fix of
        # tP / tS runs on the side stream right before the GEMMs that consume it
        split_fix = bool(_ops.DZ_ONLY and _ops.DEFER_BC and _ops.DEFER_WM and all(st.fused_dz for st in stages))
        # DA_FIRST: the critical chain (trunk / fold / PoE / encoder backward) only needs d A_m up to the point where the latent
        # gradient of the rate heads joins (red2) -- so the two d A_m GEMMs go first and the HBM-bound softmax fixes move to the
        # side stream, ahead of the regressor and mixture weight-gradient GEMMs they feed
        da_first = int(_ops.DA_FIRST) if (_ops.DEFER_BC and _ops.DEFER_WM and side is not cur and not split_fix and all(st.fused_dz for st in stages)) else 0
        sm_done = None
        wm_late = bool(_ops.wm_late_for(max(Gs), bool(ctx.grads_f32)) and side is not cur)
        # ops.DEC_PAIR: the default schedule's per-group launches (d A_m GEMMs, one-pass backward, d W_m GEMMs) as one grid per kernel
        pair = bool(_ops.dec_pair_for(B, NG) and da_first in (0, 1) and not split_fix and _ops.DEFER_BC and all(st.fused_heads and not st.grads_f32 for st in stages))

        def pair_gemm(which):
            ga = (SpvGemmArgs * NG)()
            outs = [stages[g].gemm_args(which, ga[g]) for g in range(NG)]
            _abi.call("spv_gemm_bf16_grouped", ga, NG, stream_ptr())
            return outs

        def pair_heads_bwd():
            hg = (SpvDecGroup * NG)()
            for g in range(NG):
                st = stages[g]
                hg[g].p, hg[g].Tp, hg[g].Ts, hg[g].dz_part, hg[g].dw_part = st.P, ptr(st.S["Tp"]), ptr(st.S["Ts"]), ptr(st.dz_part), ptr(st.dw_part)
            _abi.call("spv_dec_heads_bwd_grouped", hg, NG, stream_ptr())

        if pair:
            streams = [cur] * NG
            if not da_first:   # (single stream: the one-pass backward first, as the per-group schedule orders it)
                pair_heads_bwd()
            d_slabs = pair_gemm("d")
        if da_first == 2:
            side.wait_stream(cur)
        fork(streams)
        for g in range(NG):
            if pair:
                continue
            if da_first:
                with torch.cuda.stream(streams[g]):
                    d_slabs[g] = stages[g].gemm_d(fix=(g_loss, dAm[g], d_zcat[g], n_m, nt))
                continue
            with torch.cuda.stream(streams[g]):
                # group 1 issues its (MFMA / LDS-bound) d A_m GEMM BEFORE its (HBM-bound) softmax fix, group 0 the other way round: the two
                # streams then pair a bandwidth-bound kernel with a compute-bound one instead of two of a kind (ops.STAGGER_BWD)
                gemm_first = bool(_ops.STAGGER_BWD and g % 2 == 1)
                if gemm_first:
                    d_slabs[g] = stages[g].gemm_d()
                if split_fix:
                    stages[g].dz_only()
                else:
                    stages[g].softmax()
                if not gemm_first:
                    d_slabs[g] = stages[g].gemm_d()
                if not _ops.DEFER_BC:
                    bc_slabs[g] = stages[g].gemm_bc()
                if not stages[g].fused_dz:
                    ef_slabs[g] = stages[g].gemm_ef()
        join(streams)
        if _ops.DEFER_WM and da_first != 2:
            side.wait_stream(cur)   # (before the reductions: the side work needs none of them)
        if da_first:
            with torch.cuda.stream(side):
                if pair:
                    pair_heads_bwd()
                for g in range(NG):
                    if pair:
                        break
                    stages[g].softmax()
                sm_done = torch.cuda.Event()
                sm_done.record(side)
        # slab sums, scaled by the upstream gradient, straight into their consumers' buffers
        al = g_loss
        for g in range(NG):
            st, G, Gp = stages[g], Gs[g], Gps[g]
            if bc_slabs[g] is not None:
                if st.fused_heads:
                    _add_red(red, bc_slabs[g][0], st.csp_n, st.Gp * DEC_KPS, DEC_KPS, G, DEC_KP, dWp[g], DEC_KP, alpha=al)                  # d [W'_p | c_p]
                    _add_red(red, bc_slabs[g][1], st.csp_n, st.Gp * DEC_KPS, DEC_KPS, G, DEC_KS, dWs[g], DEC_KS, col_off=DEC_KP, alpha=al)  # d [W'_s | c_s]
                else:
                    _add_red(red, bc_slabs[g][0], st.csp_n, G * DEC_KP, DEC_KP, G, DEC_KP, dWp[g], DEC_KP, alpha=al)      # d [W'_p | c_p]
                    _add_red(red, bc_slabs[g][1], st.csp_n, G * DEC_KS, DEC_KS, G, DEC_KS, dWs[g], DEC_KS, alpha=al)      # d [W'_s | c_s]
            if d_slabs[g] is not None:   # (else: the GEMM summed its slabs itself, straight into dAm / d_zcat -- spv_gemm_bf16_fix)
                _add_red(red, d_slabs[g], st.ksp_m, B * KMP, KMP, B, n_m, dAm[g], n_m, alpha=al)                          # d m (trunk output)
                # gradient reaching zcat directly: through the logits GEMM (columns n_m..) and the two regressors
                _add_red(red, d_slabs[g], st.ksp_m, B * KMP, KMP, B, nt, d_zcat[g], nt, col_off=n_m, alpha=al)
            if st.fused_dz:
                _add_red(red2, st.dz_part, st.P.gene_splits, Bp * DEC_KPS, DEC_KPS, B, n_p, d_zcat[g], nt, accumulate=True, alpha=al)
                _add_red(red2, st.dz_part, st.P.gene_splits, Bp * DEC_KPS, DEC_KPS, B, n_s, d_zcat[g], nt, col_off=DEC_KP, dst_col=n_p, accumulate=True, alpha=al)
            else:
                _add_red(red2, ef_slabs[g][0], st.ksp_n, B * DEC_KP, DEC_KP, B, n_p, d_zcat[g], nt, accumulate=True, alpha=al)
                _add_red(red2, ef_slabs[g][1], st.ksp_n, B * DEC_KS, DEC_KS, B, n_s, d_zcat[g], nt, dst_col=n_p, accumulate=True, alpha=al)
            # d px_r = exp(px_r) * d theta (theta = exp(px_r): module/spVIPESmodule.py:758)
            _add_red(red, ctx.saved_g[g]["dth"], -(-(Bp // 64) // _nb_cell_tiles(Bp, Gp)), Gp, Gp, 1, G, pg[g][12][0], G, alpha=al, exp_scale=par[g][12])
        gk = None
        if ctx.n_kl:  # d loss / d kl_i[b] = g * kl_weight / B for every KL vector: rides in the second reduction launch
            gk = new(B)
            _add_red(red2, ctx.gkl, 1, B, B, 1, B, gk, B, alpha=g_loss)
        _run_red(red)
        if not da_first:
            _run_red(red2)
        # side stream: the regressor weight gradients (the main stream picks them up with an event before
        # spv_bn_fold_bwd), then the mixture-weight gradients (2 x [G, K_M] from dL^T A_m, the largest GEMMs of the backward
        # pass, which feed nothing but the optimiser)
        bc_done = None
        with torch.cuda.stream(side):
            if _ops.DEFER_BC:
                red_bc = SpvReduceBatch()
                red_bc.nprob = 0
                for g in range(NG):
                    st, G = stages[g], Gs[g]
                    if split_fix:
                        st.softmax_fix()
                    b_, c = st.gemm_bc()
                    # (spv_dec_heads_bwd's slabs are [Gp][32] for both heads; the GEMM slabs [G][16] / [G][32])
                    if st.fused_heads:   # spv_dec_heads_bwd's slabs: [Gp][48], private columns first
                        _add_red(red_bc, b_, st.csp_n, st.Gp * DEC_KPS, DEC_KPS, G, DEC_KP, dWp[g], DEC_KP, alpha=g_loss)
                        _add_red(red_bc, c, st.csp_n, st.Gp * DEC_KPS, DEC_KPS, G, DEC_KS, dWs[g], DEC_KS, col_off=DEC_KP, alpha=g_loss)
                    else:
                        _add_red(red_bc, b_, st.csp_n, G * DEC_KP, DEC_KP, G, DEC_KP, dWp[g], DEC_KP, alpha=g_loss)
                        _add_red(red_bc, c, st.csp_n, G * DEC_KS, DEC_KS, G, DEC_KS, dWs[g], DEC_KS, alpha=g_loss)
                _run_red(red_bc)
                if side is not cur:
                    bc_done = torch.cuda.Event()
                    bc_done.record(side)

        def issue_wm(after=None):
            with torch.cuda.stream(side):
                if after is not None:
                    side.wait_event(after)
                red3 = SpvReduceBatch()
                red3.nprob = 0
                wm_slabs = pair_gemm("a") if pair else None
                for g in range(NG):
                    st, G = stages[g], Gs[g]
                    a = wm_slabs[g] if pair else st.gemm_a()
                    _add_red(red3, a, st.csp_m, G * KMP, KMP, G, KM - 1, pg[g][10][0], KM - 1, alpha=g_loss)                 # d W_m
                    _add_red(red3, a, st.csp_m, G * KMP, KMP, G, 1, pg[g][11][0], 1, col_off=KM - 1, alpha=g_loss)           # d b_m
                _run_red(red3)
            if not _ops.DEFER_WM:
                pass
            elif _ops.DEFER_JOIN:
                _ops.defer(side, [g_loss])
            else:
                cur.wait_stream(side)

        # WM_LATE: the mixture-weight GEMMs are held back until the main stream reaches the BatchNorm-fold backward: they then run beside
        # the ~25 tiny launches of the fold / PoE / encoder-tail backward, where the GPU is otherwise almost idle, instead of beside the
        # trunk backward, which they slow down
        if not wm_late:
            issue_wm()
        # ---- trunk backward (BatchNorm + relu, Linear) ----------------------------------------------
        dWa, dba = [pg[g][6][0] for g in range(NG)], [pg[g][7][0] for g in range(NG)]
        d_gam_a, d_bet_a = [pg[g][8][0] for g in range(NG)], [pg[g][9][0] for g in range(NG)]
        if ctx.trunk is not None:
            # folded trunk (spv_trunk_fold_fwd): the layer is relu(zcat W'^T + c'); its input gradient and the gradients of (W', c') come from the
            # rectifier-masked linear kernels, spv_trunk_fold_bwd takes (d W', d c') back to W_a / gamma / beta and adds the part of d zcat that
            # flows through the batch statistics -- the three BatchNorm-backward launches are gone from this chain
            Wf, cf, zsum_t, zz_t = ctx.trunk
            dWf, dcf = [new(n_m, nt) for _ in range(NG)], [new(n_m) for _ in range(NG)]
            dred = [new(-(-n_m // 64), nt + nt * nt) for _ in range(NG)]
            bw, bd = _lin_batch(B, relu=True), _lin_batch(B, relu=True, accumulate=True)
            for g in range(NG):
                _add_lin(bw, N=n_m, K=nt, W=ptr(Wf[g]), X=ptr(zcat[g]), ldx=nt, Y=ptr(m[g]), ldy=n_m, dY=ptr(dAm[g]), lddy=n_m, dW=ptr(dWf[g]), db=ptr(dcf[g]))
                _add_lin(bd, N=n_m, K=nt, W=ptr(Wf[g]), Y=ptr(m[g]), ldy=n_m, dY=ptr(dAm[g]), lddy=n_m, dX=ptr(d_zcat[g]), lddx=nt)
            _abi.call("spv_linear_dgrad", C.byref(bd), stream_ptr())
            _wgrad(bw, ws if not isinstance(ws, (list, tuple)) else ws[0])
            tb_ = SpvTrunkBatch()
            tb_.nprob, tb_.B, tb_.training, tb_.eps, tb_.momentum = NG, B, int(training), 1e-3, 0.01
            wa_c = [par[g][6] if par[g][6].is_contiguous() else par[g][6].contiguous() for g in range(NG)]
            for g in range(NG):
                bnm = ctx.decoders[g].sigmoid_decoder.bn
                q = tb_.p[g]
                q.W, q.bias, q.gamma, q.beta = ptr(wa_c[g]), ptr(par[g][7]), ptr(par[g][8]), ptr(par[g][9])
                q.running_mean, q.running_var = ptr(bnm.running_mean), ptr(bnm.running_var)
                q.zsum, q.zz, q.Wf, q.cf, q.stat, q.N, q.K = ptr(zsum_t[g]), ptr(zz_t[g]), ptr(Wf[g]), ptr(cf[g]), ptr(tstat[g]), n_m, nt
                q.dWf, q.dcf, q.dW, q.dbias, q.dgamma, q.dbeta = ptr(dWf[g]), ptr(dcf[g]), ptr(dWa[g]), ptr(dba[g]), ptr(d_gam_a[g]), ptr(d_bet_a[g])
                q.dred, q.z, q.ldz, q.dz, q.lddz = ptr(dred[g]), ptr(zcat[g]), nt, ptr(d_zcat[g]), nt
            _abi.call("spv_trunk_fold_bwd", C.byref(tb_), stream_ptr())
        else:
            nblk = -(-B // _abi.BN_ROWS)
            d_pre = [new(B, n_m) for _ in range(NG)]
            bn = SpvBnBatch()
            bn.nprob, bn.B, bn.training, bn.relu, bn.eps, bn.momentum = NG, B, int(training), 1, 1e-3, 0.01
            for g in range(NG):
                tb = ctx.decoders[g].sigmoid_decoder.bn
                q = bn.p[g]
                q.X, q.ldx, q.Y, q.ldy, q.gamma, q.beta = ptr(pre_a[g]), n_m, ptr(m[g]), n_m, ptr(par[g][8]), ptr(par[g][9])
                q.running_mean, q.running_var, q.stats, q.N = ptr(tb.running_mean), ptr(tb.running_var), ptr(tstat[g]), n_m
                q.part = ptr(ws[g].get("trunk_bn_part", (nblk, n_m, 2), torch.float32))
                q.dY, q.lddy, q.dX, q.lddx, q.dgamma, q.dbeta = ptr(dAm[g]), n_m, ptr(d_pre[g]), n_m, ptr(d_gam_a[g]), ptr(d_bet_a[g])
            _abi.call("spv_bn_bwd", C.byref(bn), stream_ptr())
            bw, bd = _lin_batch(B), _lin_batch(B, accumulate=True)
            for g in range(NG):
                _add_lin(bw, N=n_m, K=nt, W=ptr(par[g][6]), X=ptr(zcat[g]), ldx=nt, dY=ptr(d_pre[g]), lddy=n_m, dW=ptr(dWa[g]), db=ptr(dba[g]))
                _add_lin(bd, N=n_m, K=nt, W=ptr(par[g][6]), dY=ptr(d_pre[g]), lddy=n_m, dX=ptr(d_zcat[g]), lddx=nt)
            _wgrad(bw, ws if not isinstance(ws, (list, tuple)) else ws[0])
            _abi.call("spv_linear_dgrad", C.byref(bd), stream_ptr())
        # ---- BatchNorm-fold backward (+ the z statistics it used) ------------------------------------
        fb = ctx.fb
        d_priv, d_poe = [new(B, n_p) for _ in range(NG)], [new(B, n_s) for _ in range(NG)]
        dWraw = [[pg[g][0][0], pg[g][3][0]] for g in range(NG)]
        dgam = [[pg[g][1][0], pg[g][4][0]] for g in range(NG)]
        dbet = [[pg[g][2][0], pg[g][5][0]] for g in range(NG)]
        i = 0
        for g in range(NG):
            for k, (dweff, ld, n, zoff) in enumerate(((dWp[g], DEC_KP, n_p, 0), (dWs[g], DEC_KS, n_s, n_p))):
                q = fb.p[i]
                q.dWeff, q.ld_dw, q.dW, q.dgamma, q.dbeta = ptr(dweff), ld, ptr(dWraw[g][k]), ptr(dgam[g][k]), ptr(dbet[g][k])
                q.red_part = ptr(ws[g].get(f"fold_red_{k}", (-(-Gs[g] // 256) + 1, n + n * n), torch.float32))
                q.dz, q.lddz = _fptr(d_zcat[g], zoff), nt
                if training:   # the latent-slicing backward rides in the same kernel (spv_fold_prob.out_priv)
                    q.zcol, q.out_priv, q.out_poe, q.n_p, q.n_s = zoff, ptr(d_priv[g]), ptr(d_poe[g]), n_p, n_s
                i += 1
        if da_first:   # the rate heads' latent gradient (softmax fix on the side stream) joins d_zcat here
            cur.wait_event(sm_done)
            _run_red(red2)
        if bc_done is not None:
            cur.wait_event(bc_done)
        if wm_late:
            ev = torch.cuda.Event()
            ev.record(cur)
            issue_wm(ev)
        _abi.call("spv_bn_fold_bwd", C.byref(fb), stream_ptr())
        # ---- latent slicing backward (training: already done by the z-statistics kernel of spv_bn_fold_bwd) ---------------------
        if not training:
            za = SpvZsplitArgs()
            za.B, za.n_p, za.n_s, za.ngroups = B, n_p, n_s, NG
            for g in range(NG):
                za.d_zcat[g], za.d_priv[g], za.d_poe[g] = ptr(d_zcat[g]), ptr(d_priv[g]), ptr(d_poe[g])
            _abi.call("spv_zsplit_bwd", C.byref(za), stream_ptr())
        grads = []
        for g in range(NG):
            grads += [d_priv[g], d_poe[g]]
        for g in range(NG):
            grads += [pg[g][j][1] for j in range(N_DEC_PARAMS)]
        if ctx.n_kl:
            grads += [gk] * ctx.n_kl
        return (None,) * 11 + tuple(grads)


@torch.no_grad()
def materialize_decoder(decoder, px_r: torch.Tensor, private_log_z: torch.Tensor, poe_log_z: torch.Tensor, library: torch.Tensor,
                        training: bool, nsplit: int, ws: Workspace, par=None) -> dict:
    """The decoder outputs the reference's ``generative`` returns for ONE group (module/spVIPESmodule.py:751-768,
    nn/networks.py:314-325): {"px_scale_private", "px_scale_shared", "px_rate_private", "px_rate_shared", "px_mixing"}
    as fp32 [B, G] tensors + "px_r" = exp(px_r).  The training step never needs them (the fused likelihood kernel consumes
    the same quantities in registers); this runs the same operand preparation as ``DecoderFused.forward`` for one group
    -- latent slicing quirk, BatchNorm fold (batch statistics in training mode, WITHOUT touching the running statistics:
    momentum 0), mixing trunk, logits GEMM, softmax statistics -- and then ``spv_dec_materialize``."""
    dev = private_log_z.device
    cont = lambda t: t if (t.is_contiguous() and t.dtype == torch.float32) else t.contiguous().float()
    par = list(par) if par is not None else decoder_params(decoder, px_r)   # (``par``: spVIPESmodule._decoder_operands' covariate-extended set)
    pz, qz = cont(private_log_z), cont(poe_log_z)
    B, n_p, n_s = pz.shape[0], pz.shape[1], qz.shape[1]
    nt = n_p + n_s
    if n_p + 1 > DEC_KP or n_s + 1 > DEC_KS:
        raise _abi.SpvError(f"decoder kernels support n_private <= {DEC_KP - 1} and n_shared <= {DEC_KS - 1}")
    n_m = par[6].shape[0]
    G = par[0].shape[0]
    Bp, Gp = round_up(B, DEC_CELLS_PER_WG), round_up(G, 256)
    mlo = nsplit == 3
    new = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
    zcat = new(B, nt)
    Am_hi, Am_lo = _bf16_image(ws, "mat_Am", Bp, KMP, mlo)
    Aps_hi, Aps_lo = _bf16_image(ws, "mat_Aps", Bp, DEC_KPS, True)
    za = SpvZsplitArgs()
    za.B, za.n_p, za.n_s, za.ngroups = B, n_p, n_s, 1
    za.ld_am, za.am_col, za.am_cols, za.Bp = KMP, n_m, KMP - n_m, Bp
    za.priv[0], za.poe[0], za.zcat[0] = ptr(pz), ptr(qz), ptr(zcat)
    za.am_hi[0], za.am_lo[0], za.aps_hi[0], za.aps_lo[0] = ptr(Am_hi), ptr(Am_lo), ptr(Aps_hi), ptr(Aps_lo)
    _abi.call("spv_zsplit_fwd", C.byref(za), stream_ptr())
    zsum, zz = [new(n_p), new(n_s)], [new(n_p, n_p), new(n_s, n_s)]
    if training:
        b = _lin_batch(B)
        for k, (off, n) in enumerate(((0, n_p), (n_p, n_s))):
            _add_lin(b, N=n, K=n, W=ptr(zz[k]), X=_fptr(zcat, off), ldx=nt, dY=_fptr(zcat, off), lddy=nt, dW=ptr(zz[k]), db=ptr(zsum[k]))
        _wgrad(b, ws)
    Wps_hi, Wps_lo = _bf16_image(ws, "mat_Wps", Gp, DEC_KPS, True)
    fstat = [new(G, 2), new(G, 2)]
    fb = SpvFoldBatch()
    fb.nprob, fb.B, fb.training, fb.eps, fb.momentum = 0, B, int(training), 1e-3, 0.0   # momentum 0: running statistics untouched
    fp, fs = decoder.factor_regressor_private, decoder.factor_regressor_shared
    for k, (reg, W, gam, bet, off, slot, n, zoff) in enumerate(((fp, par[0], par[1], par[2], 0, DEC_KP, n_p, 0),
                                                                (fs, par[3], par[4], par[5], DEC_KP, DEC_KS, n_s, n_p))):
        q = fb.p[fb.nprob]
        q.W, q.gamma, q.beta, q.running_mean, q.running_var = ptr(W), ptr(gam), ptr(bet), ptr(reg.bn.running_mean), ptr(reg.bn.running_var)
        q.zsum, q.zz, q.z, q.ldz, q.stat = ptr(zsum[k]), ptr(zz[k]), _fptr(zcat, zoff), nt, ptr(fstat[k])
        q.img_hi, q.img_lo, q.ld_img, q.col_off, q.slot = ptr(Wps_hi), ptr(Wps_lo), DEC_KPS, off, slot
        q.G, q.Gp, q.K = G, Gp, n
        fb.nprob += 1
    _abi.call("spv_bn_fold_fwd", C.byref(fb), stream_ptr())
    Wm_hi, Wm_lo = _bf16_image(ws, "mat_Wm", Gp, KMP, mlo)
    _pack(cont(par[10]), Wm_hi, Wm_lo, extra_col=cont(par[11]))
    # mixing trunk
    pre_a, m, tstat = new(B, n_m), new(B, n_m), new(n_m, 2)
    b = _lin_batch(B)
    _add_lin(b, N=n_m, K=nt, W=ptr(par[6]), bias=ptr(par[7]), X=ptr(zcat), ldx=nt, Y=ptr(pre_a), ldy=n_m)
    _abi.call("spv_linear_fwd", C.byref(b), stream_ptr())
    nblk = -(-B // _abi.BN_ROWS)
    bn = SpvBnBatch()
    bn.nprob, bn.B, bn.training, bn.relu, bn.eps, bn.momentum = 1, B, int(training), 1, 1e-3, 0.0
    tb = decoder.sigmoid_decoder.bn
    q = bn.p[0]
    q.X, q.ldx, q.Y, q.ldy, q.gamma, q.beta = ptr(pre_a), n_m, ptr(m), n_m, ptr(par[8]), ptr(par[9])
    q.running_mean, q.running_var, q.stats, q.N = ptr(tb.running_mean), ptr(tb.running_var), ptr(tstat), n_m
    q.part = ptr(ws.get("mat_bn_part", (nblk, n_m, 2), torch.float32))
    q.img_hi, q.img_lo, q.ld_img, q.img_rows = ptr(Am_hi), ptr(Am_lo), KMP, Bp
    _abi.call("spv_bn_fwd", C.byref(bn), stream_ptr())
    logits = ws.get("mat_logits_" + ("f32" if mlo else "f16"), (Bp, Gp), torch.float32 if mlo else torch.float16)
    _abi.call("spv_dec_logits", ptr(Am_hi), ptr(Am_lo), ptr(Wm_hi), ptr(Wm_lo), KMP, Bp, Gp, nsplit, ptr(logits), int(mlo), stream_ptr())
    splits, per = _gene_splits(Bp, Gp)
    vec = lambda nme: ws.get(nme, (Bp,), torch.float32)
    part = lambda nme: ws.get(nme, (splits, Bp), torch.float32)
    lse_p, lse_s, a_p, a_s = vec("mat_lse_p"), vec("mat_lse_s"), vec("mat_a_p"), vec("mat_a_s")
    p = SpvDecParams(
        X=None, ldx=0, rows=None, col_off=0, count_is_u16=0, B=B, G=G, Bp=Bp, Gp=Gp, logits=ptr(logits), n_gene_tiles=Gp // 32, logits_f32=int(mlo),
        Wps_hi=ptr(Wps_hi), Wps_lo=ptr(Wps_lo), Aps_hi=ptr(Aps_hi), Aps_lo=ptr(Aps_lo), gene_tab=None, cnt_tab=None,
        a_p=ptr(a_p), a_s=ptr(a_s), lse_p=ptr(lse_p), lse_s=ptr(lse_s), w_row=None, gene_splits=splits, genes_per_split=per,
        part_max_p=ptr(part("mat_pmp")), part_sum_p=ptr(part("mat_psp")), part_max_s=ptr(part("mat_pms")), part_sum_s=ptr(part("mat_pss")),
        rec_part=None, tp_part=None, ts_part=None, dtheta_part=None, dL=None, tP=None, tS=None, grads_f32=0, nb_splits=1, nb_genes_per_split=32,
    )
    _abi.call("spv_dec_lse", C.byref(p), ptr(cont(library.flatten())), stream_ptr())
    out = {k: new(B, G) for k in ("px_scale_private", "px_scale_shared", "px_rate_private", "px_rate_shared", "px_mixing")}
    _abi.call("spv_dec_materialize", C.byref(p), ptr(out["px_scale_private"]), ptr(out["px_scale_shared"]), ptr(out["px_rate_private"]),
              ptr(out["px_rate_shared"]), ptr(out["px_mixing"]), G, stream_ptr())
    out["px_r"] = torch.exp(cont(par[12]))
    return out
