"""TEST INFRASTRUCTURE: thin harnesses that drive single C-ABI entry points of libspvipes_hip.so in isolation (one group's
decoder + likelihood kernels without the rest of the step, a plain split-K GEMM).  Nothing in the product package uses
them: the shipped path is ``spvipes_amd.dec_ops.DecoderFused``.  Reference arithmetic exercised:
    DecoderNBLoss   nn/networks.py:314-325 (rates, mixing logits) + module/spVIPESmodule.py:758-759,817-824
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import torch

from spvipes_amd import _abi
from spvipes_amd._abi import DEC_CELLS_PER_WG, DEC_KP, DEC_KPS, DEC_KS, NB_CMAX, SpvDecParams, ptr, round_up, stream_ptr
from spvipes_amd.ops import GroupCounts, Workspace, _bf16_image, _gene_splits, _nb_splits, _pack


def _gemm(a_kmajor: bool, A_hi, A_lo, lda, B_hi, B_lo, ldb, M, N, K, nsplit, splits, ws: Workspace, name: str,
          b_col_off: int = 0, a_tiles: int = 0) -> torch.Tensor:
    out = ws.get(name, (splits, M, N), torch.float32)
    b_off = b_col_off * 2
    _abi.call("spv_gemm_bf16", int(a_kmajor), ptr(A_hi), ptr(A_lo), lda, ptr(B_hi) + b_off, (ptr(B_lo) + b_off) if B_lo is not None else None,
                            ldb, ptr(out), N, M, N, K, nsplit, splits, M * N, a_tiles, stream_ptr())
    return out[0] if splits == 1 else out.sum(0)


class DecoderNBLoss(torch.autograd.Function):
    """sum_b w_b * rec_b with rec_b = -sum_g log NBMixture(log1p(x_bg); mu1, mu2, theta_g, logits_bg),
    where mu_k = exp(library_b) * softmax_g(z_k[b] . W'_k[g] + c_k[g]) and
    logits = [m | z_p | z_s][b] . Wm[g] + bm[g].  Returns (weighted sum, rec[B] detached)."""

    @staticmethod
    def forward(ctx, counts: GroupCounts, rows, B: int, zp, zs, m, Wp, cp, Ws, cs_, Wm, bm, px_r, library, w_row,
                nsplit: int, train: bool, ws: Workspace):
        dev = zp.device
        G = counts.G
        n_p, n_s = zp.shape[1], zs.shape[1]
        if n_p + 1 > DEC_KP or n_s + 1 > DEC_KS:
            raise _abi.SpvError(f"decoder kernels support n_private <= {DEC_KP - 1} and n_shared <= {DEC_KS - 1}")
        KM = m.shape[1] + n_p + n_s + 1
        KMp = 320  # one 320-wide N tile of the backward GEMMs; K steps beyond `ksteps` are never issued
        if KM > KMp:
            raise _abi.SpvError("mixture input wider than 320 columns is not supported")
        Bp, Gp = round_up(B, DEC_CELLS_PER_WG), round_up(G, 256)
        lo = True  # the small regressor operands always travel as hi/lo pairs
        mlo = nsplit == 3
        f32 = lambda t: t.contiguous().float()
        # ---- packed operand images -------------------------------------------------------------
        Wm_hi, Wm_lo = _bf16_image(ws, "dec_Wm", Gp, KMp, mlo)
        _pack(f32(Wm), Wm_hi, Wm_lo, extra_col=f32(bm))
        Am_hi, Am_lo = _bf16_image(ws, "dec_Am", Bp, KMp, mlo)
        _pack(torch.cat([m, zp, zs], dim=1).contiguous(), Am_hi, Am_lo, extra_one=True)
        Wps_hi, Wps_lo = _bf16_image(ws, "dec_Wps", Gp, DEC_KPS, lo)
        _pack(f32(Wp), Wps_hi, Wps_lo, extra_col=f32(cp), dst_col_off=0, cslot=DEC_KP)
        _pack(f32(Ws), Wps_hi, Wps_lo, extra_col=f32(cs_), dst_col_off=DEC_KP, cslot=DEC_KS)
        Aps_hi, Aps_lo = _bf16_image(ws, "dec_Aps", Bp, DEC_KPS, lo)
        _pack(f32(zp), Aps_hi, Aps_lo, extra_one=True, dst_col_off=0, cslot=DEC_KP)
        _pack(f32(zs), Aps_hi, Aps_lo, extra_one=True, dst_col_off=DEC_KP, cslot=DEC_KS)
        # ---- tables / per-cell vectors -----------------------------------------------------------
        gene_tab = ws.get("dec_gene_tab", (Gp, 4), torch.float32)
        cnt_tab = ws.get("dec_cnt_tab", (NB_CMAX, Gp, 2), torch.float32)
        _abi.call("spv_dec_tables", ptr(f32(px_r)), G, Gp, ptr(gene_tab), ptr(cnt_tab), stream_ptr())
        splits, per = _gene_splits(Bp, Gp)
        nbs, nbper = _nb_splits(Gp)
        vec = lambda n: ws.get(n, (Bp,), torch.float32)
        part = lambda n: ws.get(n, (splits, Bp), torch.float32)
        nbpart = lambda n: ws.get(n, (nbs, Bp), torch.float32)
        w_pad = ws.get("dec_w_row", (Bp,), torch.float32, zero=True)
        w_pad[:B].copy_(w_row)
        grads_f32 = bool(train and nsplit == 3)
        gdt, gname, gplanes = (torch.int16, "split", 2) if grads_f32 else (torch.int16, "bf16", 1)   # fp32 mode: bf16 hi plane + lo plane
        if train:
            dL = ws.get("dec_dL_" + gname, (gplanes * Bp, Gp), gdt, zero=True)
            tP = ws.get("dec_tP_" + gname, (gplanes * Bp, Gp), gdt, zero=True)
            tS = ws.get("dec_tS_" + gname, (gplanes * Bp, Gp), gdt, zero=True)
            dth = ws.get("dec_dtheta", (Bp // 64, Gp), torch.float32, zero=True)
        else:
            dL = tP = tS = dth = None
        lse_p, lse_s, a_p, a_s = vec("dec_lse_p"), vec("dec_lse_s"), vec("dec_a_p"), vec("dec_a_s")
        # mixing logits [Bp][Gp]: f16 in bf16 mode, fp32 in fp32 mode (one plain MFMA GEMM, K = KMp)
        logits = ws.get("dec_logits_" + ("f32" if mlo else "f16"), (Bp, Gp), torch.float32 if mlo else torch.float16)
        _abi.call("spv_dec_logits", ptr(Am_hi), ptr(Am_lo), ptr(Wm_hi), ptr(Wm_lo), KMp, Bp, Gp, nsplit, ptr(logits), int(mlo), stream_ptr())
        cst = counts.c_struct(rows)
        P = SpvDecParams(
            X=cst.X, ldx=cst.ld, rows=cst.rows, col_off=cst.col_off, count_is_u16=int(cst.dtype == _abi.SPV_COUNT_U16),
            B=B, G=G, Bp=Bp, Gp=Gp, logits=ptr(logits), n_gene_tiles=Gp // 32, logits_f32=int(mlo), Wps_hi=ptr(Wps_hi), Wps_lo=ptr(Wps_lo), Aps_hi=ptr(Aps_hi), Aps_lo=ptr(Aps_lo),
            gene_tab=ptr(gene_tab), cnt_tab=ptr(cnt_tab), a_p=ptr(a_p), a_s=ptr(a_s), lse_p=ptr(lse_p), lse_s=ptr(lse_s),
            w_row=ptr(w_pad), gene_splits=splits, genes_per_split=per,
            part_max_p=ptr(part("dec_pmp")), part_sum_p=ptr(part("dec_psp")), part_max_s=ptr(part("dec_pms")), part_sum_s=ptr(part("dec_pss")),
            rec_part=ptr(nbpart("dec_rec")), tp_part=ptr(nbpart("dec_tp")), ts_part=ptr(nbpart("dec_ts")),
            dtheta_part=ptr(dth), dL=ptr(dL), tP=ptr(tP), tS=ptr(tS), grads_f32=int(grads_f32), nb_splits=nbs, nb_genes_per_split=nbper,
        )
        _abi.call("spv_dec_lse", C.byref(P), ptr(f32(library)), stream_ptr())
        _abi.call("spv_dec_nb_fwd", C.byref(P), int(train), stream_ptr())
        rec = nbpart("dec_rec").sum(0)[:B]
        loss = (rec * w_row).sum()
        if train:
            ctx.P, ctx.ws, ctx.nsplit, ctx.dims = P, ws, nsplit, (B, G, Bp, Gp, n_p, n_s, KM, KMp, m.shape[1])
            ctx.keep = (Wm_hi, Wm_lo, Am_hi, Am_lo, Wps_hi, Wps_lo, Aps_hi, Aps_lo, dL, tP, tS, dth, lse_p, lse_s, gene_tab)
            ctx.Tp, ctx.Ts = nbpart("dec_tp").sum(0), nbpart("dec_ts").sum(0)
            ctx.grads_f32, ctx.done = grads_f32, False
            ctx.save_for_backward(px_r)
        ctx.mark_non_differentiable(rec)
        return loss, rec

    @staticmethod
    def backward(ctx, g_loss, _g_rec):
        if ctx.done:
            raise _abi.SpvError("DecoderNBLoss.backward may run once per forward (gradient buffers are consumed in place)")
        ctx.done = True
        (px_r,) = ctx.saved_tensors
        B, G, Bp, Gp, n_p, n_s, KM, KMp, n_m = ctx.dims
        ws, nsplit, P = ctx.ws, ctx.nsplit, ctx.P
        Wm_hi, Wm_lo, Am_hi, Am_lo, Wps_hi, Wps_lo, Aps_hi, Aps_lo, dL, tP, tS, dth, lse_p, lse_s, gene_tab = ctx.keep
        _abi.call("spv_dec_softmax_bwd", C.byref(P), ptr(ctx.Tp), ptr(ctx.Ts), None, stream_ptr())
        if ctx.grads_f32:  # fp32 mode: the arrays already are the hi / lo operand images of the split-bf16 GEMMs
            (dL_hi, dL_lo), (tP_hi, tP_lo), (tS_hi, tS_lo) = (dL[:Bp], dL[Bp:]), (tP[:Bp], tP[Bp:]), (tS[:Bp], tS[Bp:])
        else:
            dL_hi, tP_hi, tS_hi, dL_lo, tP_lo, tS_lo = dL, tP, tS, None, None, None
        ksp = max(1, min(8, (Gp // 32) // 16))  # K splits of the contractions over genes
        # contraction over cells:  d W[g][k] = sum_b dY[b][g] * A[b][k]
        csp = max(1, min(4, (Bp // 32) // 16))  # K splits of the contractions over cells
        dWm = _gemm(True, dL_hi, dL_lo, Gp, Am_hi, Am_lo, KMp, G, KMp, Bp, nsplit, min(csp, 2), ws, "dec_dWm", a_tiles=Gp // 32)
        dWp = _gemm(True, tP_hi, tP_lo, Gp, Aps_hi, Aps_lo, DEC_KPS, G, DEC_KP, Bp, nsplit, csp, ws, "dec_dWp", a_tiles=Gp // 32)
        dWs = _gemm(True, tS_hi, tS_lo, Gp, Aps_hi, Aps_lo, DEC_KPS, G, DEC_KS, Bp, nsplit, csp, ws, "dec_dWs", b_col_off=DEC_KP, a_tiles=Gp // 32)
        # contraction over genes:  d A[b][k] = sum_g dY[b][g] * W[g][k]
        dAm = _gemm(False, dL_hi, dL_lo, Gp, Wm_hi, Wm_lo, KMp, B, KMp, G, nsplit, ksp, ws, "dec_dAm", a_tiles=Gp // 32)
        dAp = _gemm(False, tP_hi, tP_lo, Gp, Wps_hi, Wps_lo, DEC_KPS, B, DEC_KP, G, nsplit, ksp, ws, "dec_dAp", a_tiles=Gp // 32)
        dAs = _gemm(False, tS_hi, tS_lo, Gp, Wps_hi, Wps_lo, DEC_KPS, B, DEC_KS, G, nsplit, ksp, ws, "dec_dAs", b_col_off=DEC_KP, a_tiles=Gp // 32)
        g = g_loss
        for t in (dWm, dWp, dWs, dAm, dAp, dAs):  # scale the contiguous GEMM outputs once, slice afterwards
            t.mul_(g)
        d_m = dAm[:, :n_m]
        d_zp = dAm[:, n_m:n_m + n_p] + dAp[:, :n_p]
        d_zs = dAm[:, n_m + n_p:n_m + n_p + n_s] + dAs[:, :n_s]
        d_Wm, d_bm = dWm[:, :KM - 1], dWm[:, KM - 1]
        d_Wp, d_cp = dWp[:, :n_p], dWp[:, n_p]
        d_Ws, d_cs = dWs[:, :n_s], dWs[:, n_s]
        d_pxr = torch.exp(px_r) * dth.sum(0)[:G] * g  # theta = exp(px_r): d/d px_r = theta * d/d theta
        return (None, None, None, d_zp, d_zs, d_m, d_Wp, d_cp, d_Ws, d_cs, d_Wm, d_bm, d_pxr, None, None, None, None, None)
