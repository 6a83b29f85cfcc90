"""Host side of the HIP hot path: operand packing, workspaces and autograd wrappers.

Every function here calls into libspvipes_hip.so through the C ABI (spvipes_amd/_abi.py).  There
is no CPU / eager-PyTorch implementation of these ops in this package: if the library is missing
or a tensor is not resident on the GPU the call raises.

Reference arithmetic replaced (file:line into /root/reference/src/spVIPES):
    EncoderFC1      module/spVIPESmodule.py:428-435 (slice, log1p, library) + nn/networks.py:119 (fc1+relu)
    DecoderNBLoss   nn/networks.py:314-325 (rates, mixing logits) + module/spVIPESmodule.py:758-759,817-824
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch

from . import _abi
from ._abi import DEC_CELLS_PER_WG, DEC_KP, DEC_KPS, DEC_KS, NB_CMAX, SpvCounts, SpvDecParams, check, ptr, round_up, stream_ptr

N_HIDDEN_MIX = 256  # LinearDecoderSPVIPE's own n_hidden default (nn/networks.py:194)


@dataclass
class GroupCounts:
    """Count matrix of one group resident in HBM: X[cell][ld] (float32 or uint16 bit patterns),
    the group's own genes are columns [col_off, col_off + G)."""

    X: torch.Tensor
    G: int
    col_off: int = 0
    resident: bool = False   # the matrix stays in HBM across steps (data sets, not per-call minibatches): derived images are worth building

    def __post_init__(self):
        self._xb = None
        if not self.X.is_cuda or self.X.dim() != 2 or not self.X.is_contiguous():
            raise _abi.SpvError("GroupCounts needs a contiguous 2-D tensor resident in HBM")
        if self.X.dtype == torch.float32:
            self.dtype = _abi.SPV_COUNT_F32
        elif self.X.dtype in (torch.int16, torch.uint16):
            self.dtype = _abi.SPV_COUNT_U16
        else:
            raise _abi.SpvError(f"unsupported count dtype {self.X.dtype} (float32 or uint16)")
        if self.col_off < 0 or self.col_off + self.G > self.X.shape[1]:
            raise _abi.SpvError("gene column range outside the count matrix")

    @property
    def n_cells(self) -> int:
        return self.X.shape[0]

    def log1p_image(self):
        """(bf16 log1p(x) of every cell [n_cells][ld >= round_up(G, 96), multiple of 128], library = log(sum_g log1p(x)) per cell), built on first use
        by spv_prepare_log1p: the same arithmetic the reference does per minibatch (module/spVIPESmodule.py:428-435)."""
        if self._xb is None:
            ld = round_up(round_up(self.G, 96), 128)   # (96: the gene tile of the fc1 weight-gradient GEMM)
            xb = torch.empty((self.n_cells, ld), dtype=torch.int16, device=self.X.device)
            lib = torch.empty((self.n_cells,), dtype=torch.float32, device=self.X.device)
            cs = self.c_struct(None)
            _abi.call("spv_prepare_log1p", C.byref(cs), self.n_cells, self.G, ptr(xb), ld, ptr(lib), stream_ptr())
            self._xb = (xb, lib)
        return self._xb

    def c_struct(self, rows: Optional[torch.Tensor]) -> SpvCounts:
        if rows is not None and (rows.dtype != torch.int32 or not rows.is_cuda or not rows.is_contiguous()):
            raise _abi.SpvError("row index must be a contiguous int32 tensor in HBM")
        return SpvCounts(ptr(self.X), self.X.shape[1], ptr(rows), self.col_off, self.dtype)


class Workspace:
    """Named device buffers reused across steps (no allocation on the hot path after warm-up).
    Buffers that kernels rely on being zero outside the written region are created zero-filled."""

    def __init__(self, device):
        self.device = device
        self._buf: Dict[Tuple, torch.Tensor] = {}

    def get(self, name: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        key = (name, tuple(shape), dtype)
        t = self._buf.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
            self._buf[key] = t
        return t


_SIDE_STREAMS: Dict[Tuple, "torch.cuda.Stream"] = {}
SERIAL_STREAMS = os.environ.get("SPV_SERIAL_STREAMS", "0") == "1"


def group_streams(device, n: int = 2):
    """[current stream, side stream, ...]: the two groups' launch chains are independent for long stretches of the step
    (fc1, operand packing, logits / lse / likelihood, the backward GEMMs) and most of those kernels leave CUs idle on
    their own, so group 1 runs on a side stream: fork with ``fork(streams)``, join with ``join(streams)``.  Under
    hipGraph capture the fork/join become parallel branches of the graph."""
    dev = torch.device(device)
    out = [torch.cuda.current_stream(dev)]
    if SERIAL_STREAMS:  # every chain on the caller's stream: kernels run one at a time (per-kernel timing, debugging)
        return out * n
    for i in range(1, n):
        key = (dev.index, i)
        if key not in _SIDE_STREAMS:
            _SIDE_STREAMS[key] = torch.cuda.Stream(device=dev)
        out.append(_SIDE_STREAMS[key])
    return out


def fork(streams) -> None:
    for st in streams[1:]:
        st.wait_stream(streams[0])


def join(streams) -> None:
    for st in streams[1:]:
        streams[0].wait_stream(st)


# Work that nothing later in the backward pass depends on (the mixture-weight gradient GEMMs) is queued on a side stream
# that starts when the GPU is about to run the long chain of tiny encoder / PoE backward kernels, and is joined as late as
# possible.  With DEFER_JOIN False the join happens before DecoderFused.backward returns (any caller may then read .grad);
# train.Trainer sets it True around its backward pass and joins with ``join_pending`` right after it.
DEFER_JOIN = False
OVERLAP_SMALL = os.environ.get("SPV_OVERLAP_SMALL", "1") != "0"  # side-stream overlap of independent small-kernel groups
STAGGER = os.environ.get("SPV_STAGGER", "1") != "0"  # group 1 orders its independent kernels differently from group 0
FUSED_DZ = os.environ.get("SPV_FUSED_DZ", "1") != "0"  # softmax fix also produces the latent gradient of the rate heads
FUSED_PACK = os.environ.get("SPV_FUSED_PACK", "1") != "0"  # latent / trunk kernels also write the decoder's bf16 operand images
DEFER_WM = os.environ.get("SPV_DEFER_WM", "1") != "0"  # mixture-weight gradient GEMMs on the late side stream
DEFER_BC = os.environ.get("SPV_DEFER_BC", "1") != "0"  # regressor weight-gradient GEMMs beside the trunk backward (side stream)
_PENDING: list = []
_PENDING_KEEP: list = []


def defer(stream, keep=()) -> None:
    """register side-stream work (and the tensors it reads) to be joined by ``join_pending``"""
    if stream not in _PENDING:
        _PENDING.append(stream)
    _PENDING_KEEP.extend(keep)


def join_pending(device=None) -> None:
    while _PENDING:
        st = _PENDING.pop()
        torch.cuda.current_stream(st.device if device is None else device).wait_stream(st)
    _PENDING_KEEP.clear()


def _pack(src: torch.Tensor, dst_hi: torch.Tensor, dst_lo: Optional[torch.Tensor], *, extra_col: Optional[torch.Tensor] = None,
          extra_one: bool = False, dst_row_off: int = 0, dst_col_off: int = 0, rows_cover: Optional[int] = None,
          cslot: Optional[int] = None) -> None:
    """fp32 [R][C] -> bf16 hi/lo block of a padded image (see spv_pack_bf16)."""
    if src.dtype != torch.float32 or src.dim() != 2 or src.stride(1) != 1:
        raise _abi.SpvError("pack source must be fp32 [R][C] with unit column stride")
    R, Cc = src.shape
    ld_dst = dst_hi.shape[1]
    Rp = rows_cover if rows_cover is not None else dst_hi.shape[0] - dst_row_off
    cs = cslot if cslot is not None else ld_dst - dst_col_off
    off = dst_row_off * ld_dst * 2
    _abi.call("spv_pack_bf16", ptr(src), src.stride(0), R, Cc, ptr(extra_col), int(extra_one), ptr(dst_hi) + off,
                            (ptr(dst_lo) + off) if dst_lo is not None else None, ld_dst, dst_col_off, Rp, cs, stream_ptr())


def _bf16_image(ws: Workspace, name: str, rows: int, cols: int, lo: bool):
    hi = ws.get(name + "_hi", (rows, cols), torch.int16)
    return hi, (ws.get(name + "_lo", (rows, cols), torch.int16) if lo else None)


# ------------------------------------------------------------------------------------------------
# encoder fc1
# ------------------------------------------------------------------------------------------------
def _fc1_splits(B: int, G: int, N1: int) -> int:
    bm = 128 if N1 <= 32 else 64
    tiles = -(-B // bm) * -(-N1 // (32 if N1 <= 32 else (128 if N1 <= 128 else 256)))
    ktiles = -(-G // 32)
    want = max(1, 512 // max(tiles, 1))
    return max(1, min(want, ktiles // 8 if ktiles >= 8 else 1, 16))


class EncoderFC1(torch.autograd.Function):
    """h1 = relu(log1p(X[rows, genes]) @ [W_private; W_shared]^T + b), library = log(sum log1p(x))."""

    @staticmethod
    def forward(ctx, counts: GroupCounts, rows, B: int, w_priv, b_priv, w_sh, b_sh, nsplit: int, ws: Workspace):
        ctx.set_materialize_grads(False)
        H, G = w_priv.shape
        N1 = 2 * H
        bn = 32 if N1 <= 32 else (128 if N1 <= 128 else 256)
        N1p, Gp = round_up(N1, bn), round_up(G, 64)
        W_hi, W_lo = _bf16_image(ws, "fc1_W", N1p, Gp, nsplit == 3)
        _pack(w_priv, W_hi, W_lo, dst_row_off=0, rows_cover=H)
        _pack(w_sh, W_hi, W_lo, dst_row_off=H, rows_cover=N1p - H)
        f32c = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
        b_priv, b_sh = f32c(b_priv), f32c(b_sh)
        splits = _fc1_splits(B, G, N1)
        slabs = ws.get("fc1_slabs", (splits, B, N1), torch.float32)
        rowsum = ws.get("fc1_rowsum", (splits, B), torch.float32)
        h1 = torch.empty((B, N1), dtype=torch.float32, device=w_priv.device)
        library = torch.empty((B,), dtype=torch.float32, device=w_priv.device)
        cs = counts.c_struct(rows)
        # bf16 mode on a resident count matrix: log1p(x) of the whole data set sits in HBM as bf16 (built once, see
        # GroupCounts.log1p_image) and both fc1 GEMMs gather plain rows from it instead of decoding counts every step
        xb, ld_xb, lib_all = None, 0, None
        if nsplit == 1 and counts.resident:
            xb, lib_all = counts.log1p_image()
            ld_xb = xb.shape[1]
        _abi.call("spv_enc_fc1_fwd", C.byref(cs), B, G, ptr(W_hi), ptr(W_lo), Gp, N1, ptr(b_priv), ptr(b_sh), H, nsplit, splits, ptr(slabs),
                                  ptr(rowsum), ptr(h1), ptr(library), ptr(xb), ld_xb, ptr(lib_all), stream_ptr())
        ctx.counts, ctx.rows, ctx.B, ctx.nsplit, ctx.ws, ctx.H, ctx.G = counts, rows, B, nsplit, ws, H, G
        ctx.xb, ctx.ld_xb = xb, ld_xb
        ctx.save_for_backward(h1, w_priv, b_priv, w_sh, b_sh)
        ctx.mark_non_differentiable(library)
        return h1, library

    @staticmethod
    def backward(ctx, dh1, _dlib):
        from .nn_ops import grad_out

        if dh1 is None:
            return (None,) * 9
        h1, w_priv, b_priv, w_sh, b_sh = ctx.saved_tensors
        B, H, G, ws, nsplit = ctx.B, ctx.H, ctx.G, ctx.ws, ctx.nsplit
        N1 = 2 * H
        Bp, N1p = round_up(B, 64), round_up(N1, 128)
        dh1 = dh1 if (dh1.dtype == torch.float32 and dh1.is_contiguous()) else dh1.float().contiguous()
        dh_hi, dh_lo = _bf16_image(ws, "fc1_dh", Bp, N1p, nsplit == 3)
        part = ws.get("fc1_db_part", (Bp // 16, N1), torch.float32)
        # (kernel target, autograd return): with the trainer's gradient sink the kernels write straight into .grad
        (dWp, rWp), (dbp, rbp), (dWs, rWs), (dbs, rbs) = grad_out(w_priv), grad_out(b_priv), grad_out(w_sh), grad_out(b_sh)
        _abi.call("spv_enc_fc1_bwd_prep", ptr(dh1), ptr(h1), B, N1, ptr(dh_hi), ptr(dh_lo), N1p, Bp, ptr(part), ptr(dbp), ptr(dbs), H, stream_ptr())
        cs = ctx.counts.c_struct(ctx.rows)
        _abi.call("spv_enc_fc1_wgrad", C.byref(cs), B, G, ptr(dh_hi), ptr(dh_lo), N1p, N1, nsplit, ptr(dWp), ptr(dWs), H, G,
                  ptr(ctx.xb), ctx.ld_xb, stream_ptr())
        return None, None, None, rWp, rbp, rWs, rbs, None, None


# ------------------------------------------------------------------------------------------------
# decoder + NB-mixture likelihood
# ------------------------------------------------------------------------------------------------
def _gene_splits(Bp: int, Gp: int) -> Tuple[int, int]:
    cell_blocks = Bp // DEC_CELLS_PER_WG
    want = max(1, -(-1024 // cell_blocks))  # ~4 workgroups per CU
    tiles = Gp // 32
    splits = max(1, min(want, tiles))
    per = -(-tiles // splits) * 32
    splits = -(-Gp // per)
    return splits, per


NB_GSPL_MAX = 160  # genes per likelihood split (their regressor weights sit in LDS)


def _nb_splits(Gp: int) -> Tuple[int, int]:
    per = min(NB_GSPL_MAX, Gp)
    return -(-Gp // per), per


def _gemm_slabs(a_kmajor: bool, A_hi, A_lo, lda, B_hi, B_lo, ldb, M, N, K, nsplit, splits, ws: Workspace, name: str,
                b_col_off: int = 0, a_tiles: int = 0) -> torch.Tensor:
    """split-K GEMM leaving its ``splits`` fp32 partial slabs [splits][M][N] for spv_reduce_slabs"""
    out = ws.get(name, (splits, M, N), torch.float32)
    b_off = b_col_off * 2
    _abi.call("spv_gemm_bf16", int(a_kmajor), ptr(A_hi), ptr(A_lo), lda, ptr(B_hi) + b_off, (ptr(B_lo) + b_off) if B_lo is not None else None,
                            ldb, ptr(out), N, M, N, K, nsplit, splits, M * N, a_tiles, stream_ptr())
    return out


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
