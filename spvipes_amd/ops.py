"""Host side of the HIP hot path: operand packing, workspaces and autograd wrappers.

Every function here calls into libspvipes_hip.so through the C ABI (spvipes_amd/_abi.py).  There
is no CPU / eager-PyTorch implementation of these ops in this package: if the library is missing
or a tensor is not resident on the GPU the call raises.

Reference arithmetic replaced (file:line into /root/reference/src/spVIPES):
    EncoderFC1      module/spVIPESmodule.py:428-435 (slice, log1p, library) + nn/networks.py:119 (fc1+relu)
    (the decoder + likelihood side lives in dec_ops.DecoderFused)
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
        """(f16 log1p(x) of every cell [n_cells][ld >= round_up(G, 96), multiple of 128], library = log(sum_g log1p(x)) per cell), built on first use
        by spv_prepare_log1p: the same arithmetic the reference does per minibatch (module/spVIPESmodule.py:428-435)."""
        if self._xb is None:
            ld = round_up(round_up(self.G, 96), 128)   # (96: the gene tile of the fc1 weight-gradient GEMM)
            xb = torch.empty((self.n_cells, ld), dtype=torch.int16, device=self.X.device)
            lib = torch.empty((self.n_cells,), dtype=torch.float32, device=self.X.device)
            cs = self.c_struct(None)
            _abi.call("spv_prepare_log1p", C.byref(cs), self.n_cells, self.G, ptr(xb), ld, ptr(lib), stream_ptr())
            self._xb = (xb, lib)
        return self._xb

    def log1p_image_split(self):
        """"fp32" mode: (bf16 [hi | lo] halves of log1p(x) per row [n_cells][2 ldh], library), built on first use by spv_prepare_log1p_split"""
        if getattr(self, "_xb_split", None) is None:
            ldh = round_up(round_up(self.G, 96), 128)
            xb = torch.empty((self.n_cells, 2 * ldh), dtype=torch.int16, device=self.X.device)
            lib = torch.empty((self.n_cells,), dtype=torch.float32, device=self.X.device)
            cs = self.c_struct(None)
            _abi.call("spv_prepare_log1p_split", C.byref(cs), self.n_cells, self.G, ptr(xb), 2 * ldh, ptr(lib), stream_ptr())
            self._xb_split = (xb, lib)
        return self._xb_split

    def fc1_image(self, nsplit: int, B: int, N1: int, Gp: int):
        """(image, row pitch, library table) of the resident log1p image the fc1 kernels gather from, or (None, 0, None): the f16 image in
        the one-MFMA mode; in "fp32" mode the [hi | lo] image when the LDS-DMA kernel takes the shape (it is the only consumer)."""
        if not self.resident:
            return None, 0, None
        if nsplit == 1:
            xb, lib = self.log1p_image()
            return xb, xb.shape[1], lib
        ld = 2 * round_up(round_up(self.G, 96), 128)
        if not _abi.load().spv_enc_fc1_fwd_uses_dma(B, self.G, N1, nsplit, 1, Gp, ld):
            return None, 0, None
        xb, lib = self.log1p_image_split()
        return xb, ld, lib

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
        # packed bf16 weight images that are known to equal their fp32 parameters: name -> image_token(parameters).  Only
        # train.Trainer writes here, after its Adam launch has rewritten the images (spv_adam_step_images) or after an explicit
        # repack; a forward pass that finds the token of the CURRENT parameters skips its spv_pack_bf16 launches.
        self.fresh: Dict[str, Tuple] = {}

    def get(self, name: str, shape, dtype, zero: bool = False) -> torch.Tensor:
        key = (name, tuple(shape), dtype)
        t = self._buf.get(key)
        if t is None:
            t = (torch.zeros if zero else torch.empty)(tuple(shape), dtype=dtype, device=self.device)
            self._buf[key] = t
        return t


def image_token(*tensors) -> Tuple:
    """identity + in-place version of the parameters an image was built from (any torch-side write to a parameter bumps
    ``_version``; the HIP Adam kernel does not, and it is the one writer that keeps the images in step itself)"""
    return tuple((t.data_ptr(), t._version, tuple(t.shape)) for t in tensors)


_SIDE_STREAMS: Dict[Tuple, "torch.cuda.Stream"] = {}
SERIAL_STREAMS = os.environ.get("SPV_SERIAL_STREAMS", "auto") == "1"   # every launch on the caller's stream
# "Small" steps -- at most SMALL_STEP_ELEMS cells x genes per group and minibatch, two groups -- are launch-bound: every kernel is mostly
# fixed cost, a stream fork / join costs more than the overlap it buys, and one grid for both groups per decoder kernel (DEC_PAIR below) saves
# a launch's worth each.  Unless SPV_SERIAL_STREAMS / SPV_DEC_PAIR say otherwise such steps run on ONE stream with pair grids.  Same box,
# ms/step, pair grids + one stream / shipped before (two streams, per-group launches): G 2000, H 64 (BASELINE configs[0] shape): B 128 0.390 /
# 0.452, B 512 0.433 / 0.497, B 1024 0.519 / 0.607; G 10 000, H 128: B 128 0.532 / 0.552, B 512 0.625 / 0.620 (the threshold), B 4096
# (pair grids, two streams) 1.349 / 1.318.  One stream WITHOUT the pair grids is slower than two streams at G 10 000 (0.606 vs 0.552).
SERIAL_AUTO = os.environ.get("SPV_SERIAL_STREAMS", "auto") == "auto"
SMALL_STEP_ELEMS = 4_000_000
_SMALL_STEP = False


def set_step_shape(B: int, G_max: int, n_groups: int) -> None:
    """called by spVIPESmodule at the top of a step: is this step a "small" one (above)?"""
    global _SMALL_STEP
    _SMALL_STEP = bool(n_groups == 2 and B * G_max <= SMALL_STEP_ELEMS)


def serial_streams() -> bool:
    return SERIAL_STREAMS or (SERIAL_AUTO and _SMALL_STEP and DEC_PAIR != "0")


def group_streams(device, n: int = 2):
    """[current stream, side stream, ...]: the two groups' launch chains are independent for long stretches of the step
    (fc1, operand packing, logits / lse / likelihood, the backward GEMMs) and most of those kernels leave CUs idle on
    their own, so group 1 runs on a side stream: fork with ``fork(streams)``, join with ``join(streams)``.  Under
    hipGraph capture the fork/join become parallel branches of the graph."""
    dev = torch.device(device)
    out = [torch.cuda.current_stream(dev)]
    if serial_streams():  # every chain on the caller's stream: kernels run one at a time (per-kernel timing, debugging; small steps)
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
STAGGER_BWD = os.environ.get("SPV_STAGGER_BWD", "0") != "0"  # backward: group 1 runs its d A_m GEMM ahead of its softmax fix (A/B on MI355X: 1.703 vs 1.693 ms, noise -> off)
DA_FIRST = int(os.environ.get("SPV_DA_FIRST", "1"))  # backward: d A_m GEMMs first, softmax fixes on the side stream (1: side starts after them, 2: with them); A/B with the half-chip d A_m launches: -9 / -3 us on two boxes
# mixture-weight GEMMs held back until the BatchNorm-fold backward (beside the tiny-kernel tail) instead of beside the trunk backward, which they
# slow down.  Same box, ms/step, held back / not: C3 (G 20 000) 2.118 2.122 / 2.157 2.152; C2 1.307 1.316 1.318 1.343 1.313 1.320 / 1.323 1.314
# 1.312 1.317 1.313 1.316; C5 4.941 4.965 4.946 5.022 5.003 / 4.981 4.904 4.921 5.005 5.001; C2 split words 2.125 2.133 / 2.122 2.115; C4 3.618
# 3.570 / 3.604 3.595.  "auto": only where it paid -- 16-bit gradient words and G >= WM_LATE_MIN_G (the GEMMs grow with G, the tail does not).
WM_LATE = os.environ.get("SPV_WM_LATE", "auto")
WM_LATE_MIN_G = 16384


def wm_late_for(G_max: int, grads_f32: bool) -> bool:
    if WM_LATE == "auto":
        return (not grads_f32) and G_max >= WM_LATE_MIN_G
    return WM_LATE != "0"



FC1_GROUPED = os.environ.get("SPV_FC1_GROUPED", "1") != "0"  # both groups' fc1 GEMMs as one launch per kernel (EncoderFC1Grouped) instead of two streams
# label pairing on a side stream: 1 beside the fc1 GEMMs, 2 beside the encoder tails (module.inference).  Same-box A/B at C2, three rounds:
# 2: 1.3218 / 1.3205 / 1.3154 ms, 0: 1.3229 / 1.3236 / 1.3375.  At C1 (B 128, where the pairing is a 3 us kernel and the fork + join costs more
# than it hides): 0: 0.4578 / 0.4574 / 0.4567, 2: 0.4614 / 0.4654 / 0.4645, 1: 0.4658 / 0.4652 / 0.4654.  "auto" (-1) = 2 from LABEL_PRE_MIN_B
# cells per minibatch, else 0.
LABEL_PRE = int(os.environ.get("SPV_LABEL_PRE", "-1"))
LABEL_PRE_MIN_B = 1024


def label_pre_for(B: int) -> int:
    return LABEL_PRE if LABEL_PRE >= 0 else (2 if B >= LABEL_PRE_MIN_B else 0)



# the decoder's per-group launches (tables, logits GEMM, softmax statistics, likelihood; d A_m / d W_m GEMMs, one-pass backward) as ONE grid
# per kernel for the two groups of a step (spv_dec_*_grouped, spv_gemm_bf16_grouped; bit-identical results) instead of one launch per group
# on two streams.  "auto": in small steps (set_step_shape above); "1" / "0": always / never.
DEC_PAIR = os.environ.get("SPV_DEC_PAIR", "auto")


def dec_pair_for(B: int, n_groups: int) -> bool:
    if n_groups != 2 or DEC_PAIR == "0":
        return False
    return True if DEC_PAIR == "1" else _SMALL_STEP


FC1_PAIR_SPLITS = os.environ.get("SPV_FC1_PAIR_SPLITS", "1") != "0"  # grouped fc1 forward: K splits sized for the pair's shared grid
DEC_PAIR_SPLITS = os.environ.get("SPV_DEC_PAIR_SPLITS", "1") != "0"  # d A_m GEMMs of the two groups: K splits sized so that the pair shares one round
HEADS_DMA = os.environ.get("SPV_HEADS_DMA", "1") != "0"  # both regressor weight gradients in one LDS-DMA pass (spv_dec_heads_wgrad)
# bf16 mode: the softmax fix, the latent gradient AND the regressor weight gradients from ONE read-only pass over t_P / t_S
# (spv_dec_heads_bwd): no write-back of the corrected arrays, no second pass over them
FUSED_HEADS = os.environ.get("SPV_FUSED_HEADS", "1") != "0"
# workgroups the softmax-statistics / softmax-fix kernels are split into (cell blocks x gene splits).  512 = ONE round of two workgroups per CU, and half the
# per-split partials (statistics, latent gradient) to write and re-read.  Same-box A/B 1024 / 768 / 512 / 384 / 256 at C2: 1.440 / 1.438 / 1.412 / 1.417 / 1.426 ms;
# C3 2.39 -> 2.36, C4 3.90 -> 3.88, C5 8.29 -> 8.18, C1 unchanged
GSPLIT_WANT = int(os.environ.get("SPV_GSPLIT_WANT", "512"))
FUSED_HEADS_F32 = os.environ.get("SPV_FUSED_HEADS_F32", "1") != "0"   # ... and in "fp32" mode on the hi / lo planes of the split gradient words (spv_dec_heads_bwd, grads_f32)
FUSED_DZ = os.environ.get("SPV_FUSED_DZ", "1") != "0"  # softmax fix also produces the latent gradient of the rate heads
FUSED_DZ_F32 = os.environ.get("SPV_FUSED_DZ_F32", "1") != "0"   # "fp32" mode: the latent gradient of the rate heads comes out of the softmax-fix pass too (split-bf16 contraction)
# d A_m GEMM sums its split-K slabs inside the launch (spv_gemm_bf16_fix: the last-arriving tile of a row block adds the slabs) instead of
# a reduction launch on the critical chain.  Measured and NOT the default: the finisher of a 128 x 320 tile re-reads splits x 160 KB
# (600 KB at c2) at the tail of a short launch while the other CUs are already idle, and that costs more than the 1.5 us boundary plus the
# chip-wide reduction it replaces.  Same box, ms/step, with / without: c2 bf16 1.3268 1.3029 1.3165 / 1.2873 1.2840 1.2979;
# c2 split words 2.214 2.203 2.208 / 2.193 2.184 2.172; c5 5.018 5.016 / 5.031 5.019 (first box: 5.032 / 5.100); c4 3.761 3.789 /
# 3.797 3.793.  "1" = on wherever the LDS-DMA kernels run, "fp32" = split-word mode only.  (tests/test_gpu_gemm_fixup.py keeps the
# entry point bit-exact against the slab sum.)
DA_FIXUP = os.environ.get("SPV_DA_FIXUP", "0")
TRUNK_FOLD = os.environ.get("SPV_TRUNK_FOLD", "1") != "0"  # the mixing trunk's BatchNorm folded into its Linear (spv_trunk_fold_fwd / _bwd)
FUSED_PACK = os.environ.get("SPV_FUSED_PACK", "1") != "0"  # latent / trunk kernels also write the decoder's bf16 operand images
DEFER_WM = os.environ.get("SPV_DEFER_WM", "1") != "0"  # mixture-weight gradient GEMMs on the late side stream
DEFER_BC = os.environ.get("SPV_DEFER_BC", "1") != "0"  # regressor weight-gradient GEMMs beside the trunk backward (side stream)
# critical chain runs the read-only latent-gradient pass (spv_dec_dz), the in-place softmax fix moves beside it on the side stream.
# Measured at C2, same box, alternating (round 2): OFF 1.690 / 1.690 ms per step, ON 1.745 / 1.737 ms -- the second pass over the two
# gradient arrays (2 x 168 MB more HBM reads per step) costs more than the shorter critical chain gains.  Kept as a switch, off.
DZ_ONLY = os.environ.get("SPV_DZ_ONLY", "0") != "0"
# the two groups' decoder-forward chains (logits GEMM, softmax statistics, likelihood) / d A_m GEMMs on two streams (parallel graph
# branches) or one after the other on the caller's stream (every fork / join of graph branches leaves the GPU idle for >= 10 us)
FWD_GROUP_STREAMS = os.environ.get("SPV_FWD_GROUP_STREAMS", "1") != "0"
BWD_GROUP_STREAMS = os.environ.get("SPV_BWD_GROUP_STREAMS", "1") != "0"
_PENDING: list = []
_PENDING_KEEP: list = []


def defer(stream, keep=()) -> None:
    """register side-stream work (and the tensors it reads) to be joined by ``join_pending``"""
    if stream not in _PENDING:
        _PENDING.append(stream)
    _PENDING_KEEP.extend(keep)


def join_all_side_streams(device) -> None:
    """Make the current stream wait for every side stream of this device.  Called at the end of a backward pass: autograd runs a
    node's backward on the stream its forward ran on (group g > 0: a side stream) and joins such a stream back only through the
    parameters' cached gradient-accumulator nodes, whose stream is whatever was current the FIRST time the parameter took part
    in a graph.  If that first step ran single-stream (bench.py's per-kernel event pass, SERIAL_STREAMS), nothing joins the
    side stream any more, and a later hipGraph capture ends with "capturing stream has unjoined work".  Joining explicitly makes
    the step independent of that history."""
    dev = torch.device(device)
    cur = torch.cuda.current_stream(dev)
    capturing = torch.cuda.is_current_stream_capturing()
    for (idx, _i), st in _SIDE_STREAMS.items():
        if idx != dev.index or st is cur:
            continue
        if capturing:   # only streams that have been forked into this capture may be waited for
            with torch.cuda.stream(st):
                if not torch.cuda.is_current_stream_capturing():
                    continue
        cur.wait_stream(st)


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


def fc1_w_scale() -> float:
    """power of two the f16 fc1 weight image is multiplied by (csrc/spv_common.h: SPV_FC1_W_SCALE)"""
    return float(_abi.load().spv_fc1_w_scale())


def _pack_fc1_weights(w_priv, w_sh, W_hi, W_lo, H: int) -> None:
    """[W_private; W_shared] -> the fc1 weight operand image: bf16 hi / lo in "fp32" mode (W_lo given), IEEE f16 words of
    W * fc1_w_scale() in the one-MFMA mode (nn/networks.py:119's weight, both encoders stacked)"""
    if W_lo is not None:
        _pack(w_priv, W_hi, W_lo, dst_row_off=0, rows_cover=H)
        _pack(w_sh, W_hi, W_lo, dst_row_off=H, rows_cover=W_hi.shape[0] - H)
        return
    ld = W_hi.shape[1]
    for w, r0, cover in ((w_priv, 0, H), (w_sh, H, W_hi.shape[0] - H)):
        if w.dtype != torch.float32 or w.dim() != 2 or w.stride(1) != 1:
            raise _abi.SpvError("pack source must be fp32 [R][C] with unit column stride")
        _abi.call("spv_pack_f16", ptr(w), w.stride(0), w.shape[0], w.shape[1], ptr(W_hi) + r0 * ld * 2, ld, cover, ld, fc1_w_scale(), stream_ptr())


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



def _wgrad_takes_image(nsplit: int, xb, ld_xb: int, B: int, G: int, N1: int) -> bool:
    """may the fc1 weight gradient gather from the resident image the forward pass used? (the f16 image: always; the "fp32" mode's split
    image: where its one consumer, the LDS-DMA kernel, takes the shape)"""
    if xb is None:
        return False
    return nsplit == 1 or bool(_abi.load().spv_enc_fc1_wgrad_split_uses_dma(B, G, N1, round_up(N1, 128), ld_xb))


def _fc1_cov_table(w_priv, w_sh, G: int):
    """The weight columns of the one-hot batch covariates that follow the G gene columns of both encoders' fc1 (nn/networks.py:68,
    105-119), as the table spv_enc_fc1_fwd adds before the ReLU: fp32 [n_batch][2H] (rows = batch codes).  None without covariates."""
    if w_priv.shape[1] == G:
        return None
    return torch.cat([w_priv[:, G:], w_sh[:, G:]], 0).t().contiguous().float()


def _fc1_cov_wgrad(dh1, h1, onehot, dWp, dWs, G: int, H: int, ws: "Workspace") -> None:
    """d W[:, G:] of both encoders' fc1 = (relu'(h1) * dh1)^T one_hot(batch): the small-layer weight-gradient kernel with the one-hot
    matrix as its input operand (spv_linear_wgrad masks dY by Y > 0 itself), copied into the covariate columns of the two gradients."""
    from .nn_ops import _add_lin, _lin_batch, _wgrad

    B, N1 = h1.shape
    nb = onehot.shape[1]
    dcov = torch.empty((N1, nb), dtype=torch.float32, device=h1.device)
    b = _lin_batch(B, relu=True)
    _add_lin(b, N=N1, K=nb, W=ptr(dcov), X=ptr(onehot), ldx=nb, Y=ptr(h1), ldy=N1, dY=ptr(dh1), lddy=N1, dW=ptr(dcov), db=None)
    _wgrad(b, ws)
    dWp[:, G:].copy_(dcov[:H])
    dWs[:, G:].copy_(dcov[H:])


class EncoderFC1(torch.autograd.Function):
    """h1 = relu(log1p(X[rows, genes]) @ [W_private; W_shared]^T + b), library = log(sum log1p(x)).
    ``cov``: None, or (batch codes int32 [B], one_hot fp32 [B][n_batch]) when the fc1 weights carry n_batch covariate columns."""

    @staticmethod
    def forward(ctx, counts: GroupCounts, rows, B: int, w_priv, b_priv, w_sh, b_sh, nsplit: int, ws: Workspace, cov=None):
        ctx.set_materialize_grads(False)
        H, G = w_priv.shape[0], counts.G
        if (w_priv.shape[1] != G) != (cov is not None):
            raise _abi.SpvError("fc1 weights with covariate columns need the minibatch's batch codes (and only they do)")
        N1 = 2 * H
        bn = 32 if N1 <= 32 else (128 if N1 <= 128 else 256)
        N1p, Gp = round_up(N1, bn), round_up(G, 64)
        W_hi, W_lo = _bf16_image(ws, "fc1_W", N1p, Gp, nsplit == 3)
        if W_lo is not None or ws.fresh.get("fc1_W") != image_token(w_priv, w_sh):   # (else: Adam has just rewritten the image)
            _pack_fc1_weights(w_priv[:, :G], w_sh[:, :G], W_hi, W_lo, H)
        f32c = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
        b_priv, b_sh = f32c(b_priv), f32c(b_sh)
        cov_tab = _fc1_cov_table(w_priv, w_sh, G)
        h1 = torch.empty((B, N1), dtype=torch.float32, device=w_priv.device)
        library = torch.empty((B,), dtype=torch.float32, device=w_priv.device)
        cs = counts.c_struct(rows)
        # bf16 mode on a resident count matrix: log1p(x) of the whole data set sits in HBM as bf16 (built once, see
        # GroupCounts.log1p_image) and both fc1 GEMMs gather plain rows from it instead of decoding counts every step
        xb, ld_xb, lib_all = counts.fc1_image(nsplit, B, N1, Gp)
        if _abi.load().spv_enc_fc1_fwd_uses_dma(B, G, N1, nsplit, int(xb is not None), Gp, ld_xb):
            # LDS-DMA kernel (csrc/spv_fc1.h): 128-cell tiles x all 256 columns, K split so that ~one workgroup lands on every CU
            mt = -(-B // 128)
            splits = max(1, min(16, 256 // mt, (-(-G // 64)) // 4))
            slabs = ws.get("fc1_slabs_tiled", (splits, mt * 128, N1), torch.float32)
        else:
            splits = _fc1_splits(B, G, N1)
            slabs = ws.get("fc1_slabs", (splits, B, N1), torch.float32)
        rowsum = ws.get("fc1_rowsum", (splits, B), torch.float32)
        _abi.call("spv_enc_fc1_fwd", C.byref(cs), B, G, ptr(W_hi), ptr(W_lo), Gp, N1, ptr(b_priv), ptr(b_sh), H, nsplit, splits, ptr(slabs),
                                  ptr(rowsum), ptr(h1), ptr(library), ptr(xb), ld_xb, ptr(lib_all), ptr(cov_tab), ptr(cov[0]) if cov else None, stream_ptr())
        ctx.counts, ctx.rows, ctx.B, ctx.nsplit, ctx.ws, ctx.H, ctx.G = counts, rows, B, nsplit, ws, H, G
        ctx.xb, ctx.ld_xb, ctx.cov = (xb, ld_xb, cov) if _wgrad_takes_image(nsplit, xb, ld_xb, B, G, N1) else (None, 0, cov)
        ctx.save_for_backward(h1, w_priv, b_priv, w_sh, b_sh)
        ctx.mark_non_differentiable(library)
        return h1, library

    @staticmethod
    def backward(ctx, dh1, _dlib):
        from .nn_ops import grad_out

        if dh1 is None:
            return (None,) * 10
        h1, w_priv, b_priv, w_sh, b_sh = ctx.saved_tensors
        B, H, G, ws, nsplit = ctx.B, ctx.H, ctx.G, ctx.ws, ctx.nsplit
        N1 = 2 * H
        Bp, N1p = round_up(B, 64), round_up(N1, 128)
        dh1 = dh1 if (dh1.dtype == torch.float32 and dh1.is_contiguous()) else dh1.float().contiguous()
        dh_hi, dh_lo = _bf16_image(ws, "fc1_dh", Bp, N1p, nsplit == 3)
        part = ws.get("fc1_db_part", (Bp // 16, N1), torch.float32)
        scale_ws = ws.get("fc1_dh_scale", (2 + Bp // 16,), torch.float32) if nsplit == 1 else None   # {scale, 1 / scale, block maxima} of the f16 dh image
        # (kernel target, autograd return): with the trainer's gradient sink the kernels write straight into .grad
        (dWp, rWp), (dbp, rbp), (dWs, rWs), (dbs, rbs) = grad_out(w_priv), grad_out(b_priv), grad_out(w_sh), grad_out(b_sh)
        _abi.call("spv_enc_fc1_bwd_prep", ptr(dh1), ptr(h1), B, N1, ptr(dh_hi), ptr(dh_lo), N1p, Bp, ptr(part), ptr(dbp), ptr(dbs), H, ptr(scale_ws), stream_ptr())
        cs = ctx.counts.c_struct(ctx.rows)
        _abi.call("spv_enc_fc1_wgrad", C.byref(cs), B, G, ptr(dh_hi), ptr(dh_lo), N1p, N1, nsplit, ptr(dWp), ptr(dWs), H, w_priv.shape[1],
                  ptr(ctx.xb), ctx.ld_xb, ptr(scale_ws), stream_ptr())
        if ctx.cov is not None:
            _fc1_cov_wgrad(dh1, h1, ctx.cov[1], dWp, dWs, G, H, ws)
        return None, None, None, rWp, rbp, rWs, rbs, None, None, None


class EncoderFC1Grouped(torch.autograd.Function):
    """``EncoderFC1`` for several groups in one autograd node and -- for pairs of groups whose shapes take the LDS-DMA kernels -- ONE
    launch per kernel (spv_enc_fc1_fwd_grouped / spv_enc_fc1_bwd_grouped: both groups' tiles in one grid).  Two separate launches
    on two streams cannot overlap (each fills the chip with one-per-CU workgroups) and pay the fork / join of the graph branches.
    inputs : per-group lists counts, rows, B, workspaces, covariates (None, or per group (batch codes, one_hot): see EncoderFC1);
             then 4 parameters per group (w_priv, b_priv, w_sh, b_sh)
    outputs: (h1_0, library_0, h1_1, library_1, ...)"""

    @staticmethod
    def forward(ctx, counts, rows, Bs, nsplit: int, wss, covs, *params):
        ctx.set_materialize_grads(False)
        NG = len(counts)
        args = (_abi.SpvFc1FwdArgs * NG)()
        keep, outs, saved, meta = [], [], [], []
        f32c = lambda t: t if (t.dtype == torch.float32 and t.is_contiguous()) else t.float().contiguous()
        lib = _abi.load()
        for g in range(NG):
            w_priv, b_priv, w_sh, b_sh = params[4 * g: 4 * g + 4]
            ws, B = wss[g], Bs[g]
            H, G = w_priv.shape[0], counts[g].G
            cov = covs[g] if covs is not None else None
            if (w_priv.shape[1] != G) != (cov is not None):
                raise _abi.SpvError("fc1 weights with covariate columns need the minibatch's batch codes (and only they do)")
            N1 = 2 * H
            bn = 32 if N1 <= 32 else (128 if N1 <= 128 else 256)
            N1p, Gp = round_up(N1, bn), round_up(G, 64)
            W_hi, W_lo = _bf16_image(ws, "fc1_W", N1p, Gp, nsplit == 3)
            if W_lo is not None or ws.fresh.get("fc1_W") != image_token(w_priv, w_sh):   # (else: Adam has just rewritten the image)
                _pack_fc1_weights(w_priv[:, :G], w_sh[:, :G], W_hi, W_lo, H)
            b_priv, b_sh = f32c(b_priv), f32c(b_sh)
            cov_tab = _fc1_cov_table(w_priv, w_sh, G)
            h1 = torch.empty((B, N1), dtype=torch.float32, device=w_priv.device)
            library = torch.empty((B,), dtype=torch.float32, device=w_priv.device)
            cs = counts[g].c_struct(rows[g])
            xb, ld_xb, lib_all = counts[g].fc1_image(nsplit, B, N1, Gp)
            if lib.spv_enc_fc1_fwd_uses_dma(B, G, N1, nsplit, int(xb is not None), Gp, ld_xb):
                mt = -(-B // 128)
                # K splits: the groups of a pair share one grid, so the 256 CUs are divided by the PAIR's workgroup count (FC1_PAIR_SPLITS:
                # half the splits of a single launch = one round of workgroups with twice the K range, half the slab traffic)
                mt_all = mt * (2 if (FC1_PAIR_SPLITS and NG >= 2) else 1) * (N1 // 256 if N1 >= 256 else 1)
                splits = max(1, min(16, 256 // mt_all, (-(-G // 64)) // 4))
                slabs = ws.get("fc1_slabs_tiled", (splits, mt * 128, N1), torch.float32)
            else:
                splits = _fc1_splits(B, G, N1)
                slabs = ws.get("fc1_slabs", (splits, B, N1), torch.float32)
            rowsum = ws.get("fc1_rowsum", (splits, B), torch.float32)
            a = args[g]
            a.x, a.B, a.G = C.pointer(cs), B, G
            a.W1_hi, a.W1_lo, a.ldw, a.N1 = ptr(W_hi), ptr(W_lo), Gp, N1
            a.bias, a.bias2, a.n_first, a.nsplit, a.splits = ptr(b_priv), ptr(b_sh), H, nsplit, splits
            a.slabs, a.rowsum_ws, a.h1, a.library = ptr(slabs), ptr(rowsum), ptr(h1), ptr(library)
            a.xb_all, a.ld_xb, a.library_all = ptr(xb), ld_xb, ptr(lib_all)
            a.cov, a.cov_idx = ptr(cov_tab), (ptr(cov[0]) if cov else None)
            keep += [cs, b_priv, b_sh, W_hi, W_lo, slabs, rowsum, xb, lib_all, cov_tab]
            outs += [h1, library]
            saved += [h1, w_priv, b_priv, w_sh, b_sh]
            meta.append((counts[g], rows[g], B, H, G, ws) + ((xb, ld_xb) if _wgrad_takes_image(nsplit, xb, ld_xb, B, G, N1) else (None, 0)) + (cov,))
        _abi.call("spv_enc_fc1_fwd_grouped", args, NG, stream_ptr())
        ctx.meta, ctx.nsplit, ctx.NG = meta, nsplit, NG
        ctx.save_for_backward(*saved)
        ctx.mark_non_differentiable(*outs[1::2])
        return tuple(outs)

    @staticmethod
    def backward(ctx, *grads):
        from .nn_ops import grad_out

        NG, nsplit = ctx.NG, ctx.nsplit
        saved = ctx.saved_tensors
        live = [g for g in range(NG) if grads[2 * g] is not None]
        rets = [None] * (4 * NG)
        if live:
            args = (_abi.SpvFc1BwdArgs * len(live))()
            keep, covg = [], []
            for k, g in enumerate(live):
                h1, w_priv, b_priv, w_sh, b_sh = saved[5 * g: 5 * g + 5]
                counts, rows, B, H, G, ws, xb, ld_xb, cov = ctx.meta[g]
                N1 = 2 * H
                Bp, N1p = round_up(B, 64), round_up(N1, 128)
                dh1 = grads[2 * g]
                dh1 = dh1 if (dh1.dtype == torch.float32 and dh1.is_contiguous()) else dh1.float().contiguous()
                dh_hi, dh_lo = _bf16_image(ws, "fc1_dh", Bp, N1p, nsplit == 3)
                part = ws.get("fc1_db_part", (Bp // 16, N1), torch.float32)
                scale_ws = ws.get("fc1_dh_scale", (2 + Bp // 16,), torch.float32) if nsplit == 1 else None
                (dWp, rWp), (dbp, rbp), (dWs, rWs), (dbs, rbs) = grad_out(w_priv), grad_out(b_priv), grad_out(w_sh), grad_out(b_sh)
                cs = counts.c_struct(rows)
                a = args[k]
                a.dh1, a.h1, a.x = ptr(dh1), ptr(h1), C.pointer(cs)
                a.B, a.G, a.N1, a.n_first, a.nsplit, a.Bp = B, G, N1, H, nsplit, Bp
                a.dh_hi, a.dh_lo, a.ld_dh, a.part = ptr(dh_hi), ptr(dh_lo), N1p, ptr(part)
                a.db, a.db2, a.dW, a.dW2, a.ldc = ptr(dbp), ptr(dbs), ptr(dWp), ptr(dWs), w_priv.shape[1]
                a.xb, a.ld_xb, a.scale_ws = ptr(xb), ld_xb, ptr(scale_ws)
                keep += [cs, dh1, dh_hi, dh_lo, part, dWp, dbp, dWs, dbs, scale_ws]
                rets[4 * g: 4 * g + 4] = [rWp, rbp, rWs, rbs]
                if cov is not None:
                    covg.append((dh1, h1, cov[1], dWp, dWs, G, H, ws))
            _abi.call("spv_enc_fc1_bwd_grouped", args, len(live), stream_ptr())
            for c in covg:
                _fc1_cov_wgrad(*c)
        return (None,) * 6 + tuple(rets)


# ------------------------------------------------------------------------------------------------
# decoder + NB-mixture likelihood
# ------------------------------------------------------------------------------------------------
def _gene_splits(Bp: int, Gp: int) -> Tuple[int, int]:
    cell_blocks = Bp // DEC_CELLS_PER_WG
    want = max(1, -(-GSPLIT_WANT // cell_blocks))  # ~2 workgroups per CU
    tiles = Gp // 32
    splits = max(1, min(want, tiles))
    per = -(-tiles // splits) * 32
    splits = -(-Gp // per)
    return splits, per


NB_GSPL_MAX = 160  # genes per likelihood split (their regressor weights sit in LDS)


NB_WG_WANT = int(os.environ.get("SPV_NB_WG_WANT", "512"))


def _nb_splits(Gp: int, Bp: Optional[int] = None) -> Tuple[int, int]:
    """(splits, genes per split) of the likelihood kernel: 160 genes per workgroup (what its LDS weight slice holds) -- fewer for small
    minibatches, where 160-gene splits leave most CUs without a workgroup and every workgroup with a chain of ten 16-gene chunks: about
    NB_WG_WANT workgroups per group, at least 32 genes each"""
    per = min(NB_GSPL_MAX, Gp)
    if Bp is not None and NB_WG_WANT > 0:
        cell_tiles = max(1, Bp // 64)
        want = -(-NB_WG_WANT // cell_tiles)                      # gene splits wanted
        per = max(32, min(per, -(-(-(-Gp // want)) // 32) * 32))   # genes per split, a multiple of 32
    return -(-Gp // per), per


def _nb_cell_tiles(Bp: int, Gp: int) -> int:
    """64-cell tiles one likelihood workgroup walks with one staged weight slice (spv_dec_params.nb_cell_tiles).  Measured at C2
    (tools/probes/nb_bench.hip, alternating): 1 tile 183.8 / 179.7 us, 2 tiles 179.8 / 184.7 us, 4 tiles 188.3 us -- the other
    resident workgroups already cover a workgroup's staging phase, so one tile per workgroup (the finest work split) stays."""
    return 1


def _gemm_slabs(a_kmajor: bool, A_hi, A_lo, lda, B_hi, B_lo, ldb, M, N, K, nsplit, splits, ws: Workspace, name: str,
                b_col_off: int = 0, a_tiles: int = 0) -> torch.Tensor:
    """split-K GEMM leaving its ``splits`` fp32 partial slabs [splits][M][N] for spv_reduce_slabs"""
    out = ws.get(name, (splits, M, N), torch.float32)
    b_off = b_col_off * 2
    _abi.call("spv_gemm_bf16", int(a_kmajor), ptr(A_hi), ptr(A_lo), lda, ptr(B_hi) + b_off, (ptr(B_lo) + b_off) if B_lo is not None else None,
                            ldb, ptr(out), N, M, N, K, nsplit, splits, M * N, a_tiles, stream_ptr())
    return out
