"""Direct checks of the auxiliary C-ABI entry points against plain torch on the same inputs."""
import ctypes as C

import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from spvipes_amd import _abi
    _abi.load()  # raises if libspvipes_hip.so is missing: no fallback
    return torch.device("cuda:0")


def _slab_sum_in_kernel_order(x: torch.Tensor) -> torch.Tensor:
    """fp32 sum over dim 0 in the order spv_reduce_slabs documents: slab order up to 8 slabs, beyond that four interleaved groups
    (y, y + 4, ...) summed in order and then ((p0 + p1) + p2) + p3"""
    n = x.shape[0]
    if n <= 8:
        acc = x[0].clone()
        for k in range(1, n):
            acc = acc + x[k]
        return acc
    part = []
    for y in range(4):
        acc = torch.zeros_like(x[0])
        for k in range(y, n, 4):
            acc = acc + x[k]
        part.append(acc)
    return ((part[0] + part[1]) + part[2]) + part[3]


# (37, 50, 21, 7): nothing 16-byte aligned, one element per thread; (37, 52, 24, 8): the four-elements-per-thread form
@pytest.mark.parametrize("shape", [(37, 50, 21, 7), (37, 52, 24, 8), (1, 64, 64, 0)])
@pytest.mark.parametrize("nslabs", [1, 4, 8, 9, 13, 16, 17, 64])
def test_reduce_slabs_matches_torch(dev, nslabs, shape):
    from spvipes_amd import _abi
    from spvipes_amd._abi import SpvReduceBatch, stream_ptr
    from spvipes_amd.dec_ops import _add_red
    g = torch.Generator(device=dev).manual_seed(nslabs)
    rows, ld, cols, off = shape
    src = torch.randn(nslabs, rows, ld, generator=g, device=dev)
    alpha = torch.tensor(0.75, device=dev)
    es = torch.randn(cols, generator=g, device=dev) * 0.3
    dst0 = torch.randn(rows, 4 + cols + 4, generator=g, device=dev)
    dst = dst0.clone()
    plain = torch.empty(rows, cols, device=dev)
    b = SpvReduceBatch()
    b.nprob = 0
    _add_red(b, src, nslabs, rows * ld, ld, rows, cols, plain, cols, col_off=off)
    _add_red(b, src, nslabs, rows * ld, ld, rows, cols, dst, dst0.shape[1], col_off=off, dst_col=4, accumulate=True, alpha=alpha, exp_scale=es)
    _abi.call("spv_reduce_slabs", C.byref(b), stream_ptr())
    want = src[:, :, off:off + cols].double().sum(0)
    torch.testing.assert_close(plain.double(), want, rtol=1e-6, atol=1e-5)
    assert torch.equal(plain, _slab_sum_in_kernel_order(src[:, :, off:off + cols])), "the documented summation order, bit for bit"
    assert torch.equal(dst[:, :4], dst0[:, :4]) and torch.equal(dst[:, 4 + cols:], dst0[:, 4 + cols:]), "columns beside the target"
    want2 = dst0.double().clone()
    want2[:, 4:4 + cols] += 0.75 * torch.exp(es.double()) * want
    torch.testing.assert_close(dst.double(), want2, rtol=1e-5, atol=1e-5)
    # deterministic: a second run gives the same bits
    plain2 = torch.empty_like(plain)
    b2 = SpvReduceBatch()
    b2.nprob = 0
    _add_red(b2, src, nslabs, rows * ld, ld, rows, cols, plain2, cols, col_off=off)
    _abi.call("spv_reduce_slabs", C.byref(b2), stream_ptr())
    assert torch.equal(plain, plain2)


@pytest.mark.parametrize("B,nkl", [(7, 0), (300, 4), (4096, 2)])
def test_loss_assemble_matches_torch(dev, B, nkl):
    from spvipes_amd import _abi
    from spvipes_amd._abi import ptr, stream_ptr
    g = torch.Generator(device=dev).manual_seed(B)
    rec = [torch.randn(B, generator=g, device=dev) * 50 + 900 for _ in range(2)]
    w = torch.full((B,), 1.0 / B, device=dev)
    kls = [torch.rand(B, generator=g, device=dev) * 3 for _ in range(nkl)]
    klw = torch.tensor(0.37, device=dev)
    loss, rs, gkl = torch.empty((), device=dev), torch.empty((), device=dev), torch.empty(B, device=dev)
    klp = (C.c_void_p * 4)(*[ptr(k) for k in kls], *([None] * (4 - nkl)))
    _abi.call("spv_loss_assemble", ptr(rec[0]), ptr(rec[1]), ptr(w), klp, nkl, B, ptr(klw), ptr(loss), ptr(rs), ptr(gkl), stream_ptr())
    want_rs = (w.double() * (rec[0].double() + rec[1].double())).sum()
    want = want_rs + 0.37 * sum(k.double() for k in kls).mean() if nkl else want_rs
    assert abs(float(rs) - float(want_rs)) < 1e-5 * abs(float(want_rs))
    assert abs(float(loss) - float(want)) < 1e-5 * abs(float(want))
    torch.testing.assert_close(gkl, torch.full((B,), 0.37 / B, device=dev), rtol=1e-6, atol=0)


@pytest.mark.parametrize("B,H", [(50, 8), (4096, 128)])
def test_fc1_bwd_prep_matches_torch(dev, B, H):
    """split mode ("fp32" precision): bf16 hi / lo images of dpre = dh1 * (h1 > 0) and the bias gradients"""
    from spvipes_amd import _abi
    from spvipes_amd._abi import ptr, round_up, stream_ptr
    g = torch.Generator(device=dev).manual_seed(H)
    N1 = 2 * H
    dh1 = torch.randn(B, N1, generator=g, device=dev)
    h1 = torch.relu(torch.randn(B, N1, generator=g, device=dev))
    Bp, ld = round_up(B, 64), round_up(N1, 128)
    hi = torch.full((Bp, ld), 77, dtype=torch.int16, device=dev)
    lo = torch.full((Bp, ld), 77, dtype=torch.int16, device=dev)
    part = torch.empty(Bp // 16, N1, device=dev)
    db, db2 = torch.empty(H, device=dev), torch.empty(H, device=dev)
    _abi.call("spv_enc_fc1_bwd_prep", ptr(dh1), ptr(h1), B, N1, ptr(hi), ptr(lo), ld, Bp, ptr(part), ptr(db), ptr(db2), H, None, stream_ptr())
    dpre = dh1 * (h1 > 0)
    hi_f = (hi.to(torch.int32) << 16).view(torch.float32)
    lo_f = (lo.to(torch.int32) << 16).view(torch.float32)
    torch.testing.assert_close(hi_f[:B, :N1], dpre.to(torch.bfloat16).float(), rtol=0, atol=0)
    torch.testing.assert_close((hi_f + lo_f)[:B, :N1], dpre, rtol=2e-5, atol=1e-6)      # hi + lo carries ~16 mantissa bits
    assert float(hi_f[B:].abs().max() if Bp > B else 0) == 0 and float(hi_f[:, N1:].abs().max() if ld > N1 else 0) == 0
    want = dpre.double().sum(0)
    torch.testing.assert_close(torch.cat([db, db2]).double(), want, rtol=1e-5, atol=1e-4)


@pytest.mark.parametrize("magnitude", [1.0, 3e-7, 2e5, 0.0])
@pytest.mark.parametrize("B,H", [(50, 8), (4096, 128), (1000, 256)])
def test_fc1_bwd_prep_f16_image_and_its_scale(dev, B, H, magnitude):
    """one-MFMA mode (img_lo == NULL): the dh image is f16(dpre * scale) with scale the power of two that brings max |dpre| into
    [4096, 8192) -- whatever the magnitude of the gradient (loss-mean factors of 1e-7, exploding gradients, all zeros) -- and the
    record {scale, 1 / scale} the weight-gradient kernel multiplies by; bias gradients as in split mode"""
    import math
    from spvipes_amd import _abi
    from spvipes_amd._abi import ptr, round_up, stream_ptr
    g = torch.Generator(device=dev).manual_seed(H)
    N1 = 2 * H
    dh1 = torch.randn(B, N1, generator=g, device=dev) * magnitude
    h1 = torch.relu(torch.randn(B, N1, generator=g, device=dev))
    Bp, ld = round_up(B, 64), round_up(N1, 128)
    img = torch.full((Bp, ld), 77, dtype=torch.int16, device=dev)
    part = torch.empty(Bp // 16, N1, device=dev)
    scale_ws = torch.full((2 + Bp // 16,), float("nan"), device=dev)
    db, db2 = torch.empty(H, device=dev), torch.empty(H, device=dev)
    _abi.call("spv_enc_fc1_bwd_prep", ptr(dh1), ptr(h1), B, N1, ptr(img), None, ld, Bp, ptr(part), ptr(db), ptr(db2), H, ptr(scale_ws), stream_ptr())
    dpre = dh1 * (h1 > 0)
    amax = float(dpre.abs().max())
    scale = 1.0 if amax == 0.0 else 2.0 ** (13 - math.frexp(amax)[1])
    assert float(scale_ws[0]) == scale and float(scale_ws[1]) == 1.0 / scale
    got = img.view(torch.float16).float()
    assert torch.equal(got[:B, :N1], (dpre * scale).to(torch.float16).float())
    if amax > 0:
        assert 4096 <= float(got.abs().max()) < 8192 + 1
    assert float(got[B:].abs().max() if Bp > B else 0) == 0 and float(got[:, N1:].abs().max() if ld > N1 else 0) == 0
    torch.testing.assert_close(torch.cat([db, db2]).double(), dpre.double().sum(0), rtol=1e-5, atol=1e-4 * max(magnitude, 1e-30))


def test_pack_f16_scales_saturates_and_pads(dev):
    """spv_pack_f16: f16(W * scale), zero padding outside the source, saturation instead of infinities, NaN stays NaN"""
    from spvipes_amd import _abi
    from spvipes_amd._abi import ptr, stream_ptr
    g = torch.Generator().manual_seed(3)
    R, Cc, Rp, ld = 37, 70, 64, 128
    W = (torch.randn(R, Cc, generator=g) * 0.01)
    W[0, 0], W[0, 1], W[0, 2], W[1, 0] = 1e9, -1e9, float("nan"), 1e-9
    Wd = W.to(dev)
    img = torch.full((Rp, ld), 77, dtype=torch.int16, device=dev)
    _abi.call("spv_pack_f16", ptr(Wd), Wd.stride(0), R, Cc, ptr(img), ld, Rp, ld, 256.0, stream_ptr())
    got = img.view(torch.float16).float().cpu()
    want = (W * 256.0).clamp(-65504, 65504).to(torch.float16).float()
    assert torch.equal(got[:R, :Cc].nan_to_num(nan=-7.0), want.nan_to_num(nan=-7.0))
    assert float(got[0, 0]) == 65504.0 and float(got[0, 1]) == -65504.0 and math.isnan(float(got[0, 2]))
    assert float(got[R:].abs().max()) == 0 and float(got[:, Cc:].abs().max()) == 0


def test_plan_experts_match_dense_autograd(dev):
    """spv_plan_expert_fwd / _bwd against the masked, row-normalised dense products of the reference's cluster PoE."""
    import scipy.sparse as sp
    from spvipes_amd import _abi
    from spvipes_amd._abi import SpvPlanExpertArgs, ptr, stream_ptr
    from spvipes_amd.plan import SparsePlan
    rng = np.random.default_rng(0)
    n0, n1, B, n = 400, 350, 96, 7
    m = sp.random(n0, n1, density=0.06, random_state=3, format="csr", dtype=np.float32)
    m.data = m.data + 0.05
    plan = SparsePlan.from_scipy(m, dev)
    idx = [torch.tensor(rng.permutation(nn)[:B].astype(np.int32), device=dev) for nn in (n0, n1)]
    comp = [torch.tensor(rng.integers(0, 4, size=B).astype(np.float32), device=dev) for _ in (0, 1)]
    stats = [torch.randn(B, 2 * n, device=dev, dtype=torch.float32) for _ in (0, 1)]
    plan.bind_minibatch(idx[0], idx[1])
    inv = [plan.inv0, plan.inv1]
    expert = [torch.empty(B, 2 * n, device=dev) for _ in (0, 1)]
    rowsum = [torch.empty(B, device=dev) for _ in (0, 1)]
    d_exp = [torch.randn(B, 2 * n, device=dev) for _ in (0, 1)]
    d_stats = [torch.zeros(B, 2 * n, device=dev) for _ in (0, 1)]
    a = SpvPlanExpertArgs()
    a.plan, a.B, a.n = plan.c_struct(), B, n
    for g in (0, 1):
        a.idx[g], a.inv[g], a.comp[g], a.stats[g], a.ld[g] = ptr(idx[g]), ptr(inv[g]), ptr(comp[g]), ptr(stats[g]), 2 * n
        a.expert[g], a.ld_expert[g], a.rowsum[g], a.d_expert[g], a.d_stats[g] = ptr(expert[g]), 2 * n, ptr(rowsum[g]), ptr(d_exp[g]), ptr(d_stats[g])
    _abi.call("spv_plan_expert_fwd", C.byref(a), stream_ptr())
    _abi.call("spv_plan_expert_bwd", C.byref(a), stream_ptr())
    block = torch.tensor(m.toarray(), device=dev)[idx[0].long()][:, idx[1].long()].double()
    same = comp[0].unsqueeze(1) == comp[1].unsqueeze(0)
    for g, T, sm in ((0, block, same), (1, block.t(), same.t())):
        W = torch.where(sm, T, torch.zeros_like(T))
        W = torch.where(W > 0, W / W.sum(1, keepdim=True).clamp(min=1e-10), W)
        s = stats[g].double().requires_grad_(True)
        E = W @ s
        torch.testing.assert_close(expert[g].double(), E.detach(), rtol=1e-5, atol=1e-6)
        E.backward(d_exp[g].double())
        torch.testing.assert_close(d_stats[g].double(), s.grad, rtol=1e-5, atol=1e-6)


def test_fused_latent_gradient_matches_gemm_path(dev):
    """bf16 mode: the latent gradient produced inside the softmax fix (spv_dec_softmax_bwd dz_part) against the two
    [B,G] x [G,K] GEMMs it replaces -- every parameter gradient of a step, both ways."""
    from spvipes_amd import ops
    from spvipes_amd.data import make_synthetic_group
    from spvipes_amd.module import spVIPESmodule
    from spvipes_amd.train import Trainer
    n, G, B = 1500, 1100, 384
    groups = [make_synthetic_group(g, n, G, dev) for g in (0, 1)]
    res = []
    before = ops.FUSED_DZ
    try:
        for fused in (True, False):
            ops.FUSED_DZ = fused
            torch.manual_seed(0)
            m = spVIPESmodule({0: G, 1: G}, use_labels=True, n_hidden=64, n_dimensions_shared=12, n_dimensions_private=6, dropout_rate=0.0,
                              precision="bf16").to(dev)
            tr = Trainer(m, [g.counts for g in groups], labels=[g.labels for g in groups])
            m.train()
            rows = [torch.arange(B, dtype=torch.int32, device=dev) for _ in (0, 1)]
            torch.manual_seed(5)
            lo = tr._forward_backward(rows, 1.0)
            torch.cuda.synchronize()
            res.append((float(lo.loss.detach()), tr.fp.grad.clone()))
    finally:
        ops.FUSED_DZ = before
    assert res[0][0] == res[1][0]
    diff, scale = float((res[0][1] - res[1][1]).abs().max()), float(res[1][1].abs().max())
    assert diff < 1e-3 * scale, (diff, scale)   # same bf16 operands, different fp32 summation order


@pytest.mark.parametrize("B,G,H,gather", [(100, 333, 64, True), (256, 1000, 128, True), (64, 2001, 16, False)])
def test_fc1_resident_log1p_image_matches_count_decoding(dev, B, G, H, gather):
    """bf16 mode: the fc1 GEMMs fed from the resident f16 log1p image (spv_prepare_log1p + gathered plain operand) against
    the same GEMMs decoding the counts on the fly -- identical f16 operands, so only the fp32 summation order differs."""
    from spvipes_amd import ops
    rng = np.random.default_rng(B + G)
    n_cells = B + 37
    Xh = (rng.poisson(2.0, size=(n_cells, G)) * (rng.random((n_cells, G)) < 0.3)).astype(np.uint16)
    Xh[:, 0] += 1
    X = torch.tensor(Xh.view(np.int16)).to(dev)
    rows = torch.tensor(rng.permutation(n_cells)[:B].astype(np.int32), device=dev) if gather else None
    g = torch.Generator().manual_seed(0)
    base = [(torch.randn(*s, generator=g) * 0.1) for s in ((H, G), (H,), (H, G), (H,))]
    dh = torch.randn(B, 2 * H, generator=g).to(dev)
    out = []
    for resident in (True, False):
        counts = ops.GroupCounts(X, G, 0, resident=resident)
        params = [t.clone().to(dev).requires_grad_(True) for t in base]
        h1, lib = ops.EncoderFC1.apply(counts, rows, B, *params, 1, ops.Workspace(dev))
        (h1 * dh).sum().backward()
        torch.cuda.synchronize()
        out.append((h1.detach(), lib.detach(), [p.grad.clone() for p in params]))
    torch.testing.assert_close(out[0][0], out[1][0], rtol=1e-5, atol=1e-5)
    torch.testing.assert_close(out[0][1], out[1][1], rtol=1e-6, atol=1e-6)
    for a, b in zip(out[0][2], out[1][2]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=1e-4 * float(b.abs().max()))
    ref = torch.log(torch.log1p(torch.tensor(Xh[rows.cpu().numpy() if gather else np.arange(B)].astype(np.float64))).sum(1))
    torch.testing.assert_close(out[0][1].double().cpu(), ref, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("B,G,H,gather", [(100, 333, 128, True), (256, 1000, 128, True), (640, 2001, 128, False), (1000, 700, 256, True), (192, 26000, 128, True)])
def test_fc1_split_resident_image_matches_count_decoding_fp32_mode(dev, B, G, H, gather):
    """"fp32" mode: the fc1 GEMMs fed from the resident split-bf16 image (spv_prepare_log1p_split: hi / lo interleaved per 32-gene
    block) through the LDS-DMA kernels -- forward, and the weight gradient with 96- or 128-gene tiles by round count (G 26 000: 128) --
    against the count-decoding register-staged kernels on the same operands, and against an fp64 product: ragged cell counts (B not a
    multiple of 128 / 64), gene counts that are no multiple of 32 / 96 / 128, both encoder widths."""
    from spvipes_amd import _abi, ops
    rng = np.random.default_rng(B + G)
    n_cells = B + 37
    Xh = (rng.poisson(2.0, size=(n_cells, G)) * (rng.random((n_cells, G)) < 0.3)).astype(np.uint16)
    Xh[:, 0] += 1
    X = torch.tensor(Xh.view(np.int16)).to(dev)
    rows = torch.tensor(rng.permutation(n_cells)[:B].astype(np.int32), device=dev) if gather else None
    g = torch.Generator().manual_seed(0)
    base = [(torch.randn(*s, generator=g) * 0.1) for s in ((H, G), (H,), (H, G), (H,))]
    dh = torch.randn(B, 2 * H, generator=g).to(dev)
    out = []
    for resident in (True, False):
        counts = ops.GroupCounts(X, G, 0, resident=resident)
        params = [t.clone().to(dev).requires_grad_(True) for t in base]
        h1, lib = ops.EncoderFC1.apply(counts, rows, B, *params, 3, ops.Workspace(dev))
        (h1 * dh).sum().backward()
        torch.cuda.synchronize()
        out.append((h1.detach(), lib.detach(), [p.grad.clone() for p in params]))
        if resident:   # the path under test was really taken
            ld = 2 * ops.round_up(ops.round_up(G, 96), 128)
            assert _abi.load().spv_enc_fc1_fwd_uses_dma(B, G, 2 * H, 3, 1, ops.round_up(G, 64), ld) == 1
            assert _abi.load().spv_enc_fc1_wgrad_split_uses_dma(B, G, 2 * H, 2 * H, ld) == 1
            assert counts._xb_split is not None
    torch.testing.assert_close(out[0][0], out[1][0], rtol=2e-5, atol=2e-5)
    torch.testing.assert_close(out[0][1], out[1][1], rtol=1e-6, atol=1e-6)
    for a, b in zip(out[0][2], out[1][2]):
        torch.testing.assert_close(a, b, rtol=1e-4, atol=2e-5 * float(b.abs().max()))
    # fp64 reference of the forward value and of the two weight gradients
    idx = rows.cpu().numpy() if gather else np.arange(B)
    x = torch.log1p(torch.tensor(Xh[idx].astype(np.float64)))
    W = torch.cat([base[0], base[2]], 0).double()
    bias = torch.cat([base[1], base[3]], 0).double()
    pre = x @ W.t() + bias
    # (split-bf16 products carry ~2^-17 relative error per term: three of the four hi / lo cross terms are kept)
    tol = 1e-4 * max(1.0, (G / 1000.0) ** 0.5)   # (a random walk over G rounded terms)
    torch.testing.assert_close(out[0][0].double().cpu(), torch.relu(pre), rtol=tol, atol=tol)
    dW = ((dh.double().cpu() * (out[0][0].cpu() > 0)).t() @ x)   # (the ReLU mask of the run itself: a pre-activation within rounding of 0 may fall either way)
    got = torch.cat([out[0][2][0], out[0][2][2]], 0).double().cpu()
    assert float((got - dW).abs().max()) < 1e-4 * float(dW.abs().max()) + 1e-6
