"""spv_gemm_bf16_fix (in-launch split-K sum of the decoder's d A_m GEMM) against spv_gemm_bf16 + a slab sum in the documented order:
bit for bit, on both LDS-DMA kernels (bf16 words and split words), with ragged M / K / column windows, and called twice on the same
counters (the call must leave them zero)."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from spvipes_amd import _abi
    _abi.load()
    return torch.device("cuda:0")


def _tile_order(a: torch.Tensor) -> torch.Tensor:
    """[cells][genes] (both multiples of 32) -> accumulator-tile order T[cell/32][gene/32][qq][lane = cell%32 + 32 h][j],
    gene%32 = 8 qq + 4 h + j (include/spvipes_hip.h)"""
    Mp, Kp = a.shape
    return a.view(Mp // 32, 32, Kp // 32, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).contiguous()


@pytest.mark.parametrize("nsplit", [1, 3])
@pytest.mark.parametrize("M,K,splits,n0,c1,n1,with_alpha", [(300, 1000, 3, 256, 256, 36, True), (128, 2048, 8, 300, 0, 0, False),
                                                           (4096, 2000, 2, 256, 256, 35, True), (77, 640, 5, 17, 20, 300, True)])
def test_fixup_equals_the_slab_sum(dev, nsplit, M, K, splits, n0, c1, n1, with_alpha):
    from spvipes_amd import _abi
    from spvipes_amd._abi import ptr, round_up, stream_ptr
    N, ldb = 320, 320
    Mp, Kp = round_up(M, 128), round_up(K, 64)
    T = Kp // 32
    if _abi.load().spv_gemm_bf16_uses_dma(0, M, N, K, nsplit, T, ldb) != 1:
        pytest.fail("the LDS-DMA kernel does not take this shape: the test would not reach the fix-up")
    g = torch.Generator().manual_seed(M + K)
    A = torch.zeros(Mp, Kp)
    A[:M, :K] = torch.randn(M, K, generator=g)
    Bm = torch.zeros(Kp, ldb)
    Bm[:K, :N] = torch.randn(K, N, generator=g)

    def planes(x):
        hi = x.to(torch.bfloat16)
        lo = (x - hi.float()).to(torch.bfloat16)
        return hi, lo
    a_hi, a_lo = (_tile_order(t).view(torch.int16).to(dev) for t in planes(A))
    b_hi, b_lo = (t.contiguous().view(torch.int16).to(dev) for t in planes(Bm))
    lo = nsplit == 3
    slabs = torch.full((splits, M, N), float("nan"), device=dev)
    common = (0, ptr(a_hi), ptr(a_lo) if lo else None, Kp, ptr(b_hi), ptr(b_lo) if lo else None, ldb)
    _abi.call("spv_gemm_bf16", *common, ptr(slabs), N, M, N, K, nsplit, splits, M * N, T, stream_ptr())
    acc = slabs[0].clone()
    for s in range(1, splits):
        acc = acc + slabs[s]
    alpha = torch.tensor(0.37, device=dev) if with_alpha else None
    want = acc * alpha if with_alpha else acc
    # sanity of the operand construction itself (fp64 product of what the kernel multiplies)
    ref = (A[:M].double() @ Bm[:, :N].double()) if lo else (planes(A)[0][:M].double() @ planes(Bm)[0][:, :N].double())
    torch.testing.assert_close(acc.cpu().double(), ref, rtol=1e-4, atol=2e-4 * K ** 0.5)

    counters = torch.zeros(Mp // 128, dtype=torch.int32, device=dev)
    for _ in range(2):
        slabs2 = torch.full((splits, M, N), float("nan"), device=dev)
        d0 = torch.full((M, n0 + 3), -7.0, device=dev)
        d1 = torch.full((M, n1 + 1), -7.0, device=dev) if n1 else None
        fx = _abi.SpvGemmFixup()
        fx.counters, fx.alpha = ptr(counters), (ptr(alpha) if with_alpha else None)
        fx.dst0, fx.ld0, fx.n0 = ptr(d0), n0 + 3, n0
        fx.dst1, fx.ld1, fx.c1, fx.n1 = (ptr(d1) if n1 else None), n1 + 1, c1, n1
        _abi.call("spv_gemm_bf16_fix", *common, ptr(slabs2), N, M, N, K, nsplit, splits, M * N, T, C.byref(fx), stream_ptr())
        torch.cuda.synchronize()
        assert torch.equal(slabs2, slabs)
        assert torch.equal(d0[:, :n0], want[:, :n0]) and bool((d0[:, n0:] == -7.0).all())
        if n1:
            assert torch.equal(d1[:, :n1], want[:, c1:c1 + n1]) and bool((d1[:, n1:] == -7.0).all())
        assert int(counters.abs().sum()) == 0


def test_fixup_refuses_what_the_dma_kernels_do_not_take(dev):
    from spvipes_amd import _abi
    from spvipes_amd._abi import ptr, stream_ptr
    a = torch.zeros(128 * 64, dtype=torch.int16, device=dev)
    b = torch.zeros(64 * 48, dtype=torch.int16, device=dev)
    c = torch.zeros(128 * 32, device=dev)
    fx = _abi.SpvGemmFixup()
    fx.counters, fx.dst0, fx.ld0, fx.n0 = ptr(torch.zeros(1, dtype=torch.int32, device=dev)), ptr(c), 32, 32
    with pytest.raises(_abi.SpvError):   # N = 32 columns: the narrow kernel, no fix-up there
        _abi.call("spv_gemm_bf16_fix", 0, ptr(a), None, 64, ptr(b), None, 48, ptr(c), 32, 128, 32, 64, 1, 1, 128 * 32, 2, C.byref(fx), stream_ptr())
