// C-ABI entry points of libspvipes_hip.so (see include/spvipes_hip.h).  gfx950 only.
#include "../../include/spvipes_hip.h"
#include "spv_common.h"
#include "spv_gemm.h"
#include "spv_fc1.h"
#include "spv_dec_gemm.h"
#include "spv_decoder.h"
#include "spv_small.h"
#include "spv_poe_n.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

using namespace spv;

static thread_local char g_err[512] = "";
static int fail(int code, const char* fmt, const char* what = "") {
  snprintf(g_err, sizeof(g_err), fmt, what);
  return code;
}
static int launch_status(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) {
    snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
    return SPV_ERR_LAUNCH;
  }
  return SPV_OK;
}

#ifndef SPV_BUILD_ID
#define SPV_BUILD_ID "00000000000000000000000000000000"
#endif
static const char g_build_id[] = "SPV_BUILD_ID=" SPV_BUILD_ID;   // (the tagged form is also what build.lib_build_id() greps out of the file)
extern "C" int spv_version(void) { return 2; }
extern "C" const char* spv_build_id(void) { return g_build_id + 13; }
extern "C" const char* spv_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------
// pack
// ---------------------------------------------------------------------------------------------
__global__ void pack_bf16_kernel(const float* src, long ld_src, int R, int C, const float* extra_col, int extra_one,
                                 bf16_t* dst_hi, bf16_t* dst_lo, long ld_dst, int dst_col_off, int Rp, int cslot) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)Rp * cslot;
  if (idx >= total) return;
  const int r = (int)(idx / cslot), c = (int)(idx % cslot);
  float v = 0.f;
  if (r < R) {
    if (c < C) v = src[(long)r * ld_src + c];
    else if (c == C) v = extra_col ? extra_col[r] : (extra_one ? 1.0f : 0.f);
  }
  bf16_t hi, lo;
  split_bf16(v, hi, lo);
  const long o = (long)r * ld_dst + dst_col_off + c;
  dst_hi[o] = hi;
  if (dst_lo) dst_lo[o] = lo;
}

// 8 consecutive destination columns per thread: one 16-byte store per image
__global__ void pack_bf16_vec8_kernel(const float* src, long ld_src, int R, int C, const float* extra_col, int extra_one,
                                      bf16_t* dst_hi, bf16_t* dst_lo, long ld_dst, int dst_col_off, int Rp, int cslot8) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Rp * cslot8) return;
  const int r = (int)(idx / cslot8), c0 = (int)(idx % cslot8) * 8;
  unsigned hw[4] = {0u, 0u, 0u, 0u}, lw[4] = {0u, 0u, 0u, 0u};
  if (r < R) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      float v = 0.f;
      if (c < C) v = src[(long)r * ld_src + c];
      else if (c == C) v = extra_col ? extra_col[r] : (extra_one ? 1.0f : 0.f);
      bf16_t h, l;
      split_bf16(v, h, l);
      hw[j >> 1] |= (unsigned)h << (16 * (j & 1));
      lw[j >> 1] |= (unsigned)l << (16 * (j & 1));
    }
  }
  const long o = (long)r * ld_dst + dst_col_off + c0;
  *reinterpret_cast<u4v*>(dst_hi + o) = u4v{hw[0], hw[1], hw[2], hw[3]};
  if (dst_lo) *reinterpret_cast<u4v*>(dst_lo + o) = u4v{lw[0], lw[1], lw[2], lw[3]};
}

extern "C" int spv_pack_bf16(const float* src, int64_t ld_src, int32_t R, int32_t C, const float* extra_col,
                             int32_t extra_one, uint16_t* dst_hi, uint16_t* dst_lo, int64_t ld_dst,
                             int32_t dst_col_off, int32_t Rp, int32_t cslot, void* stream) {
  if (!src || !dst_hi || R < 0 || C < 0 || Rp < R || cslot < C + ((extra_col || extra_one) ? 1 : 0) ||
      dst_col_off + cslot > ld_dst)
    return fail(SPV_ERR_ARG, "spv_pack_bf16: bad shape%s");
  const long total = (long)Rp * cslot;
  if (total == 0) return SPV_OK;
  if ((cslot % 8) == 0 && (dst_col_off % 8) == 0 && (ld_dst % 8) == 0 && ((uintptr_t)dst_hi % 16) == 0 && (!dst_lo || ((uintptr_t)dst_lo % 16) == 0)) {
    const long tot8 = total / 8;
    hipLaunchKernelGGL(pack_bf16_vec8_kernel, dim3((unsigned)((tot8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (long)ld_src, R, C,
                       extra_col, extra_one, dst_hi, dst_lo, (long)ld_dst, dst_col_off, Rp, cslot / 8);
    return launch_status("spv_pack_bf16");
  }
  hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (long)ld_src, R, C, extra_col, extra_one, dst_hi, dst_lo, (long)ld_dst, dst_col_off, Rp, cslot);
  return launch_status("spv_pack_bf16");
}

// fp32 [R][C] * scale -> IEEE f16 image dst[Rp][ld_dst] (zero filled outside the source): the fc1 weight operand of the
// NSPLIT == 1 mode (spv_common.h).  One thread per 8 destination columns.
__global__ void pack_f16_kernel(const float* src, long ld_src, int R, int C, bf16_t* dst, long ld_dst, int Rp, int cslot8, float scale) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= (long)Rp * cslot8) return;
  const int r = (int)(idx / cslot8), c0 = (int)(idx % cslot8) * 8;
  unsigned w[4] = {0u, 0u, 0u, 0u};
  if (r < R) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = c0 + j;
      const float v = (c < C) ? src[(long)r * ld_src + c] * scale : 0.f;
      w[j >> 1] |= (unsigned)f2h(v) << (16 * (j & 1));
    }
  }
  *reinterpret_cast<u4v*>(dst + (long)r * ld_dst + c0) = u4v{w[0], w[1], w[2], w[3]};
}

extern "C" float spv_fc1_w_scale(void) { return SPV_FC1_W_SCALE; }

extern "C" int spv_pack_f16(const float* src, int64_t ld_src, int32_t R, int32_t C, uint16_t* dst, int64_t ld_dst, int32_t Rp, int32_t cslot,
                            float scale, void* stream) {
  if (!src || !dst || R < 0 || C < 0 || Rp < R || cslot < C || cslot > ld_dst || (cslot % 8) || (ld_dst % 8) || ((uintptr_t)dst % 16))
    return fail(SPV_ERR_ARG, "spv_pack_f16: bad shape (cslot, ld_dst multiples of 8, dst 16-byte aligned)%s");
  const long tot8 = (long)Rp * (cslot / 8);
  if (tot8 == 0) return SPV_OK;
  hipLaunchKernelGGL(pack_f16_kernel, dim3((unsigned)((tot8 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, (long)ld_src, R, C, dst, (long)ld_dst,
                     Rp, cslot / 8, scale);
  return launch_status("spv_pack_f16");
}

// ---------------------------------------------------------------------------------------------
// encoder fc1
// ---------------------------------------------------------------------------------------------
__global__ void fc1_epilogue_kernel(const float* slabs, const float* rowsum_ws, int splits, int B, int N1, const float* bias,
                                    const float* bias2, int n_first, float* h1, float* library, const float* library_all,
                                    const int* rows, float acc_scale, const float* cov, const int* cov_idx) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * N1;
  if (idx < total) {
    const int col = (int)(idx % N1);
    const float bv = (bias2 != nullptr && col >= n_first) ? bias2[col - n_first] : bias[col];
    const float cv = cov ? cov[(long)cov_idx[idx / N1] * N1 + col] : 0.f;   // one-hot batch covariates (nn/networks.py:105-119): a per-cell bias row
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += slabs[(long)s * total + idx];
    h1[idx] = fmaxf(v * acc_scale + bv + cv, 0.f);  // relu(fc1(x)), nn/networks.py:119
  }
  if (idx < B) {
    if (library_all != nullptr) {  // precomputed per cell of the data set (spv_prepare_log1p)
      library[idx] = library_all[rows ? rows[idx] : (int)idx];
    } else {
      float v = 0.f;
      for (int s = 0; s < splits; ++s) v += rowsum_ws[(long)s * B + idx];
      library[idx] = __logf(v);  // log(sum of log1p(x)), module/spVIPESmodule.py:435
    }
  }
}

struct Fc1EpiPlainArgs { const float* slabs; const float* rowsum_ws; int splits, B, N1; const float* bias; const float* bias2; int n_first; float* h1; float* library;
                         const float* library_all; const int* rows; float acc_scale; const float* cov; const int* cov_idx; };
__global__ void fc1_epilogue_pair_kernel(Fc1EpiPlainArgs a0, Fc1EpiPlainArgs a1) {   // blockIdx.y = group; the same arithmetic as fc1_epilogue_kernel
  const Fc1EpiPlainArgs a = blockIdx.y ? a1 : a0;
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)a.B * a.N1;
  if (idx < total) {
    const int col = (int)(idx % a.N1);
    const float bv = (a.bias2 != nullptr && col >= a.n_first) ? a.bias2[col - a.n_first] : a.bias[col];
    const float cv = a.cov ? a.cov[(long)a.cov_idx[idx / a.N1] * a.N1 + col] : 0.f;
    float v = 0.f;
    for (int s = 0; s < a.splits; ++s) v += a.slabs[(long)s * total + idx];
    a.h1[idx] = fmaxf(v * a.acc_scale + bv + cv, 0.f);
  }
  if (idx < a.B) {
    if (a.library_all != nullptr) {
      a.library[idx] = a.library_all[a.rows ? a.rows[idx] : (int)idx];
    } else {
      float v = 0.f;
      for (int s = 0; s < a.splits; ++s) v += a.rowsum_ws[(long)s * a.B + idx];
      a.library[idx] = __logf(v);
    }
  }
}

// every 8-gene chunk of a row is one aligned 16-byte load when base, row pitch and column offset are
static int counts_aligned(const spv_counts* x) {
  const long esz = (x->dtype == SPV_COUNT_U16) ? 2 : 4;
  return ((reinterpret_cast<uintptr_t>(x->X) & 15) == 0) && ((x->ld * esz) % 16 == 0) && ((x->col_off * esz) % 16 == 0);
}

template <typename CT, int NSPLIT>
static int fc1_fwd_dispatch(const GemmParams& p, int N1, int splits, hipStream_t s) {
  constexpr bool HF = NSPLIT == 1;   // 16-bit operand words of the one-MFMA mode are f16 (spv_common.h)
  if (N1 <= 32) return launch_gemm<GemmCfg<128, 32, 4, 1, false, false, SRC_COUNTS, SRC_PLAIN, CT, NSPLIT, 32, 4, 1, HF>>(p, splits, s);
  if (N1 <= 128) return launch_gemm<GemmCfg<64, 128, 2, 2, false, false, SRC_COUNTS, SRC_PLAIN, CT, NSPLIT, 32, 4, 1, HF>>(p, splits, s);
  if constexpr (NSPLIT == 1) {
    if (p.M >= 1024) return launch_gemm<GemmCfg<128, 256, 2, 2, false, false, SRC_COUNTS, SRC_PLAIN, CT, NSPLIT, 32, 2, 2, HF>>(p, splits, s);  // 64 x 128 wave tiles
  }
  return launch_gemm<GemmCfg<64, 256, 1, 4, false, false, SRC_COUNTS, SRC_PLAIN, CT, NSPLIT, 32, 4, 1, HF>>(p, splits, s);
}

// Shapes the LDS-DMA fc1 forward kernel (spv_fc1.h) takes: the resident bf16 log1p image, both encoders' 2H = 256 output columns,
// operand images zero padded to multiples of 64 genes with 16-byte aligned rows.  The host asks through this entry point
// because the path fixes the slab workspace: [splits][round_up(B, 128)][256] fp32 in accumulator-tile order.
// SPV_FC1_DMA_SPLIT=0: "fp32" mode keeps the count-decoding register-staged fc1 kernels (A/B switch)
static const bool g_fc1_dma_split = []() { const char* e = getenv("SPV_FC1_DMA_SPLIT"); return !(e && e[0] == '0'); }();
extern "C" int spv_enc_fc1_fwd_uses_dma(int32_t B, int32_t G, int32_t N1, int32_t nsplit, int32_t have_xb, int64_t ldw, int64_t ld_xb) {
  const long G64 = (G + 63) & ~63L;
  if (nsplit == 3)   // "fp32" mode: the resident image holds [hi | lo] halves per row (spv_prepare_log1p_split): ld_xb is the pitch of both
    return g_fc1_dma_split && have_xb && (N1 == F1_BN || N1 == 2 * F1_BN) && B > 0 && G > 0 && ldw >= G64 && ld_xb >= 2 * G64 && (ldw % 8) == 0 && (ld_xb % 16) == 0;
  return have_xb && nsplit == 1 && (N1 == F1_BN || N1 == 2 * F1_BN) && B > 0 && G > 0 && ldw >= G64 && ld_xb >= G64 && (ldw % 8) == 0 && (ld_xb % 8) == 0;
}

static int fc1_bn(int N1) { return N1 <= 32 ? 32 : (N1 <= 128 ? 128 : 256); }

extern "C" int spv_enc_fc1_fwd(const spv_counts* x, int32_t B, int32_t G, const uint16_t* W1_hi, const uint16_t* W1_lo,
                               int64_t ldw, int32_t N1, const float* bias, const float* bias2, int32_t n_first, int32_t nsplit,
                               int32_t splits, float* slabs, float* rowsum_ws, float* h1, float* library, const uint16_t* xb_all,
                               int64_t ld_xb, const float* library_all, const float* cov, const int32_t* cov_idx, void* stream) {
  if (!x || !x->X || !W1_hi || !bias || !slabs || !rowsum_ws || !h1 || !library) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: null pointer%s");
  if ((cov == nullptr) != (cov_idx == nullptr)) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: cov and cov_idx go together%s");
  if (B <= 0 || G <= 0 || N1 <= 0 || splits <= 0 || (ldw % 32) != 0 || ldw < ((G + 31) & ~31)) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: bad shape%s");
  if (nsplit != 1 && nsplit != 3) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: nsplit must be 1 or 3%s");
  if (nsplit == 3 && !W1_lo) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: nsplit=3 needs W1_lo%s");
  GemmParams p{};
  p.A = x->X; p.A_lo = nullptr; p.lda = x->ld;
  p.B = W1_hi; p.B_lo = W1_lo; p.ldb = ldw;
  p.rows = x->rows; p.counts_aligned = counts_aligned(x); p.col_off = x->col_off; p.n_cells = B; p.n_genes = G;
  p.rowsum = rowsum_ws;
  if (xb_all != nullptr) {  // log1p(x) of the whole data set is resident as bf16 (spv_prepare_log1p): plain gathered operand, no decode
    if (!library_all || ld_xb < ((G + 31) & ~31) || (ld_xb % 8)) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: xb_all needs library_all and ld_xb >= round_up(G, 32)%s");
    if (nsplit == 3 && !spv_enc_fc1_fwd_uses_dma(B, G, N1, nsplit, 1, ldw, ld_xb)) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: a split (nsplit 3) resident image is only taken by the LDS-DMA kernel (ask spv_enc_fc1_fwd_uses_dma)%s");
    p.A = xb_all; p.lda = ld_xb; p.rowsum = nullptr;
    // (nsplit 3: hi / lo planes interleaved inside the rows, spv_prepare_log1p_split)
  }
  p.C = slabs; p.ldc = N1; p.slab_stride = (long)B * N1;
  p.M = B; p.N = N1; p.K = G;
  // resident-image path with both encoders' 2H > 128 columns: 128 x 256 tiles, 64-deep k steps and ONE register stage
  // (the 64 x 128 wave tile holds 128 accumulator registers; a deeper prefetch ring spills at two workgroups per CU --
  // measured 88 -> 62 us) when both images are zero padded to a multiple of 64 genes, else 32-deep steps
  const int G64 = (G + 63) & ~63;
  const bool wide64 = xb_all != nullptr && N1 > 128 && ld_xb >= G64 && ldw >= G64;
  const int bk = wide64 ? 64 : 32;
  const int ktiles = (G + bk - 1) / bk;
  p.k_per_split = ((ktiles + splits - 1) / splits) * bk;
  p.epi = EPI_STORE;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  const float acc_scale = (nsplit == 1) ? 1.0f / SPV_FC1_W_SCALE : 1.0f;   // the f16 weight image holds W * SPV_FC1_W_SCALE (spv_pack_f16)
  if (nsplit == 3 && xb_all != nullptr) {
    // "fp32" mode on a resident split image: the LDS-DMA kernel on hi / lo planes (spv_fc1.h: fc1_fwd_dma_body<true>)
    if (((reinterpret_cast<uintptr_t>(xb_all) | reinterpret_cast<uintptr_t>(W1_hi) | reinterpret_cast<uintptr_t>(W1_lo)) & 15) != 0) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: images must be 16-byte aligned%s");
    const int mtiles = (B + F1_BM - 1) / F1_BM, kt = (G + F1_BK - 1) / F1_BK;
    p.k_per_split = ((kt + splits - 1) / splits) * F1_BK;
    p.c_split_row = splits;
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fc1_fwd_dma_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS_BYTES); raised = true; }
    hipLaunchKernelGGL(fc1_fwd_dma_split_kernel, dim3(mtiles * splits, N1 / F1_BN), dim3(512), F1_LDS_BYTES, s, p);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_enc_fc1_fwd dma split gemm");
    const long slab_elems = (long)mtiles * F1_BM * N1;
    hipLaunchKernelGGL(fc1_epilogue_tiled_kernel, dim3((unsigned)((slab_elems / 4 + 255) / 256)), dim3(256), 0, s, slabs, splits, slab_elems, B, N1, bias, bias2, n_first, h1, library,
                       library_all, x->rows, 1.0f, cov, cov_idx);
    return launch_status("spv_enc_fc1_fwd dma split epilogue");
  }
  if (spv_enc_fc1_fwd_uses_dma(B, G, N1, nsplit, xb_all != nullptr, ldw, ld_xb) && ((reinterpret_cast<uintptr_t>(xb_all) | reinterpret_cast<uintptr_t>(W1_hi)) & 15) == 0) {
    // LDS-DMA kernel (spv_fc1.h): 128-cell tiles, K split `splits` ways, slabs in accumulator-tile order [splits][Mp128][256]
    const int mtiles = (B + F1_BM - 1) / F1_BM, kt = (G + F1_BK - 1) / F1_BK;
    p.k_per_split = ((kt + splits - 1) / splits) * F1_BK;
    p.c_split_row = splits;
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fc1_fwd_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS_BYTES); raised = true; }
    hipLaunchKernelGGL(fc1_fwd_dma_kernel, dim3(mtiles * splits, N1 / F1_BN), dim3(512), F1_LDS_BYTES, s, p);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_enc_fc1_fwd dma gemm");
    const long slab_elems = (long)mtiles * F1_BM * N1;
    hipLaunchKernelGGL(fc1_epilogue_tiled_kernel, dim3((unsigned)((slab_elems / 4 + 255) / 256)), dim3(256), 0, s, slabs, splits, slab_elems, B, N1, bias, bias2, n_first, h1, library,
                       library_all, x->rows, acc_scale, cov, cov_idx);
    return launch_status("spv_enc_fc1_fwd dma epilogue");
  }
  if (xb_all != nullptr) {
    if (N1 <= 128) rc = launch_gemm<GemmCfg<64, 128, 2, 2, false, false, SRC_GATHER, SRC_PLAIN, unsigned short, 1, 32, 4, 1, true>>(p, splits, s);
    else if (wide64) rc = launch_gemm<GemmCfg<128, 256, 2, 2, false, false, SRC_GATHER, SRC_PLAIN, unsigned short, 1, 64, 1, 2, true>>(p, splits, s);
    else rc = launch_gemm<GemmCfg<128, 256, 2, 2, false, false, SRC_GATHER, SRC_PLAIN, unsigned short, 1, 32, 2, 2, true>>(p, splits, s);
  } else if (x->dtype == SPV_COUNT_U16) rc = (nsplit == 3) ? fc1_fwd_dispatch<unsigned short, 3>(p, N1, splits, s) : fc1_fwd_dispatch<unsigned short, 1>(p, N1, splits, s);
  else if (x->dtype == SPV_COUNT_F32) rc = (nsplit == 3) ? fc1_fwd_dispatch<float, 3>(p, N1, splits, s) : fc1_fwd_dispatch<float, 1>(p, N1, splits, s);
  else return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: unknown count dtype%s");
  if (rc != SPV_OK) return launch_status("spv_enc_fc1_fwd gemm");
  const long total = (long)B * N1;
  hipLaunchKernelGGL(fc1_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slabs, rowsum_ws, splits, B, N1, bias, bias2, n_first, h1, library, xb_all ? library_all : nullptr, x->rows, acc_scale, cov, cov_idx);
  return launch_status("spv_enc_fc1_fwd epilogue");
}

// 1 when spv_enc_fc1_wgrad takes a split (nsplit 3) resident image for these shapes (the LDS-DMA kernel is its only consumer)
extern "C" int spv_enc_fc1_wgrad_split_uses_dma(int32_t B, int32_t G, int32_t N1, int64_t ld_dh, int64_t ld_xb) {
  if (!g_fc1_dma_split || B <= 0 || G <= 0) return 0;
  const int Kpad = (B + FW_BK - 1) / FW_BK * FW_BK;
  return (N1 == FW_BM || N1 == 2 * FW_BM) && ld_dh == N1 && ld_xb >= 2 * (((long)G + 95) / 96 * 96) && (ld_xb % 128) == 0 && fw_lds_bytes(96, Kpad) <= 160 * 1024;
}
extern "C" int spv_enc_fc1_wgrad(const spv_counts* x, int32_t B, int32_t G, const uint16_t* dh_hi, const uint16_t* dh_lo,
                                 int64_t ld_dh, int32_t N1, int32_t nsplit, float* dW, float* dW2, int32_t rows_first, int64_t ldc,
                                 const uint16_t* xb, int64_t ld_xb, const float* dh_scale, void* stream) {
  if (!x || !x->X || !dh_hi || !dW) return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: null pointer%s");
  if (nsplit == 1 && !dh_scale) return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: nsplit=1 needs the dh image's scale record (spv_enc_fc1_bwd_prep)%s");
  if (B <= 0 || G <= 0 || N1 <= 0 || ld_dh < ((N1 + 127) & ~127) || (ld_dh % 8) != 0 || ldc < G) return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: bad shape%s");
  if (nsplit != 1 && nsplit != 3) return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: nsplit must be 1 or 3%s");
  if (nsplit == 3 && !dh_lo) return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: nsplit=3 needs dh_lo%s");
  GemmParams p{};
  p.A = dh_hi; p.A_lo = dh_lo; p.lda = ld_dh;
  p.B = x->X; p.B_lo = nullptr; p.ldb = x->ld;
  p.rows = x->rows; p.counts_aligned = counts_aligned(x); p.col_off = x->col_off; p.n_cells = B; p.n_genes = G;
  p.rowsum = nullptr;
  p.C = dW; p.ldc = ldc; p.slab_stride = 0;
  p.C2 = dW2; p.c_split_row = rows_first;
  p.out_scale = (nsplit == 1) ? dh_scale + 1 : nullptr;   // {scale, 1 / scale} of the f16 dh image
  p.M = N1; p.N = G; p.K = B;
  p.k_per_split = (B + 63) & ~63;
  p.epi = EPI_STORE;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (xb != nullptr && nsplit == 3) {
    // "fp32" mode on the resident split image (spv_prepare_log1p_split): the LDS-DMA kernel on hi / lo planes, 96-gene tiles
    const int Kpad = (B + FW_BK - 1) / FW_BK * FW_BK;
    if (!spv_enc_fc1_wgrad_split_uses_dma(B, G, N1, ld_dh, ld_xb) || rows_first != N1 / 2 || dW2 == nullptr ||
        ((reinterpret_cast<uintptr_t>(xb) | reinterpret_cast<uintptr_t>(dh_hi) | reinterpret_cast<uintptr_t>(dh_lo)) & 15) != 0)
      return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: a split (nsplit 3) resident image is only taken by the LDS-DMA kernel (ask spv_enc_fc1_wgrad_split_uses_dma)%s");
    p.B = xb; p.ldb = ld_xb;
    // gene tile: 128 where that saves a round of one-per-CU workgroups over 96 (and its LDS need -- 144 KiB of stages + the row table -- fits)
    const int mth = N1 / FW_BM, nb96 = (G + 95) / 96, nb128 = (G + 127) / 128;
    static const int force_bn = getenv("SPV_FC1W_SPLIT_BN") ? atoi(getenv("SPV_FC1W_SPLIT_BN")) : 0;   // (A/B switch)
    const bool fits128 = fw_lds_bytes(128, Kpad) <= 160 * 1024 && ld_xb >= 2 * (long)nb128 * 128;
    const bool use128 = fits128 && (force_bn ? force_bn == 128 : (nb128 * mth + 255) / 256 < (nb96 * mth + 255) / 256);
    void (*kfn)(GemmParams) = use128 ? fc1_wgrad_dma_split_kernel<128> : fc1_wgrad_dma_split_kernel<96>;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL(kfn, dim3(use128 ? nb128 : nb96, mth), dim3(512), fw_lds_bytes(use128 ? 128 : 96, Kpad), s, p);
    return launch_status("spv_enc_fc1_wgrad dma split");
  }
  if (xb != nullptr) {  // resident bf16 log1p(x) of the data set (spv_prepare_log1p): gathered plain k-major operand, no decode
    if (nsplit != 1 || ld_xb < ((G + 63) & ~63) || (ld_xb % 8)) return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: xb needs nsplit 1 and ld_xb >= round_up(G, 64)%s");
    p.B = xb; p.ldb = ld_xb;
    const int Kpad = (B + FW_BK - 1) / FW_BK * FW_BK;
    if ((N1 == FW_BM || N1 == 2 * FW_BM) && ld_dh == N1 && rows_first == N1 / 2 && dW2 != nullptr && fw_lds_bytes(64, Kpad) <= 160 * 1024 &&
        ((reinterpret_cast<uintptr_t>(xb) | reinterpret_cast<uintptr_t>(dh_hi)) & 15) == 0) {
      // LDS-DMA kernel (spv_fc1.h): all 256 rows x 64 genes per workgroup over the whole batch, results straight into dW / dW2
      // (measured at C2, tools/probes/fc1w_bench.hip: 51 us alone against 75 us for the register-staged kernel below)
      // 96-gene tiles when they put TWO groups' launches (two streams, one workgroup per CU) into one round of the 256 CUs and
      // 64-gene tiles do not (C2: 2 x 105 against 2 x 157 workgroups; alone the 64-gene tile is the faster one)
      const int nb64 = (G + 63) / 64, nb96 = (G + 95) / 96;
      static const int force_bn = getenv("SPV_FC1W_BN") ? atoi(getenv("SPV_FC1W_BN")) : 0;   // (A/B switch)
      const bool use96 = force_bn ? force_bn == 96 : (2 * nb96 <= 256 && 2 * nb64 > 256 && fw_lds_bytes(96, Kpad) <= 160 * 1024);
      void (*kfn)(GemmParams) = use96 ? fc1_wgrad_dma_kernel<96> : fc1_wgrad_dma_kernel<64>;
      const int lds = fw_lds_bytes(use96 ? 96 : 64, Kpad);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      hipLaunchKernelGGL(kfn, dim3(use96 ? nb96 : nb64, N1 / FW_BM), dim3(512), lds, s, p);
      return launch_status("spv_enc_fc1_wgrad dma");
    }
    // 96-gene tiles when the image rows are padded that far: at G = 10 000 that is 2 x 105 workgroups per group, so the two
    // groups' GEMMs (two streams) are resident together in one round instead of 2 x 314 workgroups in two (-0.04 ms per
    // step; alone the 64-gene tile is 4 us faster)
    if (ld_xb >= (G + 95) / 96 * 96) rc = launch_gemm<GemmCfg<128, 96, 4, 1, true, true, SRC_PLAIN, SRC_GATHER, unsigned short, 1, 64, 4, 1, true>>(p, 1, s);
    else rc = launch_gemm<GemmCfg<128, 64, 4, 1, true, true, SRC_PLAIN, SRC_GATHER, unsigned short, 1, 64, 4, 1, true>>(p, 1, s);
  } else if (x->dtype == SPV_COUNT_U16) {
    rc = (nsplit == 3) ? launch_gemm<GemmCfg<128, 64, 4, 1, true, true, SRC_PLAIN, SRC_COUNTS, unsigned short, 3, 64, 4>>(p, 1, s)
                       : launch_gemm<GemmCfg<128, 64, 4, 1, true, true, SRC_PLAIN, SRC_COUNTS, unsigned short, 1, 64, 4, 1, true>>(p, 1, s);
  } else if (x->dtype == SPV_COUNT_F32) {
    rc = (nsplit == 3) ? launch_gemm<GemmCfg<128, 64, 4, 1, true, true, SRC_PLAIN, SRC_COUNTS, float, 3, 64, 4>>(p, 1, s)
                       : launch_gemm<GemmCfg<128, 64, 4, 1, true, true, SRC_PLAIN, SRC_COUNTS, float, 1, 64, 4, 1, true>>(p, 1, s);
  } else return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: unknown count dtype%s");
  (void)rc;
  return launch_status("spv_enc_fc1_wgrad");
}


// ---------------------------------------------------------------------------------------------
template <bool A_KMAJ, int NSPLIT, int A_SRC>
static int gemm_dispatch(const GemmParams& p, int splits, hipStream_t s) {
  if (p.N <= 32) return launch_gemm<GemmCfg<128, 32, 4, 1, A_KMAJ, true, A_SRC, SRC_PLAIN, unsigned short, NSPLIT, 64, 4>>(p, splits, s);
  return launch_gemm<GemmCfg<64, 320, 2, 2, A_KMAJ, true, A_SRC, SRC_PLAIN, unsigned short, NSPLIT, 64, 1, (NSPLIT == 1 ? 2 : 1)>>(p, splits, s);
}

// Shapes the LDS-DMA 320-column kernels (spv_dec_gemm.h) take: bf16 mode, a tile-ordered [cells][genes] A operand, a 320-wide
// k-major B image.  The tiled array must cover round_up(M, 128) x round_up(K, 64) (cells x genes, a_kmajor 0) or
// round_up(K, 64) x round_up(M, 128) (a_kmajor 1): the decoder's Bp / Gp paddings (multiples of 128 / 256) do.
// SPV_GEMM_DMA_SPLIT=0: "fp32" mode keeps the register-staged 64 x 320 kernel for the two 320-column GEMMs (A/B switch)
static const bool g_dma_split = []() { const char* e = getenv("SPV_GEMM_DMA_SPLIT"); return !(e && e[0] == '0'); }();
extern "C" int spv_gemm_bf16_uses_dma(int32_t a_kmajor, int32_t M, int32_t N, int32_t K, int32_t nsplit, int32_t a_tiles, int64_t ldb) {
  if ((nsplit != 1 && nsplit != 3) || a_tiles <= 0 || N <= 32 || N > DG_BN || ldb != DG_BN || M <= 0 || K <= 0) return 0;
  if (nsplit == 3 && !g_dma_split) return 0;
  const long genes_needed = a_kmajor ? ((long)M + DG_BM - 1) / DG_BM * DG_BM : ((long)K + DG_BK - 1) / DG_BK * DG_BK;
  if ((long)a_tiles * 32 < genes_needed) return 0;
  return 1;
}

extern "C" int spv_gemm_bf16(int32_t a_kmajor, const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda, const uint16_t* B_hi,
                             const uint16_t* B_lo, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                             int32_t nsplit, int32_t splits, int64_t slab_stride, int32_t a_tiles, void* stream) {
  return spv_gemm_bf16_fix(a_kmajor, A_hi, A_lo, lda, B_hi, B_lo, ldb, C, ldc, M, N, K, nsplit, splits, slab_stride, a_tiles, nullptr, stream);
}

extern "C" int spv_gemm_bf16_fix(int32_t a_kmajor, const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda, const uint16_t* B_hi,
                                 const uint16_t* B_lo, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                                 int32_t nsplit, int32_t splits, int64_t slab_stride, int32_t a_tiles, const spv_gemm_fixup* fix, void* stream) {
  if (!A_hi || !B_hi || !C) return fail(SPV_ERR_ARG, "spv_gemm_bf16: null pointer%s");
  if (fix) {
    if (!spv_gemm_bf16_uses_dma(a_kmajor, M, N, K, nsplit, a_tiles, ldb)) return fail(SPV_ERR_UNSUPPORTED, "spv_gemm_bf16_fix: only the LDS-DMA 320-column kernels sum their slabs in the launch%s");
    if (!fix->counters || !fix->dst0 || fix->n0 <= 0 || fix->n0 > N || fix->ld0 < fix->n0 || (ldc & 3) || (slab_stride & 3) || (reinterpret_cast<uintptr_t>(C) & 15) ||
        (fix->dst1 && (fix->c1 < fix->n0 || fix->n1 <= 0 || fix->c1 + fix->n1 > N || fix->ld1 < fix->n1)) || ldc < N)
      return fail(SPV_ERR_ARG, "spv_gemm_bf16_fix: bad fix-up description%s");
    // (the LDS-DMA kernels want 16-byte aligned planes; the register-staged kernels they would fall back to know nothing of the fix-up)
    if (((reinterpret_cast<uintptr_t>(A_hi) | reinterpret_cast<uintptr_t>(B_hi) | (nsplit == 3 ? (reinterpret_cast<uintptr_t>(A_lo) | reinterpret_cast<uintptr_t>(B_lo)) : 0)) & 15) != 0)
      return fail(SPV_ERR_ARG, "spv_gemm_bf16_fix: operand planes must be 16-byte aligned%s");
  }
  if (M <= 0 || N <= 0 || K <= 0 || splits <= 0 || (!a_tiles && (lda % 8)) || (ldb % 8) || a_tiles < 0) return fail(SPV_ERR_ARG, "spv_gemm_bf16: bad shape%s");
  if (nsplit != 1 && nsplit != 3) return fail(SPV_ERR_ARG, "spv_gemm_bf16: nsplit must be 1 or 3%s");
  if (nsplit == 3 && (!A_lo || !B_lo)) return fail(SPV_ERR_ARG, "spv_gemm_bf16: nsplit=3 needs lo images%s");
  const int bn = (N <= 32) ? 32 : 320;
  if (ldb < ((N + bn - 1) / bn) * bn) return fail(SPV_ERR_ARG, "spv_gemm_bf16: B image narrower than the N tiling%s");
  GemmParams p{};
  p.A = A_hi; p.A_lo = A_lo; p.lda = lda;
  p.B = B_hi; p.B_lo = B_lo; p.ldb = ldb;
  p.C = C; p.ldc = ldc; p.slab_stride = slab_stride;
  p.M = M; p.N = N; p.K = K;
  const int ktiles = (K + 63) / 64;  // BK = 64: operands are padded to multiples of 64 along K
  p.k_per_split = ((ktiles + splits - 1) / splits) * 64;
  p.epi = EPI_STORE;
  p.tiles_inner = a_tiles;
  if (fix) {
    p.fix_cnt = fix->counters; p.fix_alpha = fix->alpha; p.fix_d0 = fix->dst0; p.fix_ld0 = fix->ld0; p.fix_n0 = fix->n0;
    p.fix_d1 = fix->dst1; p.fix_ld1 = fix->ld1; p.fix_c1 = fix->c1; p.fix_n1 = fix->n1;
  }
  hipStream_t s = (hipStream_t)stream;
  if (nsplit == 3 && spv_gemm_bf16_uses_dma(a_kmajor, M, N, K, nsplit, a_tiles, ldb) &&
      ((reinterpret_cast<uintptr_t>(A_hi) | reinterpret_cast<uintptr_t>(B_hi) | reinterpret_cast<uintptr_t>(A_lo) | reinterpret_cast<uintptr_t>(B_lo)) & 15) == 0) {
    // split-bf16 operands ("fp32" mode): the same tiles from hi and lo planes, 16-deep stages, three MFMAs per fragment pair
    const int mtiles = (M + DG_BM - 1) / DG_BM;
    p.c_split_row = splits;
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gemm320_dma4s_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, D4S_LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gemm320_dma4s_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, D4S_LDS_BYTES);
      raised = true;
    }
    if (a_kmajor) hipLaunchKernelGGL(dec_gemm320_dma4s_kernel<true>, dim3(mtiles * splits), dim3(512), D4S_LDS_BYTES, s, p);
    else hipLaunchKernelGGL(dec_gemm320_dma4s_kernel<false>, dim3(mtiles * splits), dim3(512), D4S_LDS_BYTES, s, p);
    return launch_status("spv_gemm_bf16 dma split");
  }
  if (nsplit == 1 && spv_gemm_bf16_uses_dma(a_kmajor, M, N, K, nsplit, a_tiles, ldb) && ((reinterpret_cast<uintptr_t>(A_hi) | reinterpret_cast<uintptr_t>(B_hi)) & 15) == 0) {
    // LDS-DMA kernels (spv_dec_gemm.h): 128 x 320 workgroup tiles, `splits` K ranges, four 28 KiB stages (measured at C2,
    // tools/probes/dec_gemm_bench.hip: d A_m 37.5 us, d W_m 38.9 us against 76.5 / 84.8 us for the register-staged 64 x 320 kernel below;
    // the two-stage x 64-deep version of the same kernel: 38.2 / 41.5 us)
    const int mtiles = (M + DG_BM - 1) / DG_BM;
    p.c_split_row = splits;
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gemm320_dma4_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, D4_LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gemm320_dma4_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, D4_LDS_BYTES);
      raised = true;
    }
    if (a_kmajor) hipLaunchKernelGGL(dec_gemm320_dma4_kernel<true>, dim3(mtiles * splits), dim3(512), D4_LDS_BYTES, s, p);
    else hipLaunchKernelGGL(dec_gemm320_dma4_kernel<false>, dim3(mtiles * splits), dim3(512), D4_LDS_BYTES, s, p);
    return launch_status("spv_gemm_bf16 dma");
  }
  if (a_tiles) {
    if (a_kmajor) { if (nsplit == 3) gemm_dispatch<true, 3, SRC_TILED>(p, splits, s); else gemm_dispatch<true, 1, SRC_TILED>(p, splits, s); }
    else { if (nsplit == 3) gemm_dispatch<false, 3, SRC_TILED>(p, splits, s); else gemm_dispatch<false, 1, SRC_TILED>(p, splits, s); }
  } else {
    if (a_kmajor) { if (nsplit == 3) gemm_dispatch<true, 3, SRC_PLAIN>(p, splits, s); else gemm_dispatch<true, 1, SRC_PLAIN>(p, splits, s); }
    else { if (nsplit == 3) gemm_dispatch<false, 3, SRC_PLAIN>(p, splits, s); else gemm_dispatch<false, 1, SRC_PLAIN>(p, splits, s); }
  }
  return launch_status("spv_gemm_bf16");
}

// both groups' 320-column GEMMs of the decoder backward (same direction, bf16 words, LDS-DMA shapes) as one grid; anything else: per group
static bool gemm_pair_ok(const spv_gemm_args& a) {
  return a.A_hi && a.B_hi && a.C && a.nsplit == 1 && a.M > 0 && a.N > 0 && a.K > 0 && a.splits > 0 &&
         spv_gemm_bf16_uses_dma(a.a_kmajor, a.M, a.N, a.K, a.nsplit, a.a_tiles, a.ldb) &&
         ((reinterpret_cast<uintptr_t>(a.A_hi) | reinterpret_cast<uintptr_t>(a.B_hi)) & 15) == 0;
}
static GemmParams gemm_pair_params(const spv_gemm_args& a) {
  GemmParams p{};
  p.A = a.A_hi; p.lda = a.lda; p.B = a.B_hi; p.ldb = a.ldb; p.C = a.C; p.ldc = a.ldc; p.slab_stride = a.slab_stride;
  p.M = a.M; p.N = a.N; p.K = a.K;
  const int ktiles = (a.K + 63) / 64;
  p.k_per_split = ((ktiles + a.splits - 1) / a.splits) * 64;
  p.epi = EPI_STORE; p.tiles_inner = a.a_tiles; p.c_split_row = a.splits;
  return p;
}
extern "C" int spv_gemm_bf16_grouped(const spv_gemm_args* g, int32_t n, void* stream) {
  if (!g || n <= 0) return fail(SPV_ERR_ARG, "spv_gemm_bf16_grouped: bad arguments%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  for (; i + 1 < n; i += 2) {
    const spv_gemm_args &a = g[i], &b = g[i + 1];
    if (!(gemm_pair_ok(a) && gemm_pair_ok(b) && a.a_kmajor == b.a_kmajor)) break;
    const GemmParams p0 = gemm_pair_params(a), p1 = gemm_pair_params(b);
    const int n0 = (a.M + DG_BM - 1) / DG_BM * a.splits, n1 = (b.M + DG_BM - 1) / DG_BM * b.splits;
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gemm320_dma4_pair_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, D4_LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_gemm320_dma4_pair_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, D4_LDS_BYTES);
      raised = true;
    }
    if (a.a_kmajor) hipLaunchKernelGGL(dec_gemm320_dma4_pair_kernel<true>, dim3(n0 > n1 ? n0 : n1, 2), dim3(512), D4_LDS_BYTES, s, p0, p1);
    else hipLaunchKernelGGL(dec_gemm320_dma4_pair_kernel<false>, dim3(n0 > n1 ? n0 : n1, 2), dim3(512), D4_LDS_BYTES, s, p0, p1);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_gemm_bf16_grouped");
  }
  for (; i < n; ++i) {
    const spv_gemm_args& a = g[i];
    const int rc = spv_gemm_bf16(a.a_kmajor, a.A_hi, a.A_lo, a.lda, a.B_hi, a.B_lo, a.ldb, a.C, a.ldc, a.M, a.N, a.K, a.nsplit, a.splits, a.slab_stride, a.a_tiles, stream);
    if (rc != SPV_OK) return rc;
  }
  return SPV_OK;
}

extern "C" int spv_dec_heads_wgrad(const uint16_t* tP, const uint16_t* tS, int32_t a_tiles, const uint16_t* Aps, int32_t G, int32_t Bp,
                                   int32_t splits, float* slabP, float* slabS, void* stream) {
  if (!tP || !tS || !Aps || !slabP || !slabS) return fail(SPV_ERR_ARG, "spv_dec_heads_wgrad: null pointer%s");
  if (G <= 0 || Bp <= 0 || (Bp % DH_BK) != 0 || splits <= 0 || (long)a_tiles * 32 < ((long)G + DH_BM - 1) / DH_BM * DH_BM)
    return fail(SPV_ERR_ARG, "spv_dec_heads_wgrad: Bp must be a multiple of 64 and the tile arrays must cover round_up(G, 256) genes%s");
  if (((reinterpret_cast<uintptr_t>(tP) | reinterpret_cast<uintptr_t>(tS) | reinterpret_cast<uintptr_t>(Aps)) & 15) != 0)
    return fail(SPV_ERR_ARG, "spv_dec_heads_wgrad: operands must be 16-byte aligned%s");
  static bool raised = false;
  if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_heads_wgrad_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DH_LDS_BYTES); raised = true; }
  const int mtiles = (G + DH_BM - 1) / DH_BM, ktiles = Bp / DH_BK;
  const int k_per_split = (ktiles + splits - 1) / splits * DH_BK;
  hipLaunchKernelGGL(dec_heads_wgrad_dma_kernel, dim3(mtiles * splits), dim3(512), DH_LDS_BYTES, (hipStream_t)stream, (const bf16_t*)tP, (const bf16_t*)tS, a_tiles,
                     (const bf16_t*)Aps, G, Bp, k_per_split, splits, slabP, slabS);
  return launch_status("spv_dec_heads_wgrad");
}

// ---------------------------------------------------------------------------------------------
// decoder / likelihood
// ---------------------------------------------------------------------------------------------
static int to_dec(const spv_dec_params* q, DecParams& p) {
  if (!q) return fail(SPV_ERR_ARG, "decoder: null params%s");
  p.X = q->X; p.ldx = q->ldx; p.rows = q->rows; p.col_off = q->col_off; p.count_is_u16 = q->count_is_u16;
  p.B = q->B; p.G = q->G; p.Bp = q->Bp; p.Gp = q->Gp;
  p.logits = q->logits; p.n_gene_tiles = q->n_gene_tiles; p.logits_f32 = q->logits_f32;
  p.Wps_hi = q->Wps_hi; p.Wps_lo = q->Wps_lo; p.Aps_hi = q->Aps_hi; p.Aps_lo = q->Aps_lo;
  p.gene_tab = (const float4*)q->gene_tab; p.cnt_tab = (const float2*)q->cnt_tab;
  p.a_p = q->a_p; p.a_s = q->a_s; p.lse_p = q->lse_p; p.lse_s = q->lse_s; p.w_row = q->w_row;
  p.gene_splits = q->gene_splits; p.genes_per_split = q->genes_per_split;
  p.part_max_p = q->part_max_p; p.part_sum_p = q->part_sum_p; p.part_max_s = q->part_max_s; p.part_sum_s = q->part_sum_s;
  p.rec_part = q->rec_part; p.tp_part = q->tp_part; p.ts_part = q->ts_part; p.dtheta_part = q->dtheta_part;
  p.dL = q->dL; p.tP = q->tP; p.tS = q->tS; p.grads_f32 = q->grads_f32;
  p.nb_splits = q->nb_splits; p.nb_genes_per_split = q->nb_genes_per_split; p.nb_cell_tiles = q->nb_cell_tiles > 0 ? q->nb_cell_tiles : 1;
  if (p.B <= 0 || p.G <= 0 || p.Bp < p.B || p.Gp < p.G || (p.Bp % DEC_CELLS_PER_WG) || (p.Gp % 32))
    return fail(SPV_ERR_ARG, "decoder: Bp must be a multiple of 128 and Gp of 32%s");
  if ((long)p.Bp * p.Gp * (p.grads_f32 ? 2 : 1) >= (1L << 30) || p.Gp >= (1 << 24))   // the likelihood kernel addresses its [Bp][Gp] arrays (two planes with grads_f32) with 32-bit byte offsets
    return fail(SPV_ERR_ARG, "decoder: Bp * Gp must stay below 2^30 elements (and Gp below 2^24)%s");
  if (p.gene_splits <= 0 || (p.genes_per_split % 32) || (long)p.gene_splits * p.genes_per_split < p.G)
    return fail(SPV_ERR_ARG, "decoder: gene splits must be multiples of 32 covering G%s");
  if (!p.Wps_hi || !p.Wps_lo || !p.Aps_hi || !p.Aps_lo) return fail(SPV_ERR_ARG, "decoder: missing packed regressor operands%s");
  return SPV_OK;
}

extern "C" int spv_dec_tables(const float* px_r, int32_t G, int32_t Gp, void* gene_tab, void* cnt_tab, void* stream) {
  if (!px_r || !gene_tab || !cnt_tab || G <= 0 || Gp < G) return fail(SPV_ERR_ARG, "spv_dec_tables: bad arguments%s");
  hipLaunchKernelGGL(nb_tables_kernel, dim3((Gp + 255) / 256, NB_CMAX), dim3(256), 0, (hipStream_t)stream, px_r, G, Gp,
                     (float4*)gene_tab, (float2*)cnt_tab);
  return launch_status("spv_dec_tables");
}

// ---- the decoder's per-group launches for BOTH groups of a step in one grid each (spv_dec_group; csrc/spv_decoder.h: *_pair_kernel) ------
// Two groups whose launches take the same kernel variant go out as one grid (group = blockIdx.z); anything else falls back to the
// per-group entry points in order.
extern "C" int spv_dec_tables_grouped(const spv_dec_group* g, int32_t n, void* stream) {
  if (!g || n <= 0) return fail(SPV_ERR_ARG, "spv_dec_tables_grouped: bad arguments%s");
  int i = 0;
  for (; i + 1 < n; i += 2) {
    const spv_dec_group &a = g[i], &b = g[i + 1];
    if (!a.px_r || !b.px_r || !a.p.gene_tab || !b.p.gene_tab || !a.p.cnt_tab || !b.p.cnt_tab || a.p.G <= 0 || b.p.G <= 0 || a.p.Gp < a.p.G || b.p.Gp < b.p.G)
      return fail(SPV_ERR_ARG, "spv_dec_tables_grouped: bad arguments%s");
    const NbTabArgs a0{a.px_r, a.p.G, a.p.Gp, (float4*)a.p.gene_tab, (float2*)a.p.cnt_tab}, a1{b.px_r, b.p.G, b.p.Gp, (float4*)b.p.gene_tab, (float2*)b.p.cnt_tab};
    const int gp = a.p.Gp > b.p.Gp ? a.p.Gp : b.p.Gp;
    hipLaunchKernelGGL(nb_tables_pair_kernel, dim3((gp + 255) / 256, NB_CMAX, 2), dim3(256), 0, (hipStream_t)stream, a0, a1);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_dec_tables_grouped");
  }
  for (; i < n; ++i) {
    const int rc = spv_dec_tables(g[i].px_r, g[i].p.G, g[i].p.Gp, const_cast<void*>(g[i].p.gene_tab), const_cast<void*>(g[i].p.cnt_tab), stream);
    if (rc != SPV_OK) return rc;
  }
  return SPV_OK;
}

static int lse_prepare(const spv_dec_params* q, const float* library, DecParams& p) {
  int rc = to_dec(q, p);
  if (rc != SPV_OK) return rc;
  if (!library || !p.part_max_p || !p.part_sum_p || !p.part_max_s || !p.part_sum_s || !p.lse_p || !p.lse_s || !p.a_p || !p.a_s)
    return fail(SPV_ERR_ARG, "spv_dec_lse: null pointer%s");
  return SPV_OK;
}
extern "C" int spv_dec_lse(const spv_dec_params* q, const float* library, void* stream) {
  DecParams p;
  int rc = lse_prepare(q, library, p);
  if (rc != SPV_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dec_lse_kernel, dim3(p.Bp / DEC_CELLS_PER_WG, p.gene_splits), dim3(256), 0, s, p);
  hipLaunchKernelGGL(dec_lse_combine_kernel, dim3((p.Bp + 63) / 64), dim3(256), 0, s, p.part_max_p, p.part_sum_p,
                     p.part_max_s, p.part_sum_s, p.gene_splits, p.Bp, p.B, library, (float*)p.lse_p, (float*)p.lse_s,
                     (float*)p.a_p, (float*)p.a_s);
  return launch_status("spv_dec_lse");
}
extern "C" int spv_dec_lse_grouped(const spv_dec_group* g, int32_t n, void* stream) {
  if (!g || n <= 0) return fail(SPV_ERR_ARG, "spv_dec_lse_grouped: bad arguments%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  for (; i + 1 < n; i += 2) {
    DecParams p0, p1;
    int rc = lse_prepare(&g[i].p, g[i].library, p0);
    if (rc != SPV_OK) return rc;
    rc = lse_prepare(&g[i + 1].p, g[i + 1].library, p1);
    if (rc != SPV_OK) return rc;
    const int bp = p0.Bp > p1.Bp ? p0.Bp : p1.Bp, gs = p0.gene_splits > p1.gene_splits ? p0.gene_splits : p1.gene_splits;
    hipLaunchKernelGGL(dec_lse_pair_kernel, dim3(bp / DEC_CELLS_PER_WG, gs, 2), dim3(256), 0, s, p0, p1);
    const LseCombArgs c0{p0.part_max_p, p0.part_sum_p, p0.part_max_s, p0.part_sum_s, p0.gene_splits, p0.Bp, p0.B, g[i].library, (float*)p0.lse_p, (float*)p0.lse_s, (float*)p0.a_p, (float*)p0.a_s};
    const LseCombArgs c1{p1.part_max_p, p1.part_sum_p, p1.part_max_s, p1.part_sum_s, p1.gene_splits, p1.Bp, p1.B, g[i + 1].library, (float*)p1.lse_p, (float*)p1.lse_s, (float*)p1.a_p, (float*)p1.a_s};
    hipLaunchKernelGGL(dec_lse_combine_pair_kernel, dim3((bp + 63) / 64, 1, 2), dim3(256), 0, s, c0, c1);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_dec_lse_grouped");
  }
  for (; i < n; ++i) {
    const int rc = spv_dec_lse(&g[i].p, g[i].library, stream);
    if (rc != SPV_OK) return rc;
  }
  return SPV_OK;
}

template <bool TRAIN, typename GT, int CM>
static void nb_launch_mode(const DecParams& p, hipStream_t s) {
  const int ctiles = (p.Bp + NB_CELLS_PER_WG - 1) / NB_CELLS_PER_WG;
  dim3 grid((ctiles + p.nb_cell_tiles - 1) / p.nb_cell_tiles, p.nb_splits);
  if (p.logits_f32) hipLaunchKernelGGL((dec_nb_kernel<TRAIN, GT, float, CM>), grid, dim3(256), 0, s, p);
  else hipLaunchKernelGGL((dec_nb_kernel<TRAIN, GT, _Float16, CM>), grid, dim3(256), 0, s, p);
}
template <bool TRAIN, typename GT>
static void nb_launch(const DecParams& p, hipStream_t s) {
  if (!p.count_is_u16) return nb_launch_mode<TRAIN, GT, CNT_F32>(p, s);
  const bool aligned = ((reinterpret_cast<uintptr_t>(p.X) & 7) == 0) && (p.ldx % 4 == 0) && (p.col_off % 4 == 0);
  if (aligned) nb_launch_mode<TRAIN, GT, CNT_U16_ALIGNED>(p, s);
  else nb_launch_mode<TRAIN, GT, CNT_U16_ANY>(p, s);
}

static int nb_prepare(const spv_dec_params* q, int32_t train, DecParams& p) {
  int rc = to_dec(q, p);
  if (rc != SPV_OK) return rc;
  if (!p.X || !p.logits || !p.gene_tab || !p.cnt_tab || !p.a_p || !p.a_s || !p.w_row || !p.rec_part)
    return fail(SPV_ERR_ARG, "spv_dec_nb_fwd: null pointer%s");
  if (p.n_gene_tiles != p.Gp / 32) return fail(SPV_ERR_ARG, "spv_dec_nb_fwd: n_gene_tiles must be Gp / 32%s");
  if (p.nb_splits <= 0 || p.nb_genes_per_split <= 0 || (p.nb_genes_per_split % 32) || p.nb_genes_per_split > NB_GSPL_MAX ||
      (long)p.nb_splits * p.nb_genes_per_split < p.G)
    return fail(SPV_ERR_ARG, "spv_dec_nb_fwd: nb_genes_per_split must be a multiple of 32, at most 160, and the splits must cover G%s");
  if (train && (!p.dL || !p.tP || !p.tS || !p.tp_part || !p.ts_part || !p.dtheta_part))
    return fail(SPV_ERR_ARG, "spv_dec_nb_fwd: training outputs missing%s");
  return SPV_OK;
}
extern "C" int spv_dec_nb_fwd(const spv_dec_params* q, int32_t train, void* stream) {
  DecParams p;
  int rc = nb_prepare(q, train, p);
  if (rc != SPV_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  if (train) {
    if (p.grads_f32) nb_launch<true, split_t>(p, s);
    else nb_launch<true, bf16_t>(p, s);
  } else {
    nb_launch<false, bf16_t>(p, s);
  }
  return launch_status("spv_dec_nb_fwd");
}
static int nb_count_mode(const DecParams& p) {
  if (!p.count_is_u16) return CNT_F32;
  const bool aligned = ((reinterpret_cast<uintptr_t>(p.X) & 7) == 0) && (p.ldx % 4 == 0) && (p.col_off % 4 == 0);
  return aligned ? CNT_U16_ALIGNED : CNT_U16_ANY;
}
template <bool TRAIN, typename GT, int CM>
static void nb_launch_pair_mode(const DecParams& p0, const DecParams& p1, hipStream_t s) {
  auto gx = [](const DecParams& p) { return ((p.Bp + NB_CELLS_PER_WG - 1) / NB_CELLS_PER_WG + p.nb_cell_tiles - 1) / p.nb_cell_tiles; };
  dim3 grid(gx(p0) > gx(p1) ? gx(p0) : gx(p1), p0.nb_splits > p1.nb_splits ? p0.nb_splits : p1.nb_splits, 2);
  if (p0.logits_f32) hipLaunchKernelGGL((dec_nb_pair_kernel<TRAIN, GT, float, CM>), grid, dim3(256), 0, s, p0, p1);
  else hipLaunchKernelGGL((dec_nb_pair_kernel<TRAIN, GT, _Float16, CM>), grid, dim3(256), 0, s, p0, p1);
}
template <bool TRAIN, typename GT>
static void nb_launch_pair(const DecParams& p0, const DecParams& p1, int cm, hipStream_t s) {
  if (cm == CNT_F32) nb_launch_pair_mode<TRAIN, GT, CNT_F32>(p0, p1, s);
  else if (cm == CNT_U16_ALIGNED) nb_launch_pair_mode<TRAIN, GT, CNT_U16_ALIGNED>(p0, p1, s);
  else nb_launch_pair_mode<TRAIN, GT, CNT_U16_ANY>(p0, p1, s);
}
extern "C" int spv_dec_nb_fwd_grouped(const spv_dec_group* g, int32_t n, int32_t train, void* stream) {
  if (!g || n <= 0) return fail(SPV_ERR_ARG, "spv_dec_nb_fwd_grouped: bad arguments%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  for (; i + 1 < n; i += 2) {
    DecParams p0, p1;
    int rc = nb_prepare(&g[i].p, train, p0);
    if (rc != SPV_OK) return rc;
    rc = nb_prepare(&g[i + 1].p, train, p1);
    if (rc != SPV_OK) return rc;
    // one grid needs one kernel variant: count storage, logits type and gradient word type of both groups must agree
    if (nb_count_mode(p0) != nb_count_mode(p1) || p0.logits_f32 != p1.logits_f32 || p0.grads_f32 != p1.grads_f32) break;
    const int cm = nb_count_mode(p0);
    if (train) {
      if (p0.grads_f32) nb_launch_pair<true, split_t>(p0, p1, cm, s);
      else nb_launch_pair<true, bf16_t>(p0, p1, cm, s);
    } else {
      nb_launch_pair<false, bf16_t>(p0, p1, cm, s);
    }
    if (hipGetLastError() != hipSuccess) return launch_status("spv_dec_nb_fwd_grouped");
  }
  for (; i < n; ++i) {
    const int rc = spv_dec_nb_fwd(&g[i].p, train, stream);
    if (rc != SPV_OK) return rc;
  }
  return SPV_OK;
}

extern "C" int spv_dec_logits(const uint16_t* Am_hi, const uint16_t* Am_lo, const uint16_t* Wm_hi, const uint16_t* Wm_lo,
                              int32_t K, int32_t Bp, int32_t Gp, int32_t nsplit, void* out, int32_t out_f32, void* stream) {
  if (!Am_hi || !Wm_hi || !out) return fail(SPV_ERR_ARG, "spv_dec_logits: null pointer%s");
  if (K <= 0 || (K % 32) || Bp <= 0 || Gp <= 0 || (Bp % 128) || (Gp % 128)) return fail(SPV_ERR_ARG, "spv_dec_logits: bad shape%s");
  if (nsplit != 1 && nsplit != 3) return fail(SPV_ERR_ARG, "spv_dec_logits: nsplit must be 1 or 3%s");
  if (nsplit == 3 && (!Am_lo || !Wm_lo)) return fail(SPV_ERR_ARG, "spv_dec_logits: nsplit=3 needs lo images%s");
  // genes on MFMA rows (M), cells on MFMA columns (N): the orientation the likelihood kernel consumes
  GemmParams p{};
  p.A = Wm_hi; p.A_lo = Wm_lo; p.lda = K;
  p.B = Am_hi; p.B_lo = Am_lo; p.ldb = K;
  p.C = (float*)out; p.ldc = 0; p.slab_stride = 0;
  p.M = Gp; p.N = Bp; p.K = K; p.k_per_split = K;
  p.epi = out_f32 ? EPI_TILED_F32 : EPI_TILED_F16;
  p.tiles_inner = Gp / 32;
  hipStream_t s = (hipStream_t)stream;
  static const int logits_dma = getenv("SPV_LOGITS_DMA") ? atoi(getenv("SPV_LOGITS_DMA")) : 1;   // (A/B switch)
  if (logits_dma && nsplit == 1 && !out_f32 && Gp % DL_BM == 0 && Bp % DL_BN == 0 && ((reinterpret_cast<uintptr_t>(Am_hi) | reinterpret_cast<uintptr_t>(Wm_hi)) & 15) == 0) {
    // LDS-DMA kernel (spv_dec_gemm.h): 256 x 128 tiles, two workgroups per CU
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_logits_dma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DL_LDS_BYTES); raised = true; }
    hipLaunchKernelGGL(dec_logits_dma_kernel, dim3(Gp / DL_BM, Bp / DL_BN), dim3(512), DL_LDS_BYTES, s, (const bf16_t*)Wm_hi, (const bf16_t*)Am_hi, K, Gp / 32, (_Float16*)out);
    return launch_status("spv_dec_logits dma");
  }
  if (nsplit == 3) launch_gemm<GemmCfg<128, 128, 2, 2, false, false, SRC_PLAIN, SRC_PLAIN, unsigned short, 3, 32, 4>>(p, 1, s);
  else if (Gp % 256 == 0) launch_gemm<GemmCfg<256, 128, 2, 2, false, false, SRC_PLAIN, SRC_PLAIN, unsigned short, 1, 32, 2, 2>>(p, 1, s);  // 128 x 64 wave tiles: 25 % less LDS traffic per MFMA
  else launch_gemm<GemmCfg<128, 128, 2, 2, false, false, SRC_PLAIN, SRC_PLAIN, unsigned short, 1, 32, 4>>(p, 1, s);
  return launch_status("spv_dec_logits");
}

static bool logits_dma_ok(const spv_dec_group& a) {
  static const int logits_dma = getenv("SPV_LOGITS_DMA") ? atoi(getenv("SPV_LOGITS_DMA")) : 1;
  return logits_dma && a.Am_hi && a.Wm_hi && a.p.logits && a.nsplit == 1 && !a.p.logits_f32 && a.K > 0 && (a.K % 32) == 0 && a.p.Bp > 0 && a.p.Gp > 0 &&
         a.p.Gp % DL_BM == 0 && a.p.Bp % DL_BN == 0 && ((reinterpret_cast<uintptr_t>(a.Am_hi) | reinterpret_cast<uintptr_t>(a.Wm_hi)) & 15) == 0;
}
extern "C" int spv_dec_logits_grouped(const spv_dec_group* g, int32_t n, void* stream) {
  if (!g || n <= 0) return fail(SPV_ERR_ARG, "spv_dec_logits_grouped: bad arguments%s");
  int i = 0;
  for (; i + 1 < n; i += 2) {
    const spv_dec_group &a = g[i], &b = g[i + 1];
    if (!(logits_dma_ok(a) && logits_dma_ok(b))) break;
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_logits_dma_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DL_LDS_BYTES); raised = true; }
    const LogitsArgs a0{(const bf16_t*)a.Wm_hi, (const bf16_t*)a.Am_hi, a.K, a.p.Gp / 32, a.p.Bp, (_Float16*)const_cast<void*>(a.p.logits)};
    const LogitsArgs a1{(const bf16_t*)b.Wm_hi, (const bf16_t*)b.Am_hi, b.K, b.p.Gp / 32, b.p.Bp, (_Float16*)const_cast<void*>(b.p.logits)};
    const int gp = a.p.Gp > b.p.Gp ? a.p.Gp : b.p.Gp, bp = a.p.Bp > b.p.Bp ? a.p.Bp : b.p.Bp;
    hipLaunchKernelGGL(dec_logits_dma_pair_kernel, dim3(gp / DL_BM, bp / DL_BN, 2), dim3(512), DL_LDS_BYTES, (hipStream_t)stream, a0, a1);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_dec_logits_grouped");
  }
  for (; i < n; ++i) {
    const spv_dec_group& a = g[i];
    const int rc = spv_dec_logits(a.Am_hi, a.Am_lo, a.Wm_hi, a.Wm_lo, a.K, a.p.Bp, a.p.Gp, a.nsplit, const_cast<void*>(a.p.logits), a.p.logits_f32, stream);
    if (rc != SPV_OK) return rc;
  }
  return SPV_OK;
}

extern "C" int spv_dec_materialize(const spv_dec_params* q, float* scale_p, float* scale_s, float* rate_p, float* rate_s,
                                   float* logits_out, int64_t ld, void* stream) {
  DecParams p;
  int rc = to_dec(q, p);
  if (rc != SPV_OK) return rc;
  if (!scale_p || !scale_s || !rate_p || !rate_s || !logits_out || ld < p.G || !p.logits || !p.lse_p || !p.lse_s || !p.a_p || !p.a_s)
    return fail(SPV_ERR_ARG, "spv_dec_materialize: null pointer / ld < G%s");
  if (p.n_gene_tiles != p.Gp / 32) return fail(SPV_ERR_ARG, "spv_dec_materialize: n_gene_tiles must be Gp / 32%s");
  dim3 grid(p.Bp / DEC_CELLS_PER_WG, p.gene_splits);
  hipStream_t s = (hipStream_t)stream;
  if (p.logits_f32) hipLaunchKernelGGL((dec_materialize_kernel<float>), grid, dim3(256), 0, s, p, scale_p, scale_s, rate_p, rate_s, logits_out, (long)ld);
  else hipLaunchKernelGGL((dec_materialize_kernel<_Float16>), grid, dim3(256), 0, s, p, scale_p, scale_s, rate_p, rate_s, logits_out, (long)ld);
  return launch_status("spv_dec_materialize");
}

extern "C" int spv_dec_dz(const spv_dec_params* q, const float* Tp, const float* Ts, float* dz_part, void* stream) {
  DecParams p;
  int rc = to_dec(q, p);
  if (rc != SPV_OK) return rc;
  if (!Tp || !Ts || !p.tP || !p.tS || !p.lse_p || !p.lse_s || !dz_part) return fail(SPV_ERR_ARG, "spv_dec_dz: null pointer%s");
  if (p.n_gene_tiles != p.Gp / 32) return fail(SPV_ERR_ARG, "spv_dec_dz: n_gene_tiles must be Gp / 32%s");
  if (p.grads_f32) return fail(SPV_ERR_UNSUPPORTED, "spv_dec_dz: needs bf16 gradient arrays%s");
  hipLaunchKernelGGL((dec_softmax_bwd_kernel<bf16_t, true, false>), dim3(p.Bp / DEC_CELLS_PER_WG, p.gene_splits), dim3(256), 0, (hipStream_t)stream, p, Tp, Ts, dz_part, (float*)nullptr, (float*)nullptr);
  return launch_status("spv_dec_dz");
}

extern "C" int spv_dec_softmax_bwd(const spv_dec_params* q, const float* Tp, const float* Ts, float* dz_part, void* stream) {
  DecParams p;
  int rc = to_dec(q, p);
  if (rc != SPV_OK) return rc;
  if (!Tp || !Ts || !p.tP || !p.tS || !p.lse_p || !p.lse_s) return fail(SPV_ERR_ARG, "spv_dec_softmax_bwd: null pointer%s");
  if (p.n_gene_tiles != p.Gp / 32) return fail(SPV_ERR_ARG, "spv_dec_softmax_bwd: n_gene_tiles must be Gp / 32%s");
  dim3 grid(p.Bp / DEC_CELLS_PER_WG, p.gene_splits);
  hipStream_t s = (hipStream_t)stream;
  float* const none = nullptr;
  if (p.grads_f32 && dz_part) hipLaunchKernelGGL((dec_softmax_bwd_kernel<split_t, true>), grid, dim3(256), 0, s, p, Tp, Ts, dz_part, none, none);
  else if (p.grads_f32) hipLaunchKernelGGL((dec_softmax_bwd_kernel<split_t, false>), grid, dim3(256), 0, s, p, Tp, Ts, none, none, none);
  else if (dz_part) hipLaunchKernelGGL((dec_softmax_bwd_kernel<bf16_t, true>), grid, dim3(256), 0, s, p, Tp, Ts, dz_part, none, none);
  else hipLaunchKernelGGL((dec_softmax_bwd_kernel<bf16_t, false>), grid, dim3(256), 0, s, p, Tp, Ts, none, none, none);
  return launch_status("spv_dec_softmax_bwd");
}

// One read-only pass over t_P / t_S (bf16 gradient arrays) for everything the backward pass needs from them: the latent gradient of the
// two rate heads (dz_part, as spv_dec_softmax_bwd) AND the two regressor weight gradients as per-workgroup-row partial slabs
// dw_part [Bp / 128][Gp][48] = [d W'_p (16 columns) | d W'_s (32 columns)] (sum the first G rows with spv_reduce_slabs).  t_P / t_S are left uncorrected: nothing reads
// them afterwards.  Replaces spv_dec_softmax_bwd + spv_dec_heads_wgrad in bf16 mode.
static int heads_bwd_prepare(const spv_dec_params* q, const float* Tp, const float* Ts, const float* dz_part, const float* dw_part, DecParams& p) {
  int rc = to_dec(q, p);
  if (rc != SPV_OK) return rc;
  if (!Tp || !Ts || !p.tP || !p.tS || !p.lse_p || !p.lse_s || !dz_part || !dw_part) return fail(SPV_ERR_ARG, "spv_dec_heads_bwd: null pointer%s");
  if (p.n_gene_tiles != p.Gp / 32) return fail(SPV_ERR_ARG, "spv_dec_heads_bwd: n_gene_tiles must be Gp / 32%s");
  if (p.Bp % DEC_CELLS_PER_WG) return fail(SPV_ERR_ARG, "spv_dec_heads_bwd: Bp must be a multiple of 128%s");
  if (p.grads_f32 && (!p.Wps_lo || !p.Aps_lo)) return fail(SPV_ERR_ARG, "spv_dec_heads_bwd: split gradient words need the lo planes of the operand images%s");
  return SPV_OK;
}
extern "C" int spv_dec_heads_bwd(const spv_dec_params* q, const float* Tp, const float* Ts, float* dz_part, float* dw_part, void* stream) {
  DecParams p;
  int rc = heads_bwd_prepare(q, Tp, Ts, dz_part, dw_part, p);
  if (rc != SPV_OK) return rc;
  dim3 grid(p.Bp / DEC_CELLS_PER_WG, p.gene_splits);
  if (p.grads_f32) {   // hi / lo planes of split-bf16 gradient words ("fp32" mode)
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dec_heads_bwd_split_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, HBS_LDS_BYTES); raised = true; }
    hipLaunchKernelGGL(dec_heads_bwd_split_kernel, grid, dim3(256), HBS_LDS_BYTES, (hipStream_t)stream, p, Tp, Ts, dz_part, dw_part);
    return launch_status("spv_dec_heads_bwd");
  }
  // (the partial slabs cover gene rows the splits do not reach only when genes_per_split * gene_splits < round_up(G, 32): the host sizes
  // the splits to cover; rows in [round_up(G, 32), Gp) are never written and never read)
  hipLaunchKernelGGL(dec_heads_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, p, Tp, Ts, dz_part, dw_part);
  return launch_status("spv_dec_heads_bwd");
}
extern "C" int spv_dec_heads_bwd_grouped(const spv_dec_group* g, int32_t n, void* stream) {
  if (!g || n <= 0) return fail(SPV_ERR_ARG, "spv_dec_heads_bwd_grouped: bad arguments%s");
  int i = 0;
  for (; i + 1 < n; i += 2) {
    HeadsBwdArgs a0, a1;
    int rc = heads_bwd_prepare(&g[i].p, g[i].Tp, g[i].Ts, g[i].dz_part, g[i].dw_part, a0.p);
    if (rc != SPV_OK) return rc;
    rc = heads_bwd_prepare(&g[i + 1].p, g[i + 1].Tp, g[i + 1].Ts, g[i + 1].dz_part, g[i + 1].dw_part, a1.p);
    if (rc != SPV_OK) return rc;
    if (a0.p.grads_f32 || a1.p.grads_f32) break;   // (the split-word kernel takes a whole CU's LDS: no pair form)
    a0.Tp = g[i].Tp; a0.Ts = g[i].Ts; a0.dz_part = g[i].dz_part; a0.dw_part = g[i].dw_part;
    a1.Tp = g[i + 1].Tp; a1.Ts = g[i + 1].Ts; a1.dz_part = g[i + 1].dz_part; a1.dw_part = g[i + 1].dw_part;
    const int bp = a0.p.Bp > a1.p.Bp ? a0.p.Bp : a1.p.Bp, gs = a0.p.gene_splits > a1.p.gene_splits ? a0.p.gene_splits : a1.p.gene_splits;
    hipLaunchKernelGGL(dec_heads_bwd_pair_kernel, dim3(bp / DEC_CELLS_PER_WG, gs, 2), dim3(256), 0, (hipStream_t)stream, a0, a1);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_dec_heads_bwd_grouped");
  }
  for (; i < n; ++i) {
    const int rc = spv_dec_heads_bwd(&g[i].p, g[i].Tp, g[i].Ts, g[i].dz_part, g[i].dw_part, stream);
    if (rc != SPV_OK) return rc;
  }
  return SPV_OK;
}

// ---------------------------------------------------------------------------------------------
// optimiser: Adam over one flat fp32 parameter buffer (scvi TrainingPlan defaults are passed by the
// host: lr 1e-3, eps 0.01, weight_decay 1e-6 -- training_mixin.py:111).  torch.optim.Adam semantics.
// ---------------------------------------------------------------------------------------------
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float gscale) {
  const long stride = (long)gridDim.x * blockDim.x * 4;
  const float step = lr / bc1, isq = rsqrtf(bc2);
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    if (i + 4 <= n) {
      f4v pp = *reinterpret_cast<f4v*>(p + i), gg = *reinterpret_cast<const f4v*>(g + i);
      f4v mm = *reinterpret_cast<f4v*>(m + i), vv = *reinterpret_cast<f4v*>(v + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gr = gg[j] * gscale + wd * pp[j];
        mm[j] = b1 * mm[j] + (1.f - b1) * gr;
        vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
        pp[j] -= step * mm[j] / (sqrtf(vv[j]) * isq + eps);
      }
      *reinterpret_cast<f4v*>(p + i) = pp; *reinterpret_cast<f4v*>(m + i) = mm; *reinterpret_cast<f4v*>(v + i) = vv;
    } else {
      for (long k = i; k < n; ++k) {
        const float gr = g[k] * gscale + wd * p[k];
        m[k] = b1 * m[k] + (1.f - b1) * gr;
        v[k] = b2 * v[k] + (1.f - b2) * gr * gr;
        p[k] -= step * m[k] / (sqrtf(v[k]) * isq + eps);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// several small gathers / copies of 4-byte words in one launch: dst[i] = src[idx ? idx[i] : i].  The step's prologue (two row-index
// copies into the graph's static buffers, two label gathers) is a chain of 5-us launches at the head of the critical path.
// ---------------------------------------------------------------------------------------------
struct GatherBatch { int n; spv_gather_prob p[SPV_MAXP]; };
__global__ __launch_bounds__(256) void gather_u32_kernel(GatherBatch a) {
  const spv_gather_prob& q = a.p[blockIdx.y];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < q.count) q.dst[i] = q.src[q.idx ? (long)q.idx[i] : i];
}
extern "C" int spv_gather_u32(const spv_gather_prob* probs, int32_t nprob, void* stream) {
  if (!probs || nprob <= 0 || nprob > SPV_MAXP) return fail(SPV_ERR_ARG, "spv_gather_u32: bad arguments%s");
  GatherBatch a{};
  a.n = nprob;
  long nmax = 0;
  for (int i = 0; i < nprob; ++i) {
    if (!probs[i].src || !probs[i].dst || probs[i].count < 0) return fail(SPV_ERR_ARG, "spv_gather_u32: bad problem%s");
    a.p[i] = probs[i];
    nmax = probs[i].count > nmax ? probs[i].count : nmax;
  }
  if (nmax == 0) return SPV_OK;
  hipLaunchKernelGGL(gather_u32_kernel, dim3((unsigned)((nmax + 255) / 256), nprob), dim3(256), 0, (hipStream_t)stream, a);
  return launch_status("spv_gather_u32");
}

// ---------------------------------------------------------------------------------------------
// standard-normal draws from a counter-based generator (Philox 4x32-10 + Box-Muller), keyed by (key, *counter): the step's noise
// comes out of ONE launch that reads a device-resident step counter, so a captured hipGraph needs no generator state from the host
// (torch's graph-safe generator costs two fill launches per replay + its own kernel, at the head of the step's critical path).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
  const unsigned long long p0 = (unsigned long long)0xD2511F53u * c[0], p1 = (unsigned long long)0xCD9E8D57u * c[2];
  const unsigned n0 = (unsigned)(p1 >> 32) ^ c[1] ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c[3] ^ k1, n3 = (unsigned)p0;
  c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
__global__ __launch_bounds__(256) void randn_kernel(float* __restrict__ out, long n, const long long* __restrict__ counter, unsigned long long key) {
  const long i4 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  if (i4 >= n) return;
  const unsigned long long step = counter ? (unsigned long long)*counter : 0ull;
  unsigned c[4] = {(unsigned)(i4 >> 2), (unsigned)((unsigned long long)(i4 >> 2) >> 32), (unsigned)step, (unsigned)(step >> 32)};
  unsigned k0 = (unsigned)key, k1 = (unsigned)(key >> 32);
#pragma unroll
  for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
  float z[4];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const float u1 = ((float)(c[2 * h] >> 8) + 0.5f) * (1.0f / 16777216.0f);      // (0, 1)
    const float u2 = ((float)(c[2 * h + 1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
    const float rad = sqrtf(-2.0f * __logf(u1));
    float sn, cs;
    __sincosf(6.283185307179586f * u2, &sn, &cs);
    z[2 * h] = rad * cs; z[2 * h + 1] = rad * sn;
  }
  if (i4 + 4 <= n && ((reinterpret_cast<uintptr_t>(out + i4) & 15) == 0)) *reinterpret_cast<f4v*>(out + i4) = f4v{z[0], z[1], z[2], z[3]};
  else for (int j = 0; j < 4 && i4 + j < n; ++j) out[i4 + j] = z[j];
}
__global__ void counter_bump_kernel(long long* counter) { if (threadIdx.x == 0 && blockIdx.x == 0) *counter += 1; }

extern "C" int spv_counter_bump(int64_t* counter, void* stream) {
  if (!counter) return fail(SPV_ERR_ARG, "spv_counter_bump: null counter%s");
  hipLaunchKernelGGL(counter_bump_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (long long*)counter);
  return launch_status("spv_counter_bump");
}

extern "C" int spv_randn(float* out, int64_t n, const int64_t* counter, uint64_t key, void* stream) {
  if (!out || n < 0) return fail(SPV_ERR_ARG, "spv_randn: bad arguments%s");
  if (n == 0) return SPV_OK;
  hipLaunchKernelGGL(randn_kernel, dim3((unsigned)((n + 1023) / 1024)), dim3(256), 0, (hipStream_t)stream, out, (long)n, (const long long*)counter, (unsigned long long)key);
  return launch_status("spv_randn");
}

// Adam that also refreshes the bf16 operand images of the matrices it has just updated (the fc1 weights of both encoders, the
// mixing head's [W_m | b_m]): the step that follows then starts without its spv_pack_bf16 launches.  Same arithmetic as
// adam_kernel; per 4-element chunk one range test per image (<= SPV_ADAM_MAX_IMAGES, all warp-uniform but at range borders).
struct AdamImages { int n; spv_adam_image img[SPV_ADAM_MAX_IMAGES]; };

__global__ void adam_images_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                   long n, float lr, float b1, float b2, float eps, float wd, float bc1, float bc2, float gscale, AdamImages im,
                                   long long* step_counter, const long long* t_dev, double b1d, double b2d) {
  if (step_counter && blockIdx.x == 0 && threadIdx.x == 0) *step_counter += 1;   // (read by the NEXT step's spv_randn / dropout: stream order)
  if (t_dev) {
    // bias corrections 1 - beta^t from the DEVICE-resident count of completed steps (t = *t_dev + 1), in double like torch.optim.Adam's
    // host arithmetic: the launch then carries no per-step host value and can sit inside the captured graph of the step.  Nobody
    // writes *t_dev while this kernel runs (the caller advances it with spv_counter_bump afterwards).  beta^t by repeated squaring.
    long long e = *t_dev + 1;
    double r1 = 1.0, r2 = 1.0, x1 = b1d, x2 = b2d;
    while (e > 0) { if (e & 1) { r1 *= x1; r2 *= x2; } x1 *= x1; x2 *= x2; e >>= 1; }
    bc1 = (float)(1.0 - r1); bc2 = (float)(1.0 - r2);
  }
  const long stride = (long)gridDim.x * blockDim.x * 4;
  const float step = lr / bc1, isq = rsqrtf(bc2);
  for (long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 4; i < n; i += stride) {
    float out[4];
    const int cnt = (i + 4 <= n) ? 4 : (int)(n - i);
    if (cnt == 4) {
      f4v pp = *reinterpret_cast<f4v*>(p + i), gg = *reinterpret_cast<const f4v*>(g + i);
      f4v mm = *reinterpret_cast<f4v*>(m + i), vv = *reinterpret_cast<f4v*>(v + i);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float gr = gg[j] * gscale + wd * pp[j];
        mm[j] = b1 * mm[j] + (1.f - b1) * gr;
        vv[j] = b2 * vv[j] + (1.f - b2) * gr * gr;
        pp[j] -= step * mm[j] / (sqrtf(vv[j]) * isq + eps);
        out[j] = pp[j];
      }
      *reinterpret_cast<f4v*>(p + i) = pp; *reinterpret_cast<f4v*>(m + i) = mm; *reinterpret_cast<f4v*>(v + i) = vv;
    } else {
      for (int j = 0; j < cnt; ++j) {
        const long k = i + j;
        const float gr = g[k] * gscale + wd * p[k];
        m[k] = b1 * m[k] + (1.f - b1) * gr;
        v[k] = b2 * v[k] + (1.f - b2) * gr * gr;
        p[k] -= step * m[k] / (sqrtf(v[k]) * isq + eps);
        out[j] = p[k];
      }
    }
    for (int d = 0; d < im.n; ++d) {
      const spv_adam_image& q = im.img[d];
      const long k0 = i - q.begin;
      if (k0 + cnt <= 0 || k0 >= q.count) continue;
      // (parameters start on 4-element boundaries of the flat buffer, so k0 >= 0 here; count need not be a multiple of 4)
      const unsigned cols = (unsigned)q.cols;
      unsigned r = (unsigned)k0 / cols, c = (unsigned)k0 - r * cols;
      uint16_t* dst = q.dst + ((long)r + q.row_off) * q.ld + q.col_off;
      const bool half = q.fmt == SPV_IMAGE_F16;   // (uniform per image: the fc1 weight images are f16 words of W * scale, spv_common.h)
      if (cnt == 4 && k0 + 4 <= q.count && c + 4 <= cols && (((reinterpret_cast<uintptr_t>(dst + c)) & 7) == 0)) {
        typedef __attribute__((ext_vector_type(4))) unsigned short us4;
        *reinterpret_cast<us4*>(dst + c) = half ? us4{f2h(out[0] * q.scale), f2h(out[1] * q.scale), f2h(out[2] * q.scale), f2h(out[3] * q.scale)}
                                                : us4{f2bf(out[0]), f2bf(out[1]), f2bf(out[2]), f2bf(out[3])};
      } else {
        for (int j = 0; j < cnt && k0 + j < q.count; ++j) {
          dst[c] = half ? f2h(out[j] * q.scale) : f2bf(out[j]);
          if (++c == cols) { c = 0; dst += q.ld; }
        }
      }
      break;   // ranges are disjoint
    }
  }
}

extern "C" int spv_adam_step_images(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                                    float weight_decay, float bc1, float bc2, float grad_scale, const spv_adam_image* images, int32_t n_images,
                                    int64_t* step_counter, const int64_t* t_dev, double beta1_d, double beta2_d, void* stream) {
  if (!p || !g || !m || !v || n < 0 || n_images < 0 || n_images > SPV_ADAM_MAX_IMAGES || (n_images && !images)) return fail(SPV_ERR_ARG, "spv_adam_step_images: bad arguments%s");
  if (n == 0) return SPV_OK;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15)
    return fail(SPV_ERR_ARG, "spv_adam_step_images: buffers must be 16-byte aligned%s");
  AdamImages im{};
  im.n = n_images;
  for (int d = 0; d < n_images; ++d) {
    const spv_adam_image& q = images[d];
    if (!q.dst || q.begin < 0 || (q.begin & 3) || q.count <= 0 || q.begin + q.count > n || q.cols <= 0 || q.ld < q.cols + q.col_off || q.row_off < 0 || q.col_off < 0 ||
        q.count >= (1L << 31) || (q.fmt != SPV_IMAGE_BF16 && q.fmt != SPV_IMAGE_F16) || (q.fmt == SPV_IMAGE_F16 && !(q.scale > 0.f)))
      return fail(SPV_ERR_ARG, "spv_adam_step_images: bad image descriptor%s");
    im.img[d] = q;
  }
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_images_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2, grad_scale, im, (long long*)step_counter, (const long long*)t_dev, beta1_d, beta2_d);
  return launch_status("spv_adam_step_images");
}

extern "C" int spv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, float weight_decay, float bc1, float bc2, float grad_scale, void* stream) {
  if (!p || !g || !m || !v || n < 0) return fail(SPV_ERR_ARG, "spv_adam_step: bad arguments%s");
  if (n == 0) return SPV_OK;
  if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15)
    return fail(SPV_ERR_ARG, "spv_adam_step: buffers must be 16-byte aligned%s");
  long blocks = (n / 4 + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n, lr, beta1, beta2, eps,
                     weight_decay, bc1, bc2, grad_scale);
  return launch_status("spv_adam_step");
}

// ---------------------------------------------------------------------------------------------
// small dense primitives (spv_small.h)
// ---------------------------------------------------------------------------------------------
static int check_linear(const spv_linear_batch* a, const char* who) {
  if (!a || a->nprob <= 0 || a->nprob > SPV_MAXP || a->B <= 0) return fail(SPV_ERR_ARG, "%s: bad batch", who);
  if (a->drop_p < 0.f || a->drop_p >= 1.f) return fail(SPV_ERR_ARG, "%s: dropout probability outside [0,1)", who);
  for (int i = 0; i < a->nprob; ++i)
    if (a->p[i].N <= 0 || a->p[i].K <= 0 || !a->p[i].W) return fail(SPV_ERR_ARG, "%s: bad problem", who);
  return SPV_OK;
}
static void linear_extents(const spv_linear_batch* a, int& nmax, int& kmax) {
  nmax = kmax = 0;
  for (int i = 0; i < a->nprob; ++i) { nmax = a->p[i].N > nmax ? a->p[i].N : nmax; kmax = a->p[i].K > kmax ? a->p[i].K : kmax; }
}

extern "C" int spv_linear_fwd(const spv_linear_batch* a, void* stream) {
  int rc = check_linear(a, "spv_linear_fwd");
  if (rc) return rc;
  for (int i = 0; i < a->nprob; ++i) if (!a->p[i].X || !a->p[i].Y) return fail(SPV_ERR_ARG, "spv_linear_fwd: null pointer%s");
  int nmax, kmax; linear_extents(a, nmax, kmax);
  int rows = a->B;
  for (int i = 0; i < a->nprob; ++i) {
    const spv_linear_prob& q = a->p[i];
    if (q.img_hi && (q.ld_img < q.N || q.img_rows < 0)) return fail(SPV_ERR_ARG, "spv_linear_fwd: bad operand image%s");
    if (q.img_hi && q.img_rows > rows) rows = q.img_rows;   // the image's padding rows are zero filled by the same launch
  }
  hipLaunchKernelGGL(linear_fwd_kernel, dim3((rows + 63) / 64, (nmax + 63) / 64, a->nprob), dim3(256), 0, (hipStream_t)stream, *a);
  return launch_status("spv_linear_fwd");
}
extern "C" int spv_linear_dgrad(const spv_linear_batch* a, void* stream) {
  int rc = check_linear(a, "spv_linear_dgrad");
  if (rc) return rc;
  for (int i = 0; i < a->nprob; ++i)
    if (!a->p[i].dY || !a->p[i].dX || ((a->relu || a->drop_p > 0.f) && !a->p[i].Y)) return fail(SPV_ERR_ARG, "spv_linear_dgrad: null pointer%s");
  int nmax, kmax; linear_extents(a, nmax, kmax);
  hipLaunchKernelGGL(linear_dgrad_kernel, dim3((a->B + 63) / 64, (kmax + 63) / 64, a->nprob), dim3(256), 0, (hipStream_t)stream, *a);
  return launch_status("spv_linear_dgrad");
}
extern "C" int spv_linear_wgrad(const spv_linear_batch* a, float* wpart, int64_t wpart_elems, void* stream) {
  int rc = check_linear(a, "spv_linear_wgrad");
  if (rc) return rc;
  for (int i = 0; i < a->nprob; ++i)
    if (!a->p[i].dY || !a->p[i].X || !a->p[i].dW || ((a->relu || a->drop_p > 0.f) && !a->p[i].Y)) return fail(SPV_ERR_ARG, "spv_linear_wgrad: null pointer%s");
  if (!wpart) return fail(SPV_ERR_ARG, "spv_linear_wgrad: workspace missing%s");
  int nmax, kmax; linear_extents(a, nmax, kmax);
  const long prob_stride = (long)WG_SLICES * nmax * (kmax + 1);
  if (wpart_elems < prob_stride * a->nprob) return fail(SPV_ERR_ARG, "spv_linear_wgrad: workspace too small%s");
  hipStream_t s = (hipStream_t)stream;
  if (a->B <= WG_DIRECT_MAX_B) {   // one slice holds the whole minibatch: results straight into dW / db
    hipLaunchKernelGGL(linear_wgrad_direct_kernel, dim3((nmax + 31) / 32, (kmax + 31) / 32, a->nprob), dim3(256), 0, s, *a);
    return launch_status("spv_linear_wgrad");
  }
  hipLaunchKernelGGL(linear_wgrad_kernel, dim3((nmax + 31) / 32, (kmax + 31) / 32, a->nprob * WG_SLICES), dim3(256), 0, s, *a, wpart, prob_stride);
  hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)(((long)nmax * (kmax + 1) + 255) / 256), a->nprob), dim3(256), 0, s, *a, wpart, prob_stride);
  return launch_status("spv_linear_wgrad");
}

static int check_bn(const spv_bn_batch* a, const char* who) {
  if (!a || a->nprob <= 0 || a->nprob > SPV_MAXP || a->B <= 0) return fail(SPV_ERR_ARG, "%s: bad batch", who);
  for (int i = 0; i < a->nprob; ++i) {
    const spv_bn_prob& q = a->p[i];
    if (q.N <= 0 || q.N > 256 || !q.X || !q.gamma || !q.beta || !q.running_mean || !q.running_var || !q.stats || !q.part)
      return fail(SPV_ERR_ARG, "%s: bad problem (N <= 256, non-null pointers)", who);
  }
  return SPV_OK;
}
static bool bn_all_narrow(const spv_bn_batch* a) {   // every problem N <= 64: the consumers finalise the statistics themselves
  for (int i = 0; i < a->nprob; ++i) if (a->p[i].N > 64) return false;
  return true;
}
extern "C" int spv_bn_fwd(const spv_bn_batch* a, void* stream) {
  int rc = check_bn(a, "spv_bn_fwd");
  if (rc) return rc;
  for (int i = 0; i < a->nprob; ++i) if (!a->p[i].Y) return fail(SPV_ERR_ARG, "spv_bn_fwd: null output%s");
  dim3 grid((a->B + BN_ROWS - 1) / BN_ROWS, a->nprob);
  hipStream_t s = (hipStream_t)stream;
  int nmax = 0;
  for (int i = 0; i < a->nprob; ++i) nmax = a->p[i].N > nmax ? a->p[i].N : nmax;
  if (a->training) hipLaunchKernelGGL(bn_stats_kernel, grid, dim3(256), 0, s, *a);
  int rmax = a->B;
  for (int i = 0; i < a->nprob; ++i) if (a->p[i].img_hi && a->p[i].img_rows > rmax) rmax = a->p[i].img_rows;
  const dim3 agrid((rmax + BN_ROWS - 1) / BN_ROWS, a->nprob);
  if (bn_all_narrow(a)) {   // two launches: every workgroup of the second re-derives its columns' statistics from the partials
    hipLaunchKernelGGL(bn_apply_fin_kernel, agrid, dim3(256), 0, s, *a);
  } else {
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((nmax + 63) / 64, a->nprob), dim3(256), 0, s, *a);
    hipLaunchKernelGGL(bn_apply_kernel, agrid, dim3(256), 0, s, *a);
  }
  return launch_status("spv_bn_fwd");
}
extern "C" int spv_bn_bwd(const spv_bn_batch* a, void* stream) {
  int rc = check_bn(a, "spv_bn_bwd");
  if (rc) return rc;
  for (int i = 0; i < a->nprob; ++i)
    if (!a->p[i].dY || !a->p[i].dX || !a->p[i].dgamma || !a->p[i].dbeta || (a->relu && !a->p[i].Y)) return fail(SPV_ERR_ARG, "spv_bn_bwd: null pointer%s");
  dim3 grid((a->B + BN_ROWS - 1) / BN_ROWS, a->nprob);
  hipStream_t s = (hipStream_t)stream;
  int nmax = 0;
  for (int i = 0; i < a->nprob; ++i) nmax = a->p[i].N > nmax ? a->p[i].N : nmax;
  hipLaunchKernelGGL(bn_bwd_reduce_kernel, grid, dim3(256), 0, s, *a);
  if (bn_all_narrow(a)) {
    hipLaunchKernelGGL(bn_bwd_apply_fin_kernel, grid, dim3(256), 0, s, *a);
  } else {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((nmax + 63) / 64, a->nprob), dim3(256), 0, s, *a);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, grid, dim3(256), 0, s, *a);
  }
  return launch_status("spv_bn_bwd");
}

static int check_sample(const spv_sample_batch* a, const char* who) {
  if (!a || a->nprob <= 0 || a->nprob > SPV_MAXP || a->B <= 0) return fail(SPV_ERR_ARG, "%s: bad batch", who);
  for (int i = 0; i < a->nprob; ++i)
    if (a->p[i].n <= 0 || a->p[i].n > 32 || !a->p[i].post || !a->p[i].eps || !a->p[i].scale) return fail(SPV_ERR_ARG, "%s: bad problem", who);
  return SPV_OK;
}
// encoder heads in two launches per direction (csrc/spv_small.h: enc_heads_*): bn->p[2e], bn->p[2e + 1] = the BatchNorm problems of
// encoder e's mu / logvar heads, sb->p[e] its sampling problem
static int check_heads(const spv_bn_batch* bn, const spv_sample_batch* sb, const char* who) {
  int rc = check_bn(bn, who);
  if (rc) return rc;
  rc = check_sample(sb, who);
  if (rc) return rc;
  if (bn->nprob != 2 * sb->nprob || bn->B != sb->B || bn->relu) return fail(SPV_ERR_ARG, "%s: two BatchNorm problems per sampling problem, same batch, no relu", who);
  for (int e = 0; e < sb->nprob; ++e) {
    const spv_bn_prob &q0 = bn->p[2 * e], &q1 = bn->p[2 * e + 1];
    const int n = sb->p[e].n;
    if (q0.N != n || q1.N != n || q0.img_hi || q1.img_hi) return fail(SPV_ERR_ARG, "%s: head widths differ from the sampling problem's n", who);
    if (q0.Y != sb->p[e].post || q1.Y != sb->p[e].post + n || q0.ldy != 2 * n || q1.ldy != 2 * n)
      return fail(SPV_ERR_ARG, "%s: the BatchNorm outputs must be the two halves of the sampling problem's post [B][2n]", who);
  }
  return SPV_OK;
}
extern "C" int spv_enc_heads_fwd(const spv_bn_batch* bn, const spv_sample_batch* sb, void* stream) {
  int rc = check_heads(bn, sb, "spv_enc_heads_fwd");
  if (rc) return rc;
  for (int i = 0; i < sb->nprob; ++i) if (!sb->p[i].logz || !sb->p[i].theta || !sb->p[i].kl) return fail(SPV_ERR_ARG, "spv_enc_heads_fwd: null output%s");
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = (bn->B + BN_ROWS - 1) / BN_ROWS;
  if (bn->training) hipLaunchKernelGGL(bn_stats_kernel, dim3(nb, bn->nprob), dim3(256), 0, s, *bn);
  hipLaunchKernelGGL(enc_heads_bn_sample_fwd_kernel, dim3(nb, sb->nprob), dim3(256), 0, s, *bn, *sb);
  return launch_status("spv_enc_heads_fwd");
}
extern "C" int spv_enc_heads_bwd(const spv_bn_batch* bn, const spv_sample_batch* sb, void* stream) {
  int rc = check_heads(bn, sb, "spv_enc_heads_bwd");
  if (rc) return rc;
  for (int i = 0; i < sb->nprob; ++i) if (!sb->p[i].d_post) return fail(SPV_ERR_ARG, "spv_enc_heads_bwd: null output%s");
  for (int i = 0; i < bn->nprob; ++i) {
    const spv_bn_prob& q = bn->p[i];
    if (!q.dX || !q.dgamma || !q.dbeta) return fail(SPV_ERR_ARG, "spv_enc_heads_bwd: null pointer%s");
    const int e = i / 2, n = sb->p[e].n;
    if (q.dY != sb->p[e].d_post + (i & 1) * n || q.lddy != 2 * n) return fail(SPV_ERR_ARG, "spv_enc_heads_bwd: dY must be the halves of d_post [B][2n]%s");
  }
  if (!bn->training) return fail(SPV_ERR_UNSUPPORTED, "spv_enc_heads_bwd: training-mode BatchNorm only (use spv_enc_sample_bwd + spv_bn_bwd)%s");
  hipStream_t s = (hipStream_t)stream;
  const unsigned nb = (bn->B + BN_ROWS - 1) / BN_ROWS;
  hipLaunchKernelGGL(enc_heads_bwd_reduce_kernel, dim3(nb, sb->nprob), dim3(256), 0, s, *bn, *sb);
  hipLaunchKernelGGL(bn_bwd_apply_fin_kernel, dim3(nb, bn->nprob), dim3(256), 0, s, *bn);
  return launch_status("spv_enc_heads_bwd");
}

extern "C" int spv_enc_sample_fwd(const spv_sample_batch* a, void* stream) {
  int rc = check_sample(a, "spv_enc_sample_fwd");
  if (rc) return rc;
  for (int i = 0; i < a->nprob; ++i) if (!a->p[i].logz || !a->p[i].theta || !a->p[i].kl) return fail(SPV_ERR_ARG, "spv_enc_sample_fwd: null output%s");
  hipLaunchKernelGGL(enc_sample_fwd_kernel, dim3((a->B + 7) / 8, a->nprob), dim3(256), 0, (hipStream_t)stream, *a);
  return launch_status("spv_enc_sample_fwd");
}
extern "C" int spv_enc_sample_bwd(const spv_sample_batch* a, void* stream) {
  int rc = check_sample(a, "spv_enc_sample_bwd");
  if (rc) return rc;
  for (int i = 0; i < a->nprob; ++i) if (!a->p[i].d_post) return fail(SPV_ERR_ARG, "spv_enc_sample_bwd: null output%s");
  hipLaunchKernelGGL(enc_sample_bwd_kernel, dim3((a->B + 7) / 8, a->nprob), dim3(256), 0, (hipStream_t)stream, *a);
  return launch_status("spv_enc_sample_bwd");
}

// ---------------------------------------------------------------------------------------------
// PoE (label) and decoder preparation
// ---------------------------------------------------------------------------------------------
extern "C" int spv_poe_partner(const float* labels0, const float* labels1, int32_t B0, int32_t B1, int32_t* order0, int32_t* order1,
                               int32_t* rank0, int32_t* rank1, int32_t* tables, int32_t* partner0, int32_t* mode0, int32_t* partner1,
                               int32_t* mode1, int32_t* err, void* stream) {
  if (!labels0 || !labels1 || !order0 || !order1 || !rank0 || !rank1 || !tables || !partner0 || !partner1 || !mode0 || !mode1 || !err || B0 <= 0 || B1 <= 0)
    return fail(SPV_ERR_ARG, "spv_poe_partner: bad arguments%s");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(poe_rank_kernel, dim3(2), dim3(512), 0, s, labels0, labels1, B0, B1, order0, order1, rank0, rank1, tables, err);
  const int Bm = B0 > B1 ? B0 : B1;
  hipLaunchKernelGGL(poe_lookup_kernel, dim3((Bm + 255) / 256, 2), dim3(256), 0, s, labels0, labels1, B0, B1, order0, order1, rank0, rank1,
                     tables, partner0, mode0, partner1, mode1);
  return launch_status("spv_poe_partner");
}
extern "C" int spv_poe_rank(const float* labels0, const float* labels1, int32_t B0, int32_t B1, int32_t* order0, int32_t* order1,
                            int32_t* rank0, int32_t* rank1, int32_t* tables, int32_t* err, void* stream) {
  if (!labels0 || !labels1 || !order0 || !order1 || !rank0 || !rank1 || !tables || !err || B0 <= 0 || B1 <= 0) return fail(SPV_ERR_ARG, "spv_poe_rank: bad arguments%s");
  hipLaunchKernelGGL(poe_rank_kernel, dim3(2), dim3(512), 0, (hipStream_t)stream, labels0, labels1, B0, B1, order0, order1, rank0, rank1, tables, err);
  return launch_status("spv_poe_rank");
}
static int check_poe(const spv_poe_args* a, const char* who) {
  if (!a || a->n <= 0 || a->n > 32 || a->B[0] <= 0 || a->B[1] <= 0) return fail(SPV_ERR_ARG, "%s: bad shape (latent dimension <= 32)", who);
  for (int g = 0; g < 2; ++g)
    if (!a->stats[g] || !a->partner[g] || !a->mode[g] || !a->eps[g] || !a->loc[g] || !a->scale[g] || a->ld[g] < 2 * a->n ||
        (a->expert[g] && a->ld_expert[g] < 2 * a->n))
      return fail(SPV_ERR_ARG, "%s: null pointer / bad pitch", who);
  return SPV_OK;
}
extern "C" int spv_poe_fuse_fwd(const spv_poe_args* a, void* stream) {
  int rc = check_poe(a, "spv_poe_fuse_fwd");
  if (rc) return rc;
  for (int g = 0; g < 2; ++g) if (!a->logvar[g] || !a->logz[g] || !a->theta[g] || !a->kl[g]) return fail(SPV_ERR_ARG, "spv_poe_fuse_fwd: null output%s");
  if (a->lab[0] || a->lab[1]) {
    if (!a->lab[0] || !a->lab[1] || !a->order[0] || !a->order[1] || !a->rank[0] || !a->rank[1] || !a->tables || a->expert[0] || a->expert[1])
      return fail(SPV_ERR_ARG, "spv_poe_fuse_fwd: the label lookup needs lab, order, rank of both groups and tables (and no plan experts)%s");
  }
  const int Bm = a->B[0] > a->B[1] ? a->B[0] : a->B[1];
  hipLaunchKernelGGL(poe_fuse_fwd_kernel, dim3((Bm + 7) / 8, 2), dim3(256), 0, (hipStream_t)stream, *a);
  return launch_status("spv_poe_fuse_fwd");
}
// zero fill as an ordinary kernel launch.  (hipMemsetAsync inside a captured graph becomes a memset node; with two processes sharing the
// GPU the two-rank hipGraph test intermittently -- ~1 run in 10 -- showed NaN gradients in exactly the tensors fed by the buffers these
// memsets clear, as if the fill had not happened before the kernel that accumulates into them; a kernel node has not shown it.)
__global__ void zero_fill_kernel(float* p, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = 0.f;
}
__global__ void fill_i32_kernel(int32_t* p, long n, int32_t v) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[i] = v;
}
static int zero_fill(void* p, size_t bytes, hipStream_t s) {
  const long n = (long)(bytes / 4);
  if (n <= 0) return SPV_OK;
  // DIAGNOSTIC switch (tools/graph_dump.py): the round-2 form, a memset NODE in a captured step, to dump that graph's edges
  static const bool memset_nodes = getenv("SPV_MEMSET_NODES") && atoi(getenv("SPV_MEMSET_NODES")) == 1;
  if (memset_nodes) return hipMemsetAsync(p, 0, bytes, s) == hipSuccess ? SPV_OK : SPV_ERR_LAUNCH;
  hipLaunchKernelGGL(zero_fill_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (float*)p, n);
  return hipGetLastError() == hipSuccess ? SPV_OK : SPV_ERR_LAUNCH;
}

extern "C" int spv_poe_fuse_bwd(const spv_poe_args* a, void* stream) {
  int rc = check_poe(a, "spv_poe_fuse_bwd");
  if (rc) return rc;
  for (int g = 0; g < 2; ++g) if (!a->d_stats[g]) return fail(SPV_ERR_ARG, "spv_poe_fuse_bwd: null output%s");
  const int Bm = a->B[0] > a->B[1] ? a->B[0] : a->B[1];
  const size_t bytes0 = (size_t)a->B[0] * a->ld[0] * sizeof(float);
  const bool adjacent = reinterpret_cast<char*>(a->d_stats[0]) + bytes0 == reinterpret_cast<char*>(a->d_stats[1]);  // one buffer: one memset
  for (int g = 0; g < 2; ++g) {  // the kernel accumulates (own expert + partner's): start from zero
    const size_t bytes = (size_t)a->B[g] * a->ld[g] * sizeof(float);
    if (g == 1 && adjacent) {
    } else if (zero_fill(a->d_stats[g], (g == 0 && adjacent) ? bytes0 + (size_t)a->B[1] * a->ld[1] * sizeof(float) : bytes, (hipStream_t)stream) != SPV_OK)
      return fail(SPV_ERR_LAUNCH, "spv_poe_fuse_bwd: fill failed%s");
    if (a->expert[g]) {
      if (!a->d_expert[g]) return fail(SPV_ERR_ARG, "spv_poe_fuse_bwd: d_expert missing%s");
      if (zero_fill(a->d_expert[g], (size_t)a->B[g] * a->ld_expert[g] * sizeof(float), (hipStream_t)stream) != SPV_OK)
        return fail(SPV_ERR_LAUNCH, "spv_poe_fuse_bwd: fill failed%s");
    }
  }
  if (a->lab[0] != nullptr)   // label PoE: one-to-one pairs
    hipLaunchKernelGGL(poe_fuse_bwd_kernel<false>, dim3((Bm + 7) / 8, 2), dim3(256), 0, (hipStream_t)stream, *a);
  else                        // arg-max partners may coincide: per-workgroup sums per partner, one atomic per column and partner
    hipLaunchKernelGGL(poe_fuse_bwd_kernel<true>, dim3((Bm + 31) / 32, 2), dim3(1024), 0, (hipStream_t)stream, *a);
  return launch_status("spv_poe_fuse_bwd");
}

// ---------------------------------------------------------------------------------------------
// N-group cluster-matched PoE (spv_poe_n.h)
// ---------------------------------------------------------------------------------------------
static int check_poe_comp(const spv_poe_comp_args* a, const char* who, bool bwd) {
  if (!a || a->ngroups < 2 || a->ngroups > SPV_POE_MAXG || a->n <= 0 || a->n > 32 || a->ncomp <= 0 || a->ncomp > SPV_POE_COMP_CMAX || !a->part || !a->mean)
    return fail(SPV_ERR_ARG, "%s: bad shape (2..4 groups, latent dimension <= 32, <= 64 components)", who);
  for (int g = 0; g < a->ngroups; ++g) {
    if (a->B[g] <= 0 || !a->stats[g] || a->ld[g] < 2 * a->n || !a->comp[g] || !a->eps[g] || !a->loc[g] || !a->scale[g]) return fail(SPV_ERR_ARG, "%s: null pointer / bad pitch", who);
    if (!bwd && (!a->logvar[g] || !a->logz[g] || !a->theta[g] || !a->kl[g])) return fail(SPV_ERR_ARG, "%s: null output", who);
    if (bwd && (!a->dpn[g] || !a->d_stats[g])) return fail(SPV_ERR_ARG, "%s: null backward buffer", who);
  }
  return SPV_OK;
}
extern "C" int spv_poe_comp_fwd(const spv_poe_comp_args* a, void* stream) {
  int rc = check_poe_comp(a, "spv_poe_comp_fwd", false);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  int Bm = 0;
  for (int g = 0; g < a->ngroups; ++g) Bm = a->B[g] > Bm ? a->B[g] : Bm;
  const int W = 2 * a->n + 1;
  hipLaunchKernelGGL(pn_stats_kernel, dim3(PN_SEG, a->ngroups), dim3(128), 0, s, *a);
  hipLaunchKernelGGL(pn_means_kernel, dim3((a->ncomp * W + 255) / 256, a->ngroups), dim3(256), 0, s, *a);
  hipLaunchKernelGGL(pn_fuse_fwd_kernel, dim3((Bm + 7) / 8, a->ngroups), dim3(256), 0, s, *a);
  return launch_status("spv_poe_comp_fwd");
}
extern "C" int spv_poe_comp_bwd(const spv_poe_comp_args* a, void* stream) {
  int rc = check_poe_comp(a, "spv_poe_comp_bwd", true);
  if (rc) return rc;
  hipStream_t s = (hipStream_t)stream;
  int Bm = 0;
  for (int g = 0; g < a->ngroups; ++g) Bm = a->B[g] > Bm ? a->B[g] : Bm;
  hipLaunchKernelGGL(pn_cell_bwd_kernel, dim3((Bm + 7) / 8, a->ngroups), dim3(256), 0, s, *a);
  hipLaunchKernelGGL(pn_comp_bwd_kernel, dim3(PN_SEG, a->ngroups), dim3(64), 0, s, *a);
  hipLaunchKernelGGL(pn_apply_bwd_kernel, dim3((Bm + 7) / 8, a->ngroups), dim3(256), 0, s, *a);
  return launch_status("spv_poe_comp_bwd");
}

extern "C" int spv_zsplit_fwd(const spv_zsplit_args* a, void* stream) {
  if (!a || a->B <= 0 || a->n_p <= 0 || a->n_s <= 0 || a->ngroups <= 0 || a->ngroups > 2) return fail(SPV_ERR_ARG, "spv_zsplit_fwd: bad shape%s");
  for (int g = 0; g < a->ngroups; ++g) if (!a->priv[g] || !a->poe[g] || !a->zcat[g]) return fail(SPV_ERR_ARG, "spv_zsplit_fwd: null pointer%s");
  const long tot = (long)a->B * (a->n_p + a->n_s);
  bool pack = false, pack_all = true;
  for (int g = 0; g < a->ngroups; ++g) { pack = pack || a->am_hi[g] || a->aps_hi[g]; pack_all = pack_all && a->am_hi[g] && a->aps_hi[g]; }
  // with both operand images requested for every group ONE kernel writes zcat and the images (zsplit_pack_kernel computes the latents'
  // positions itself); otherwise zcat first, then whatever images were asked for
  if (!pack_all) hipLaunchKernelGGL(zsplit_fwd_kernel, dim3((unsigned)((tot + 255) / 256), a->ngroups), dim3(256), 0, (hipStream_t)stream, *a);
  if (pack) {
    if (a->Bp < a->B || a->n_p + 1 > SPV_DEC_KP || a->n_s + 1 > SPV_DEC_KS || a->am_cols < a->n_p + a->n_s + 1 || a->am_cols < 0 ||
        a->ld_am < a->am_col + a->am_cols)
      return fail(SPV_ERR_ARG, "spv_zsplit_fwd: bad operand-image shape%s");
    const long tp = (long)a->Bp * (SPV_DEC_KP + SPV_DEC_KS + a->am_cols);
    hipLaunchKernelGGL(zsplit_pack_kernel, dim3((unsigned)((tp + 255) / 256), a->ngroups), dim3(256), 0, (hipStream_t)stream, *a);
  }
  return launch_status("spv_zsplit_fwd");
}
extern "C" int spv_zsplit_bwd(const spv_zsplit_args* a, void* stream) {
  if (!a || a->B <= 0 || a->n_p <= 0 || a->n_s <= 0 || a->ngroups <= 0 || a->ngroups > 2) return fail(SPV_ERR_ARG, "spv_zsplit_bwd: bad shape%s");
  for (int g = 0; g < a->ngroups; ++g) if (!a->d_zcat[g] || !a->d_priv[g] || !a->d_poe[g]) return fail(SPV_ERR_ARG, "spv_zsplit_bwd: null pointer%s");
  const long tot = (long)a->B * (a->n_p + a->n_s);
  hipLaunchKernelGGL(zsplit_bwd_kernel, dim3((unsigned)((tot + 255) / 256), a->ngroups), dim3(256), 0, (hipStream_t)stream, *a);
  return launch_status("spv_zsplit_bwd");
}

static int check_fold(const spv_fold_batch* a, const char* who) {
  if (!a || a->nprob <= 0 || a->nprob > SPV_MAXP || a->B <= 0) return fail(SPV_ERR_ARG, "%s: bad batch", who);
  for (int i = 0; i < a->nprob; ++i) {
    const spv_fold_prob& q = a->p[i];
    if (q.K <= 0 || q.K > FOLD_KMAX || q.G <= 0 || q.Gp < q.G || !q.W || !q.gamma || !q.beta || !q.running_mean || !q.running_var || !q.stat ||
        (a->training && (!q.zsum || !q.zz)))
      return fail(SPV_ERR_ARG, "%s: bad problem (K <= 32, non-null pointers)", who);
  }
  return SPV_OK;
}
extern "C" int spv_bn_fold_fwd(const spv_fold_batch* a, void* stream) {
  int rc = check_fold(a, "spv_bn_fold_fwd");
  if (rc) return rc;
  int gmax = 0;
  for (int i = 0; i < a->nprob; ++i) {
    const spv_fold_prob& q = a->p[i];
    if (!q.img_hi || !q.img_lo || q.slot < q.K + 1 || q.col_off + q.slot > q.ld_img) return fail(SPV_ERR_ARG, "spv_bn_fold_fwd: bad image slot%s");
    if ((q.slot | q.col_off | q.ld_img) % 8 || ((reinterpret_cast<uintptr_t>(q.img_hi) | reinterpret_cast<uintptr_t>(q.img_lo)) & 15))
      return fail(SPV_ERR_ARG, "spv_bn_fold_fwd: slot, col_off, ld_img must be multiples of 8 elements and the images 16-byte aligned%s");
    gmax = q.Gp > gmax ? q.Gp : gmax;
  }
  hipLaunchKernelGGL(bn_fold_fwd_kernel, dim3((gmax + FOLD_FWD_GENES - 1) / FOLD_FWD_GENES, a->nprob), dim3(256), 0, (hipStream_t)stream, *a);
  return launch_status("spv_bn_fold_fwd");
}
extern "C" int spv_bn_fold_bwd(const spv_fold_batch* a, void* stream) {
  int rc = check_fold(a, "spv_bn_fold_bwd");
  if (rc) return rc;
  int gmax = 0;
  for (int i = 0; i < a->nprob; ++i) {
    const spv_fold_prob& q = a->p[i];
    if (!q.dWeff || !q.dW || !q.dgamma || !q.dbeta || q.ld_dw < q.K + 1 || (a->training && (!q.red_part || !q.z || !q.dz)))
      return fail(SPV_ERR_ARG, "spv_bn_fold_bwd: null pointer%s");
    gmax = q.G > gmax ? q.G : gmax;
  }
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bn_fold_bwd_kernel, dim3((gmax + 255) / 256, a->nprob), dim3(FOLD_BWD_THREADS), 0, s, *a);
  if (a->training) {   // (the sum over the gene blocks' partials happens inside zstats_bwd_kernel)
    for (int i = 0; i < a->nprob; ++i) {
      const spv_fold_prob& q = a->p[i];
      if (q.out_priv && (!q.out_poe || q.n_p <= 0 || q.n_s <= 0 || q.zcol < 0 || q.zcol + q.K > q.n_p + q.n_s))
        return fail(SPV_ERR_ARG, "spv_bn_fold_bwd: bad latent-slicing outputs%s");
    }
    hipLaunchKernelGGL(zstats_bwd_kernel, dim3((a->B + 63) / 64, a->nprob), dim3(256), 0, s, *a);
  }
  return launch_status("spv_bn_fold_bwd");
}

static int check_trunk(const spv_trunk_batch* a, const char* who) {
  if (!a || a->nprob <= 0 || a->nprob > 2 || a->B <= 0) return fail(SPV_ERR_ARG, "%s: bad batch", who);
  for (int i = 0; i < a->nprob; ++i) {
    const spv_trunk_prob& q = a->p[i];
    if (q.N <= 0 || q.N > 256 || q.K <= 0 || q.K > SPV_TRUNK_KMAX) return fail(SPV_ERR_UNSUPPORTED, "%s: N <= 256 and K <= 48", who);
    if (!q.W || !q.gamma || !q.beta || !q.running_mean || !q.running_var || !q.stat) return fail(SPV_ERR_ARG, "%s: null pointer", who);
    if (a->training && (!q.zsum || !q.zz)) return fail(SPV_ERR_ARG, "%s: training mode needs the z statistics", who);
  }
  return SPV_OK;
}
extern "C" int spv_trunk_fold_fwd(const spv_trunk_batch* a, void* stream) {
  int rc = check_trunk(a, "spv_trunk_fold_fwd");
  if (rc) return rc;
  int nmax = 0;
  for (int i = 0; i < a->nprob; ++i) {
    if (!a->p[i].Wf || !a->p[i].cf) return fail(SPV_ERR_ARG, "spv_trunk_fold_fwd: null output%s");
    nmax = a->p[i].N > nmax ? a->p[i].N : nmax;
  }
  hipLaunchKernelGGL(trunk_fold_fwd_kernel, dim3((nmax + TRUNK_FWD_ROWS - 1) / TRUNK_FWD_ROWS, a->nprob), dim3(1024), 0, (hipStream_t)stream, *a);
  return launch_status("spv_trunk_fold_fwd");
}
extern "C" int spv_trunk_fold_bwd(const spv_trunk_batch* a, void* stream) {
  int rc = check_trunk(a, "spv_trunk_fold_bwd");
  if (rc) return rc;
  for (int i = 0; i < a->nprob; ++i) {
    const spv_trunk_prob& q = a->p[i];
    if (!q.dWf || !q.dcf || !q.dW || !q.dgamma || !q.dbeta) return fail(SPV_ERR_ARG, "spv_trunk_fold_bwd: null pointer%s");
    if (a->training && (!q.dred || !q.z || !q.dz || q.ldz < q.K || q.lddz < q.K)) return fail(SPV_ERR_ARG, "spv_trunk_fold_bwd: z-statistics buffers missing%s");
  }
  hipStream_t s = (hipStream_t)stream;
  int nmax = 0;
  for (int i = 0; i < a->nprob; ++i) nmax = a->p[i].N > nmax ? a->p[i].N : nmax;
  const int nblk = (nmax + TRUNK_BWD_ROWS - 1) / TRUNK_BWD_ROWS;
  hipLaunchKernelGGL(trunk_fold_bwd_kernel, dim3(nblk, a->nprob), dim3(1024), 0, s, *a);
  if (a->training) hipLaunchKernelGGL(trunk_zstats_kernel, dim3((a->B + 63) / 64, a->nprob), dim3(256), 0, s, *a, nblk);
  return launch_status("spv_trunk_fold_bwd");
}

extern "C" int spv_reduce_slabs(const spv_reduce_batch* b, void* stream) {
  if (!b || b->nprob <= 0 || b->nprob > SPV_MAXR) return fail(SPV_ERR_ARG, "spv_reduce_slabs: bad batch%s");
  long blocks = 1;
  for (int i = 0; i < b->nprob; ++i) {
    const spv_reduce_prob& q = b->p[i];
    if (!q.src || !q.dst || q.nslabs <= 0 || q.rows <= 0 || q.cols <= 0 || q.ld_src < q.col_off + q.cols || q.ld_dst < q.cols || q.col_off < 0 ||
        (long)q.rows * q.cols >= (1l << 31))
      return fail(SPV_ERR_ARG, "spv_reduce_slabs: bad problem%s");
    const long t = reduce_blocks_needed(q);   // the grid covers the neediest problem in one pass (beyond 8192 workgroups: grid-stride loops)
    if (t > blocks) blocks = t;
  }
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(reduce_slabs_kernel, dim3((unsigned)blocks, b->nprob), dim3(256), 0, (hipStream_t)stream, *b);
  return launch_status("spv_reduce_slabs");
}

extern "C" int spv_loss_assemble(const float* rec0, const float* rec1, const float* w, const float* const* kl, int32_t nkl, int32_t B,
                                 const float* kl_weight, float* loss, float* rec_sum, float* gkl, void* stream) {
  if (!rec0 || !w || !loss || B <= 0 || nkl < 0 || nkl > 4 || (nkl > 0 && !kl)) return fail(SPV_ERR_ARG, "spv_loss_assemble: bad arguments%s");
  const float* k[4] = {nullptr, nullptr, nullptr, nullptr};
  for (int i = 0; i < nkl; ++i) k[i] = kl[i];   // kl is a HOST array of device pointers
  hipLaunchKernelGGL(loss_assemble_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, rec0, rec1, w, k[0], k[1], k[2], k[3], B, kl_weight,
                     loss, rec_sum, gkl);
  return launch_status("spv_loss_assemble");
}

// relu mask + 16-bit packing of the fc1 output gradient and its column sums (the bias gradients), two launches:
//   fc1_bwd_prep_*: one workgroup per 16 rows, one thread per column: column partial sums of dpre = dh1 * (h1 > 0); split mode
//     (img_lo != NULL, "fp32" precision) also writes the bf16 hi / lo images here; f16 mode records the block's max |dpre| instead;
//   fc1_bwd_finish_*: f16 mode: every workgroup reduces the block maxima (fixed order, <= 1024 values) to the step's power-of-two
//     scale (spv_common.h: pow2_scale_for; the largest element lands in [4096, 8192)), writes f16(dpre * scale) for its 16 rows and
//     workgroup 0 leaves {scale, 1 / scale} in scale_ws[0..1] for the weight-gradient kernel; the first ceil(N1 / 64) workgroups
//     (both modes) sum the column partials in block order into the bias gradients.
// No atomics, no grid synchronisation: the scale is a pure function of dpre, so eager and replayed steps agree bit for bit.
struct Fc1PrepArgs { const float* dh1; const float* h1; int B, N1; bf16_t* img_hi; bf16_t* img_lo; long ld_img; float* part; int nblk; float* db; float* db2; int n_first;
                     float* scale_ws; };   // scale_ws: fp32 [2 + nblk] (f16 mode), NULL in split mode

__device__ __forceinline__ float block_max_256(float v, float* s_red) {   // max over the 256 threads of a workgroup, returned to all of them
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
}

__device__ __forceinline__ void fc1_bwd_prep_body(const Fc1PrepArgs& a, const int blk, float* s_red) {
  const int r0 = blk * 16;
  const bool half = a.img_lo == nullptr;
  float amax = 0.f;
  for (int col = threadIdx.x; col < (int)a.ld_img; col += 256) {
    float sum = 0.f;
#pragma unroll 4
    for (int r = r0; r < r0 + 16; ++r) {
      // (unconditional loads from a clamped address, then a select: a load behind the bounds test is waited for inside that branch)
      const long i = (long)min(r, a.B - 1) * a.N1 + min(col, a.N1 - 1);
      const float hv = a.h1[i], gv = a.dh1[i];
      const float v = (r < a.B && col < a.N1 && hv > 0.f) ? gv : 0.f;
      if (!half) {
        bf16_t hi, lo;
        split_bf16(v, hi, lo);
        a.img_hi[(long)r * a.ld_img + col] = hi;
        a.img_lo[(long)r * a.ld_img + col] = lo;
      }
      amax = fmaxf(amax, fabsf(v));
      sum += v;
    }
    if (col < a.N1) a.part[(long)blk * a.N1 + col] = sum;
  }
  if (half) {
    amax = block_max_256(amax, s_red);
    if (threadIdx.x == 0) a.scale_ws[2 + blk] = amax;
  }
}

__device__ __forceinline__ void fc1_bwd_finish_body(const Fc1PrepArgs& a, const int blk, float* s_red, float (*s_p)[64]) {
  const bool half = a.img_lo == nullptr;
  if (half && blk < a.nblk) {
    float m = 0.f;
    for (int k = threadIdx.x; k < a.nblk; k += 256) m = fmaxf(m, a.scale_ws[2 + k]);
    m = block_max_256(m, s_red);
    float scale, inv;
    pow2_scale_for(m, scale, inv);
    if (blk == 0 && threadIdx.x == 0) { a.scale_ws[0] = scale; a.scale_ws[1] = inv; }
    const int r0 = blk * 16;
    for (int col = threadIdx.x; col < (int)a.ld_img; col += 256) {
#pragma unroll 4
      for (int r = r0; r < r0 + 16; ++r) {
        const long i = (long)min(r, a.B - 1) * a.N1 + min(col, a.N1 - 1);
        const float hv = a.h1[i], gv = a.dh1[i];
        const float v = (r < a.B && col < a.N1 && hv > 0.f) ? gv : 0.f;
        a.img_hi[(long)r * a.ld_img + col] = f2h(v * scale);
      }
    }
  }
  if (blk >= (a.N1 + 63) / 64) return;   // (workgroup-uniform)
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int col = blk * 64 + tx;
  float v = 0.f;
  if (col < a.N1)
#pragma unroll 8
    for (int k = ty; k < a.nblk; k += 4) v += a.part[(long)k * a.N1 + col];
  s_p[ty][tx] = v;
  __syncthreads();
  if (ty != 0 || col >= a.N1) return;
  v = ((s_p[0][tx] + s_p[1][tx]) + s_p[2][tx]) + s_p[3][tx];
  if (a.db2 != nullptr && col >= a.n_first) a.db2[col - a.n_first] = v;
  else a.db[col] = v;
}

__global__ __launch_bounds__(256) void fc1_bwd_prep_kernel(Fc1PrepArgs a) {
  __shared__ float s_red[4];
  fc1_bwd_prep_body(a, blockIdx.x, s_red);
}
__global__ __launch_bounds__(256) void fc1_bwd_finish_kernel(Fc1PrepArgs a) {
  __shared__ float s_red[4];
  __shared__ float s_p[4][64];
  fc1_bwd_finish_body(a, blockIdx.x, s_red, s_p);
}
__global__ __launch_bounds__(256) void fc1_bwd_prep_pair_kernel(Fc1PrepArgs a0, Fc1PrepArgs a1) {   // blockIdx.y = group
  __shared__ float s_red[4];
  const Fc1PrepArgs a = blockIdx.y ? a1 : a0;
  if ((int)blockIdx.x >= a.nblk) return;
  fc1_bwd_prep_body(a, blockIdx.x, s_red);
}
__global__ __launch_bounds__(256) void fc1_bwd_finish_pair_kernel(Fc1PrepArgs a0, Fc1PrepArgs a1) {   // blockIdx.y = group
  __shared__ float s_red[4];
  __shared__ float s_p[4][64];
  const Fc1PrepArgs a = blockIdx.y ? a1 : a0;
  fc1_bwd_finish_body(a, blockIdx.x, s_red, s_p);
}
static int fc1_finish_blocks(const Fc1PrepArgs& a) {
  const int nb_bias = (a.N1 + 63) / 64;
  return (a.img_lo == nullptr && a.nblk > nb_bias) ? a.nblk : nb_bias;
}

extern "C" int spv_enc_fc1_bwd_prep(const float* dh1, const float* h1, int32_t B, int32_t N1, uint16_t* img_hi, uint16_t* img_lo,
                                    int64_t ld_img, int32_t Bp, float* part, float* db, float* db2, int32_t n_first, float* scale_ws, void* stream) {
  if (!dh1 || !h1 || !img_hi || !part || !db) return fail(SPV_ERR_ARG, "spv_enc_fc1_bwd_prep: null pointer%s");
  if (B <= 0 || N1 <= 0 || Bp < B || (Bp % 64) || ld_img < N1) return fail(SPV_ERR_ARG, "spv_enc_fc1_bwd_prep: bad shape (Bp a multiple of 64)%s");
  if (!img_lo && !scale_ws) return fail(SPV_ERR_ARG, "spv_enc_fc1_bwd_prep: the f16 image (img_lo == NULL) needs scale_ws [2 + Bp / 16]%s");
  if (Bp / 16 > 65536) return fail(SPV_ERR_ARG, "spv_enc_fc1_bwd_prep: minibatch too large%s");
  hipStream_t s = (hipStream_t)stream;
  const Fc1PrepArgs a{dh1, h1, B, N1, (bf16_t*)img_hi, (bf16_t*)img_lo, (long)ld_img, part, Bp / 16, db, db2, n_first, scale_ws};
  hipLaunchKernelGGL(fc1_bwd_prep_kernel, dim3(a.nblk), dim3(256), 0, s, a);
  hipLaunchKernelGGL(fc1_bwd_finish_kernel, dim3(fc1_finish_blocks(a)), dim3(256), 0, s, a);
  return launch_status("spv_enc_fc1_bwd_prep");
}

// ---- both groups of a step in one launch per kernel (csrc/spv_fc1.h: *_pair_kernel) ---------------------------------------------------
static bool fc1_fwd_dma_ok(const spv_fc1_fwd_args& a) {
  return a.x && a.x->X && a.W1_hi && a.bias && a.slabs && a.h1 && a.library && a.xb_all && a.library_all && a.B > 0 && a.G > 0 && a.splits > 0 &&
         spv_enc_fc1_fwd_uses_dma(a.B, a.G, a.N1, a.nsplit, 1, a.ldw, a.ld_xb) &&
         ((reinterpret_cast<uintptr_t>(a.xb_all) | reinterpret_cast<uintptr_t>(a.W1_hi)) & 15) == 0 &&
         (a.nsplit == 1 || (a.W1_lo && (reinterpret_cast<uintptr_t>(a.W1_lo) & 15) == 0));
}
static GemmParams fc1_fwd_dma_params(const spv_fc1_fwd_args& a) {
  GemmParams p{};
  p.A = a.xb_all; p.lda = a.ld_xb; p.B = a.W1_hi; p.ldb = a.ldw; p.rows = a.x->rows;
  if (a.nsplit == 3) p.B_lo = a.W1_lo;   // "fp32" mode: hi / lo planes interleaved inside the image rows, hi / lo weight images
  p.n_cells = a.B; p.n_genes = a.G; p.C = a.slabs; p.ldc = a.N1; p.M = a.B; p.N = a.N1; p.K = a.G;
  const int kt = (a.G + F1_BK - 1) / F1_BK;
  p.k_per_split = ((kt + a.splits - 1) / a.splits) * F1_BK;
  p.c_split_row = a.splits;
  return p;
}

extern "C" int spv_enc_fc1_fwd_grouped(const spv_fc1_fwd_args* g, int32_t n_groups, void* stream) {
  if (!g || n_groups <= 0) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd_grouped: bad arguments%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  for (; i + 1 < n_groups; i += 2) {
    const spv_fc1_fwd_args &a = g[i], &b = g[i + 1];
    if (!(fc1_fwd_dma_ok(a) && fc1_fwd_dma_ok(b) && a.N1 == b.N1 && a.nsplit == b.nsplit)) break;
    const GemmParams p0 = fc1_fwd_dma_params(a), p1 = fc1_fwd_dma_params(b);
    const int mt0 = (a.B + F1_BM - 1) / F1_BM, mt1 = (b.B + F1_BM - 1) / F1_BM;
    static bool raised = false;
    if (!raised) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fc1_fwd_dma_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS_BYTES);
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(fc1_fwd_dma_split_pair_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, F1_LDS_BYTES);
      raised = true;
    }
    const float acc_scale = (a.nsplit == 1) ? 1.0f / SPV_FC1_W_SCALE : 1.0f;
    if (a.nsplit == 3) hipLaunchKernelGGL(fc1_fwd_dma_split_pair_kernel, dim3(mt0 * a.splits + mt1 * b.splits, a.N1 / F1_BN), dim3(512), F1_LDS_BYTES, s, p0, p1, mt0 * a.splits);
    else hipLaunchKernelGGL(fc1_fwd_dma_pair_kernel, dim3(mt0 * a.splits + mt1 * b.splits, a.N1 / F1_BN), dim3(512), F1_LDS_BYTES, s, p0, p1, mt0 * a.splits);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_enc_fc1_fwd_grouped gemm");
    const long se0 = (long)mt0 * F1_BM * a.N1, se1 = (long)mt1 * F1_BM * b.N1;
    const Fc1EpiArgs e0{a.slabs, a.splits, se0, a.B, a.N1, a.bias, a.bias2, a.n_first, a.h1, a.library, a.library_all, a.x->rows, acc_scale, a.cov, a.cov_idx};
    const Fc1EpiArgs e1{b.slabs, b.splits, se1, b.B, b.N1, b.bias, b.bias2, b.n_first, b.h1, b.library, b.library_all, b.x->rows, acc_scale, b.cov, b.cov_idx};
    const long sem = se0 > se1 ? se0 : se1;
    hipLaunchKernelGGL(fc1_epilogue_tiled_pair_kernel, dim3((unsigned)((sem / 4 + 255) / 256), 2), dim3(256), 0, s, e0, e1);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_enc_fc1_fwd_grouped epilogue");
  }
  // shapes the LDS-DMA kernel does not take, 16-bit mode on resident images, at most 128 output units (n_hidden <= 64): the register-staged
  // GEMM and its epilogue for both groups in one grid each (gemm_pair_kernel) -- at 128 cells per minibatch a launch is mostly fixed cost
  typedef GemmCfg<64, 128, 2, 2, false, false, SRC_GATHER, SRC_PLAIN, unsigned short, 1, 32, 4, 1, true> PlainCfg;
  auto plain_ok = [](const spv_fc1_fwd_args& a) {
    return a.x && a.x->X && a.W1_hi && a.bias && a.slabs && a.rowsum_ws && a.h1 && a.library && a.xb_all && a.library_all && a.B > 0 && a.G > 0 && a.splits > 0 &&
           a.nsplit == 1 && a.N1 > 32 && a.N1 <= 128 && (a.ldw % 32) == 0 && a.ldw >= ((a.G + 31) & ~31) && a.ld_xb >= ((a.G + 31) & ~31) && (a.ld_xb % 8) == 0 &&
           (a.cov == nullptr) == (a.cov_idx == nullptr);
  };
  for (; i + 1 < n_groups; i += 2) {
    const spv_fc1_fwd_args &a = g[i], &b = g[i + 1];
    if (!(plain_ok(a) && plain_ok(b) && a.N1 == b.N1 && a.splits == b.splits)) break;
    GemmParams p[2];
    for (int k = 0; k < 2; ++k) {
      const spv_fc1_fwd_args& c = k ? b : a;
      GemmParams& w = p[k];
      w = GemmParams{};
      w.A = c.xb_all; w.lda = c.ld_xb; w.B = c.W1_hi; w.ldb = c.ldw;
      w.rows = c.x->rows; w.counts_aligned = counts_aligned(c.x); w.col_off = c.x->col_off; w.n_cells = c.B; w.n_genes = c.G;
      w.C = c.slabs; w.ldc = c.N1; w.slab_stride = (long)c.B * c.N1;
      w.M = c.B; w.N = c.N1; w.K = c.G;
      w.k_per_split = (((c.G + 31) / 32 + c.splits - 1) / c.splits) * 32;
      w.epi = EPI_STORE;
    }
    if (launch_gemm_pair<PlainCfg>(p[0], p[1], a.splits, s) != SPV_OK) return launch_status("spv_enc_fc1_fwd_grouped gemm pair");
    const float sc = 1.0f / SPV_FC1_W_SCALE;
    const Fc1EpiPlainArgs e0{a.slabs, a.rowsum_ws, a.splits, a.B, a.N1, a.bias, a.bias2, a.n_first, a.h1, a.library, a.library_all, a.x->rows, sc, a.cov, a.cov_idx};
    const Fc1EpiPlainArgs e1{b.slabs, b.rowsum_ws, b.splits, b.B, b.N1, b.bias, b.bias2, b.n_first, b.h1, b.library, b.library_all, b.x->rows, sc, b.cov, b.cov_idx};
    const long tmax = (long)(a.B > b.B ? a.B : b.B) * a.N1;
    hipLaunchKernelGGL(fc1_epilogue_pair_kernel, dim3((unsigned)((tmax + 255) / 256), 2), dim3(256), 0, s, e0, e1);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_enc_fc1_fwd_grouped epilogue pair");
  }
  for (; i < n_groups; ++i) {   // a leftover group, or shapes no pair form takes: the per-group entry point
    const spv_fc1_fwd_args& a = g[i];
    const int rc = spv_enc_fc1_fwd(a.x, a.B, a.G, a.W1_hi, a.W1_lo, a.ldw, a.N1, a.bias, a.bias2, a.n_first, a.nsplit, a.splits, a.slabs, a.rowsum_ws, a.h1,
                                   a.library, a.xb_all, a.ld_xb, a.library_all, a.cov, a.cov_idx, stream);
    if (rc != SPV_OK) return rc;
  }
  return SPV_OK;
}

static bool fc1_bwd_dma_ok(const spv_fc1_bwd_args& a) {
  if (!a.x || !a.x->X || !a.dh1 || !a.h1 || !a.dh_hi || !a.part || !a.db || !a.dW || !a.dW2 || !a.xb || !a.scale_ws || a.B <= 0 || a.G <= 0) return false;
  const int Kpad = (a.B + FW_BK - 1) / FW_BK * FW_BK;
  return a.nsplit == 1 && (a.N1 == FW_BM || a.N1 == 2 * FW_BM) && a.ld_dh == a.N1 && a.n_first == a.N1 / 2 && a.Bp >= Kpad && (a.Bp % 64) == 0 && a.ldc >= a.G &&
         a.ld_xb >= ((a.G + 63) & ~63) && (a.ld_xb % 8) == 0 && fw_lds_bytes(96, Kpad) <= 160 * 1024 &&
         ((reinterpret_cast<uintptr_t>(a.xb) | reinterpret_cast<uintptr_t>(a.dh_hi)) & 15) == 0;
}

extern "C" int spv_enc_fc1_bwd_grouped(const spv_fc1_bwd_args* g, int32_t n_groups, void* stream) {
  if (!g || n_groups <= 0) return fail(SPV_ERR_ARG, "spv_enc_fc1_bwd_grouped: bad arguments%s");
  hipStream_t s = (hipStream_t)stream;
  int i = 0;
  for (; i + 1 < n_groups; i += 2) {
    const spv_fc1_bwd_args &a = g[i], &b = g[i + 1];
    if (!(fc1_bwd_dma_ok(a) && fc1_bwd_dma_ok(b) && a.N1 == b.N1)) break;
    const Fc1PrepArgs q0{a.dh1, a.h1, a.B, a.N1, (bf16_t*)a.dh_hi, nullptr, (long)a.ld_dh, a.part, a.Bp / 16, a.db, a.db2, a.n_first, a.scale_ws};
    const Fc1PrepArgs q1{b.dh1, b.h1, b.B, b.N1, (bf16_t*)b.dh_hi, nullptr, (long)b.ld_dh, b.part, b.Bp / 16, b.db, b.db2, b.n_first, b.scale_ws};
    const int nb = q0.nblk > q1.nblk ? q0.nblk : q1.nblk;
    const int nf0 = fc1_finish_blocks(q0), nf1 = fc1_finish_blocks(q1);
    hipLaunchKernelGGL(fc1_bwd_prep_pair_kernel, dim3(nb, 2), dim3(256), 0, s, q0, q1);
    hipLaunchKernelGGL(fc1_bwd_finish_pair_kernel, dim3(nf0 > nf1 ? nf0 : nf1, 2), dim3(256), 0, s, q0, q1);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_enc_fc1_bwd_grouped prep");
    GemmParams p[2];
    for (int k = 0; k < 2; ++k) {
      const spv_fc1_bwd_args& c = k ? b : a;
      GemmParams& w = p[k];
      w = GemmParams{};
      w.A = c.dh_hi; w.lda = c.ld_dh; w.B = c.xb; w.ldb = c.ld_xb; w.rows = c.x->rows; w.n_cells = c.B; w.n_genes = c.G;
      w.C = c.dW; w.ldc = c.ldc; w.C2 = c.dW2; w.c_split_row = c.n_first; w.M = c.N1; w.N = c.G; w.K = c.B;
      w.out_scale = c.scale_ws + 1;
    }
    // tile: the one that gets BOTH groups into the fewest rounds of 256 one-per-CU workgroups (64-gene tiles are the faster ones alone)
    const int n64 = (a.G + 63) / 64 + (b.G + 63) / 64, n96 = (a.G + 95) / 96 + (b.G + 95) / 96;
    const int mth = a.N1 / FW_BM;   // 256-unit halves of the dh image (H = 256: one per encoder)
    const bool use96 = (n96 * mth + 255) / 256 < (n64 * mth + 255) / 256;
    const int Kp0 = (a.B + FW_BK - 1) / FW_BK * FW_BK, Kp1 = (b.B + FW_BK - 1) / FW_BK * FW_BK;
    const int Kp = Kp0 > Kp1 ? Kp0 : Kp1;
    // ... unless a WIDE tile (192 / 256 genes, fc1_wgrad_dma_wide_body) needs fewer ROUNDS of workgroups: a workgroup re-reads the dh image
    // (512 B per cell) beside its 2 t bytes per cell of image rows, so a wide tile is the slower one per workgroup and the faster one only
    // where the narrow tiles need a second round (rocprofv3, us per pair launch: C2 96: 77, 128: 83, 192: 112, 256: 114; G 20 000 96: 154,
    // 192: 123, 256: 125)
    const char* const force_env = getenv("SPV_FC1_WGRAD_TILE");   // (tuning / test switch, read per call: 64 / 96 / 128 / 192 / 256)
    const int force_tile = force_env ? atoi(force_env) : 0;
    int t = use96 ? 96 : 64;
    auto model_us = [&](int tile) {   // every workgroup pulls its bytes at the ONE CU's fill rate, ~32 GB/s measured on every LDS-DMA kernel here, round after round
      const long nwg = (long)((a.G + tile - 1) / tile + (b.G + tile - 1) / tile) * mth;
      return (double)((nwg + 255) / 256) * (double)Kp * (512 + 2 * tile) / 32e3;
    };
    if (Kp >= 512) {
      if (fww_lds_bytes(192, Kp) <= 160 * 1024 && model_us(192) < model_us(t)) t = 192;
      if (fww_lds_bytes(256, Kp) <= 160 * 1024 && model_us(256) < model_us(t)) t = 256;
    }
    if (force_tile == 64 || force_tile == 96 || force_tile == 128 || force_tile == 192 || force_tile == 256) {
      const int need = force_tile >= 192 ? fww_lds_bytes(force_tile, Kp) : fw_lds_bytes(force_tile, Kp);
      if (need <= 160 * 1024) t = force_tile;
    }
    const int lds = t >= 192 ? fww_lds_bytes(t, Kp) : fw_lds_bytes(t, Kp);
    void (*kfn)(GemmParams, GemmParams, int) = t == 256 ? fc1_wgrad_dma_wide_pair_kernel<256> : t == 192 ? fc1_wgrad_dma_wide_pair_kernel<192> :
                                               t == 96 ? fc1_wgrad_dma_pair_kernel<96> : (t == 128 ? fc1_wgrad_dma_pair_kernel<128> : fc1_wgrad_dma_pair_kernel<64>);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int nA = (a.G + t - 1) / t, nB = (b.G + t - 1) / t;
    hipLaunchKernelGGL(kfn, dim3(nA + nB, mth), dim3(512), lds, s, p[0], p[1], nA);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_enc_fc1_bwd_grouped wgrad");
  }
  // shapes the LDS-DMA kernel does not take (n_hidden <= 64), 16-bit mode on resident images: prep / finish and the register-staged weight-gradient
  // GEMM for both groups in one grid each
  auto plain_ok = [](const spv_fc1_bwd_args& a) {
    return a.x && a.x->X && a.dh1 && a.h1 && a.dh_hi && a.part && a.db && a.dW && a.xb && a.scale_ws && a.B > 0 && a.G > 0 && a.nsplit == 1 &&
           a.N1 > 32 && a.N1 <= 128 && a.Bp >= a.B && (a.Bp % 64) == 0 && a.Bp / 16 <= 65536 && a.ld_dh >= ((a.N1 + 127) & ~127) && (a.ld_dh % 8) == 0 &&
           a.ldc >= a.G && a.ld_xb >= ((a.G + 63) & ~63) && (a.ld_xb % 8) == 0;
  };
  for (; i + 1 < n_groups; i += 2) {
    const spv_fc1_bwd_args &a = g[i], &b = g[i + 1];
    const bool t96a = a.ld_xb >= (a.G + 95) / 96 * 96, t96b = b.ld_xb >= (b.G + 95) / 96 * 96;
    if (!(plain_ok(a) && plain_ok(b) && a.N1 == b.N1 && t96a == t96b)) break;
    const Fc1PrepArgs q0{a.dh1, a.h1, a.B, a.N1, (bf16_t*)a.dh_hi, nullptr, (long)a.ld_dh, a.part, a.Bp / 16, a.db, a.db2, a.n_first, a.scale_ws};
    const Fc1PrepArgs q1{b.dh1, b.h1, b.B, b.N1, (bf16_t*)b.dh_hi, nullptr, (long)b.ld_dh, b.part, b.Bp / 16, b.db, b.db2, b.n_first, b.scale_ws};
    const int nb = q0.nblk > q1.nblk ? q0.nblk : q1.nblk;
    const int nf0 = fc1_finish_blocks(q0), nf1 = fc1_finish_blocks(q1);
    hipLaunchKernelGGL(fc1_bwd_prep_pair_kernel, dim3(nb, 2), dim3(256), 0, s, q0, q1);
    hipLaunchKernelGGL(fc1_bwd_finish_pair_kernel, dim3(nf0 > nf1 ? nf0 : nf1, 2), dim3(256), 0, s, q0, q1);
    if (hipGetLastError() != hipSuccess) return launch_status("spv_enc_fc1_bwd_grouped prep pair");
    GemmParams p[2];
    for (int k = 0; k < 2; ++k) {   // (as spv_enc_fc1_wgrad builds them for a resident f16 image)
      const spv_fc1_bwd_args& c = k ? b : a;
      GemmParams& w = p[k];
      w = GemmParams{};
      w.A = c.dh_hi; w.lda = c.ld_dh; w.B = c.xb; w.ldb = c.ld_xb;
      w.rows = c.x->rows; w.counts_aligned = counts_aligned(c.x); w.col_off = c.x->col_off; w.n_cells = c.B; w.n_genes = c.G;
      w.C = c.dW; w.ldc = c.ldc; w.C2 = c.dW2; w.c_split_row = c.n_first; w.out_scale = c.scale_ws + 1;
      w.M = c.N1; w.N = c.G; w.K = c.B; w.k_per_split = (c.B + 63) & ~63; w.epi = EPI_STORE;
    }
    int rcg;
    if (t96a) rcg = launch_gemm_pair<GemmCfg<128, 96, 4, 1, true, true, SRC_PLAIN, SRC_GATHER, unsigned short, 1, 64, 4, 1, true>>(p[0], p[1], 1, s);
    else rcg = launch_gemm_pair<GemmCfg<128, 64, 4, 1, true, true, SRC_PLAIN, SRC_GATHER, unsigned short, 1, 64, 4, 1, true>>(p[0], p[1], 1, s);
    if (rcg != SPV_OK) return launch_status("spv_enc_fc1_bwd_grouped wgrad pair");
  }
  for (; i < n_groups; ++i) {
    const spv_fc1_bwd_args& a = g[i];
    int rc = spv_enc_fc1_bwd_prep(a.dh1, a.h1, a.B, a.N1, a.dh_hi, a.dh_lo, a.ld_dh, a.Bp, a.part, a.db, a.db2, a.n_first, a.scale_ws, stream);
    if (rc != SPV_OK) return rc;
    rc = spv_enc_fc1_wgrad(a.x, a.B, a.G, a.dh_hi, a.dh_lo, a.ld_dh, a.N1, a.nsplit, a.dW, a.dW2, a.n_first, a.ldc, a.xb, a.ld_xb, a.scale_ws, stream);
    if (rc != SPV_OK) return rc;
  }
  return SPV_OK;
}

extern "C" int spv_plan_invmap(const int32_t* idx0, int32_t B0, const int32_t* idx1, int32_t B1, int32_t* inv0, int32_t n0, int32_t* inv1,
                               int32_t n1, void* stream) {
  if (!idx0 || !idx1 || !inv0 || !inv1 || B0 <= 0 || B1 <= 0 || n0 <= 0 || n1 <= 0) return fail(SPV_ERR_ARG, "spv_plan_invmap: bad arguments%s");
  hipStream_t s = (hipStream_t)stream;
  // (fill kernels, not hipMemsetAsync: see zero_fill)
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0, s, inv0, (long)n0, -1);
  hipLaunchKernelGGL(fill_i32_kernel, dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0, s, inv1, (long)n1, -1);
  if (hipGetLastError() != hipSuccess) return fail(SPV_ERR_LAUNCH, "spv_plan_invmap: fill failed%s");
  const int Bm = B0 > B1 ? B0 : B1;
  hipLaunchKernelGGL(plan_invmap_kernel, dim3((Bm + 255) / 256, 2), dim3(256), 0, s, idx0, B0, idx1, B1, inv0, n0, inv1, n1);
  return launch_status("spv_plan_invmap");
}

static int check_plan(const spv_plan* p, const char* who) {
  if (!p || !p->ptr0 || !p->ind0 || !p->val0 || !p->ptr1 || !p->ind1 || !p->val1 || p->n0 <= 0 || p->n1 <= 0) return fail(SPV_ERR_ARG, "%s: incomplete plan", who);
  return SPV_OK;
}

extern "C" int spv_plan_argmax(const spv_plan* plan, const int32_t* idx0, const int32_t* idx1, const int32_t* inv0, const int32_t* inv1,
                               int32_t B0, int32_t B1, int32_t* partner0, int32_t* partner1, void* stream) {
  int rc = check_plan(plan, "spv_plan_argmax");
  if (rc) return rc;
  if (!idx0 || !idx1 || !inv0 || !inv1 || !partner0 || !partner1 || B0 <= 0 || B1 <= 0) return fail(SPV_ERR_ARG, "spv_plan_argmax: bad arguments%s");
  const int Bm = B0 > B1 ? B0 : B1;
  hipLaunchKernelGGL(plan_argmax_kernel, dim3((Bm + 255) / 256, 2), dim3(256), 0, (hipStream_t)stream, plan->ptr0, plan->ind0, plan->val0,
                     plan->ptr1, plan->ind1, plan->val1, idx0, idx1, inv0, inv1, B0, B1, plan->n0, plan->n1, partner0, partner1);
  return launch_status("spv_plan_argmax");
}

static int check_expert(const spv_plan_expert_args* a, const char* who) {
  if (!a) return fail(SPV_ERR_ARG, "%s: null arguments", who);
  int rc = check_plan(&a->plan, who);
  if (rc) return rc;
  if (a->B <= 0 || a->n <= 0 || a->n > 32) return fail(SPV_ERR_ARG, "%s: bad shape (latent dimension <= 32)", who);
  for (int g = 0; g < 2; ++g)
    if (!a->idx[g] || !a->inv[g] || !a->comp[g] || !a->stats[g] || !a->rowsum[g] || a->ld[g] < 2 * a->n || a->ld_expert[g] < 2 * a->n)
      return fail(SPV_ERR_ARG, "%s: null pointer / bad pitch", who);
  return SPV_OK;
}
extern "C" int spv_plan_expert_fwd(const spv_plan_expert_args* a, void* stream) {
  int rc = check_expert(a, "spv_plan_expert_fwd");
  if (rc) return rc;
  for (int g = 0; g < 2; ++g) if (!a->expert[g]) return fail(SPV_ERR_ARG, "spv_plan_expert_fwd: null output%s");
  hipLaunchKernelGGL(plan_expert_fwd_kernel, dim3((a->B + 7) / 8, 2), dim3(256), 0, (hipStream_t)stream, *a);
  return launch_status("spv_plan_expert_fwd");
}
extern "C" int spv_plan_expert_bwd(const spv_plan_expert_args* a, void* stream) {
  int rc = check_expert(a, "spv_plan_expert_bwd");
  if (rc) return rc;
  for (int g = 0; g < 2; ++g) if (!a->d_expert[g] || !a->d_stats[g]) return fail(SPV_ERR_ARG, "spv_plan_expert_bwd: null pointer%s");
  hipLaunchKernelGGL(plan_expert_bwd_kernel, dim3((a->B + 7) / 8, 2), dim3(256), 0, (hipStream_t)stream, *a);
  return launch_status("spv_plan_expert_bwd");
}

// f16(log1p(x)) of a whole resident count matrix and library = log(sum_g log1p(x)) per cell, once per data set:
// one workgroup per cell, fixed-order block reduction.
__global__ __launch_bounds__(256) void prepare_log1p_kernel(const void* X, long ldx, int col_off, int is_u16, int G, bf16_t* xb, long ld_xb,
                                                            float* library) {
  __shared__ float s_sum[256];
  const long cell = blockIdx.x;
  float acc = 0.f;
  for (int g = threadIdx.x; g < (int)ld_xb; g += 256) {
    float v = 0.f;
    if (g < G) {
      const float c = is_u16 ? (float)reinterpret_cast<const unsigned short*>(X)[cell * ldx + col_off + g]
                             : reinterpret_cast<const float*>(X)[cell * ldx + col_off + g];
      v = log1p_count(c);
    }
    xb[cell * ld_xb + g] = f2h(v);   // IEEE half: log1p(count) <= 11.1, 11 significant bits (spv_common.h)
    acc += v;
  }
  s_sum[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) library[cell] = __logf(s_sum[0]);
}

// "fp32" mode: bf16 hi / lo planes of log1p(x) interleaved in blocks of 32 genes ([32 hi words | 32 lo words] = 128 bytes per block:
// what one K tile of the LDS-DMA fc1 kernels takes from a row), ld_xb = 2 ldh words per row, + the library
__global__ __launch_bounds__(256) void prepare_log1p_split_kernel(const void* X, long ldx, int col_off, int is_u16, int G, bf16_t* xb, long ld_xb,
                                                                  float* library) {
  __shared__ float s_sum[256];
  const long cell = blockIdx.x;
  const long ldh = ld_xb / 2;
  float acc = 0.f;
  for (int g = threadIdx.x; g < (int)ldh; g += 256) {
    float v = 0.f;
    if (g < G) {
      const float c = is_u16 ? (float)reinterpret_cast<const unsigned short*>(X)[cell * ldx + col_off + g]
                             : reinterpret_cast<const float*>(X)[cell * ldx + col_off + g];
      v = log1p_count(c);
    }
    const bf16_t hi = f2bf(v);
    const long o = cell * ld_xb + (long)(g >> 5) * 64 + (g & 31);
    xb[o] = hi;
    xb[o + 32] = f2bf(v - bf2f(hi));
    acc += v;
  }
  s_sum[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) s_sum[threadIdx.x] += s_sum[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) library[cell] = __logf(s_sum[0]);
}
extern "C" int spv_prepare_log1p_split(const spv_counts* x, int32_t n_cells, int32_t G, uint16_t* xb, int64_t ld_xb, float* library, void* stream) {
  if (!x || !x->X || !xb || !library || n_cells <= 0 || G <= 0 || ld_xb < 2 * (((long)G + 31) & ~31L) || (ld_xb % 128)) return fail(SPV_ERR_ARG, "spv_prepare_log1p_split: bad arguments (ld_xb a multiple of 128 words, >= 2 round_up(G, 32))%s");
  if (x->dtype != SPV_COUNT_U16 && x->dtype != SPV_COUNT_F32) return fail(SPV_ERR_ARG, "spv_prepare_log1p_split: unknown count dtype%s");
  hipLaunchKernelGGL(prepare_log1p_split_kernel, dim3((unsigned)n_cells), dim3(256), 0, (hipStream_t)stream, x->X, (long)x->ld, x->col_off,
                     (int)(x->dtype == SPV_COUNT_U16), G, (bf16_t*)xb, (long)ld_xb, library);
  return launch_status("spv_prepare_log1p_split");
}

extern "C" int spv_prepare_log1p(const spv_counts* x, int32_t n_cells, int32_t G, uint16_t* xb, int64_t ld_xb, float* library, void* stream) {
  if (!x || !x->X || !xb || !library || n_cells <= 0 || G <= 0 || ld_xb < G) return fail(SPV_ERR_ARG, "spv_prepare_log1p: bad arguments%s");
  if (x->dtype != SPV_COUNT_U16 && x->dtype != SPV_COUNT_F32) return fail(SPV_ERR_ARG, "spv_prepare_log1p: unknown count dtype%s");
  hipLaunchKernelGGL(prepare_log1p_kernel, dim3((unsigned)n_cells), dim3(256), 0, (hipStream_t)stream, x->X, (long)x->ld, x->col_off,
                     (int)(x->dtype == SPV_COUNT_U16), G, xb, (long)ld_xb, library);
  return launch_status("spv_prepare_log1p");
}

