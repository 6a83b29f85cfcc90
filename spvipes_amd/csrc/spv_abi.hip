// C-ABI entry points of libspvipes_hip.so (see include/spvipes_hip.h).  gfx950 only.
#include "../../include/spvipes_hip.h"
#include "spv_common.h"
#include "spv_gemm.h"
#include "spv_decoder.h"

#include <cstdio>
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

extern "C" int spv_version(void) { return 1; }
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

extern "C" int spv_pack_bf16(const float* src, int64_t ld_src, int32_t R, int32_t C, const float* extra_col,
                             int32_t extra_one, uint16_t* dst_hi, uint16_t* dst_lo, int64_t ld_dst,
                             int32_t dst_col_off, int32_t Rp, int32_t cslot, void* stream) {
  if (!src || !dst_hi || R < 0 || C < 0 || Rp < R || cslot < C + ((extra_col || extra_one) ? 1 : 0) ||
      dst_col_off + cslot > ld_dst)
    return fail(SPV_ERR_ARG, "spv_pack_bf16: bad shape%s");
  const long total = (long)Rp * cslot;
  if (total == 0) return SPV_OK;
  hipLaunchKernelGGL(pack_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (long)ld_src, R, C, extra_col, extra_one, dst_hi, dst_lo, (long)ld_dst, dst_col_off, Rp, cslot);
  return launch_status("spv_pack_bf16");
}

// ---------------------------------------------------------------------------------------------
// encoder fc1
// ---------------------------------------------------------------------------------------------
__global__ void fc1_epilogue_kernel(const float* slabs, const float* rowsum_ws, int splits, int B, int N1, const float* bias,
                                    float* h1, float* library) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const long total = (long)B * N1;
  if (idx < total) {
    float v = bias[idx % N1];
    for (int s = 0; s < splits; ++s) v += slabs[(long)s * total + idx];
    h1[idx] = fmaxf(v, 0.f);  // relu(fc1(x)), nn/networks.py:119
  }
  if (idx < B) {
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += rowsum_ws[(long)s * B + idx];
    library[idx] = __logf(v);  // log(sum of log1p(x)), module/spVIPESmodule.py:435
  }
}

template <typename CT, int NSPLIT>
static int fc1_fwd_dispatch(const GemmParams& p, int N1, int splits, hipStream_t s) {
  if (N1 <= 32) return launch_gemm<GemmCfg<128, 32, 4, 1, false, false, SRC_COUNTS, SRC_PLAIN, CT, NSPLIT>>(p, splits, s);
  if (N1 <= 128) return launch_gemm<GemmCfg<64, 128, 2, 2, false, false, SRC_COUNTS, SRC_PLAIN, CT, NSPLIT>>(p, splits, s);
  return launch_gemm<GemmCfg<64, 256, 1, 4, false, false, SRC_COUNTS, SRC_PLAIN, CT, NSPLIT>>(p, splits, s);
}

static int fc1_bn(int N1) { return N1 <= 32 ? 32 : (N1 <= 128 ? 128 : 256); }

extern "C" int spv_enc_fc1_fwd(const spv_counts* x, int32_t B, int32_t G, const uint16_t* W1_hi, const uint16_t* W1_lo,
                               int64_t ldw, int32_t N1, const float* bias, int32_t nsplit, int32_t splits, float* slabs,
                               float* rowsum_ws, float* h1, float* library, void* stream) {
  if (!x || !x->X || !W1_hi || !bias || !slabs || !rowsum_ws || !h1 || !library) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: null pointer%s");
  if (B <= 0 || G <= 0 || N1 <= 0 || splits <= 0 || (ldw % 32) != 0 || ldw < ((G + 31) & ~31)) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: bad shape%s");
  if (nsplit != 1 && nsplit != 3) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: nsplit must be 1 or 3%s");
  if (nsplit == 3 && !W1_lo) return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: nsplit=3 needs W1_lo%s");
  GemmParams p{};
  p.A = x->X; p.A_lo = nullptr; p.lda = x->ld;
  p.B = W1_hi; p.B_lo = W1_lo; p.ldb = ldw;
  p.rows = x->rows; p.col_off = x->col_off; p.n_cells = B; p.n_genes = G;
  p.rowsum = rowsum_ws;
  p.C = slabs; p.ldc = N1; p.slab_stride = (long)B * N1;
  p.M = B; p.N = N1; p.K = G;
  const int ktiles = (G + 31) / 32;
  p.k_per_split = ((ktiles + splits - 1) / splits) * 32;
  p.epi = EPI_STORE;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (x->dtype == SPV_COUNT_U16) rc = (nsplit == 3) ? fc1_fwd_dispatch<unsigned short, 3>(p, N1, splits, s) : fc1_fwd_dispatch<unsigned short, 1>(p, N1, splits, s);
  else if (x->dtype == SPV_COUNT_F32) rc = (nsplit == 3) ? fc1_fwd_dispatch<float, 3>(p, N1, splits, s) : fc1_fwd_dispatch<float, 1>(p, N1, splits, s);
  else return fail(SPV_ERR_ARG, "spv_enc_fc1_fwd: unknown count dtype%s");
  if (rc != SPV_OK) return launch_status("spv_enc_fc1_fwd gemm");
  const long total = (long)B * N1;
  hipLaunchKernelGGL(fc1_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, slabs, rowsum_ws, splits, B, N1, bias, h1, library);
  return launch_status("spv_enc_fc1_fwd epilogue");
}

extern "C" int spv_enc_fc1_wgrad(const spv_counts* x, int32_t B, int32_t G, const uint16_t* dh_hi, const uint16_t* dh_lo,
                                 int64_t ld_dh, int32_t N1, int32_t nsplit, float* dW, int64_t ldc, void* stream) {
  if (!x || !x->X || !dh_hi || !dW) return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: null pointer%s");
  if (B <= 0 || G <= 0 || N1 <= 0 || ld_dh < ((N1 + 127) & ~127) || (ld_dh % 8) != 0 || ldc < G) return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: bad shape%s");
  if (nsplit != 1 && nsplit != 3) return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: nsplit must be 1 or 3%s");
  if (nsplit == 3 && !dh_lo) return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: nsplit=3 needs dh_lo%s");
  GemmParams p{};
  p.A = dh_hi; p.A_lo = dh_lo; p.lda = ld_dh;
  p.B = x->X; p.B_lo = nullptr; p.ldb = x->ld;
  p.rows = x->rows; p.col_off = x->col_off; p.n_cells = B; p.n_genes = G;
  p.rowsum = nullptr;
  p.C = dW; p.ldc = ldc; p.slab_stride = 0;
  p.M = N1; p.N = G; p.K = B;
  p.k_per_split = (B + 31) & ~31;
  p.epi = EPI_STORE;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (x->dtype == SPV_COUNT_U16) {
    rc = (nsplit == 3) ? launch_gemm<GemmCfg<128, 64, 4, 1, true, true, SRC_PLAIN, SRC_COUNTS, unsigned short, 3>>(p, 1, s)
                       : launch_gemm<GemmCfg<128, 64, 4, 1, true, true, SRC_PLAIN, SRC_COUNTS, unsigned short, 1>>(p, 1, s);
  } else if (x->dtype == SPV_COUNT_F32) {
    rc = (nsplit == 3) ? launch_gemm<GemmCfg<128, 64, 4, 1, true, true, SRC_PLAIN, SRC_COUNTS, float, 3>>(p, 1, s)
                       : launch_gemm<GemmCfg<128, 64, 4, 1, true, true, SRC_PLAIN, SRC_COUNTS, float, 1>>(p, 1, s);
  } else return fail(SPV_ERR_ARG, "spv_enc_fc1_wgrad: unknown count dtype%s");
  (void)rc;
  return launch_status("spv_enc_fc1_wgrad");
}

// ---------------------------------------------------------------------------------------------
// plain GEMM
// ---------------------------------------------------------------------------------------------
template <bool A_KMAJ, int NSPLIT>
static int gemm_dispatch(const GemmParams& p, int splits, hipStream_t s) {
  if (p.N <= 32) return launch_gemm<GemmCfg<128, 32, 4, 1, A_KMAJ, true, SRC_PLAIN, SRC_PLAIN, unsigned short, NSPLIT>>(p, splits, s);
  return launch_gemm<GemmCfg<64, 320, 2, 2, A_KMAJ, true, SRC_PLAIN, SRC_PLAIN, unsigned short, NSPLIT>>(p, splits, s);
}

extern "C" int spv_gemm_bf16(int32_t a_kmajor, const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda, const uint16_t* B_hi,
                             const uint16_t* B_lo, int64_t ldb, float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                             int32_t nsplit, int32_t splits, int64_t slab_stride, void* stream) {
  if (!A_hi || !B_hi || !C) return fail(SPV_ERR_ARG, "spv_gemm_bf16: null pointer%s");
  if (M <= 0 || N <= 0 || K <= 0 || splits <= 0 || (lda % 8) || (ldb % 8)) return fail(SPV_ERR_ARG, "spv_gemm_bf16: bad shape%s");
  if (nsplit != 1 && nsplit != 3) return fail(SPV_ERR_ARG, "spv_gemm_bf16: nsplit must be 1 or 3%s");
  if (nsplit == 3 && (!A_lo || !B_lo)) return fail(SPV_ERR_ARG, "spv_gemm_bf16: nsplit=3 needs lo images%s");
  const int bn = (N <= 32) ? 32 : 320;
  if (ldb < ((N + bn - 1) / bn) * bn) return fail(SPV_ERR_ARG, "spv_gemm_bf16: B image narrower than the N tiling%s");
  GemmParams p{};
  p.A = A_hi; p.A_lo = A_lo; p.lda = lda;
  p.B = B_hi; p.B_lo = B_lo; p.ldb = ldb;
  p.C = C; p.ldc = ldc; p.slab_stride = slab_stride;
  p.M = M; p.N = N; p.K = K;
  const int ktiles = (K + 31) / 32;
  p.k_per_split = ((ktiles + splits - 1) / splits) * 32;
  p.epi = EPI_STORE;
  hipStream_t s = (hipStream_t)stream;
  if (a_kmajor) { if (nsplit == 3) gemm_dispatch<true, 3>(p, splits, s); else gemm_dispatch<true, 1>(p, splits, s); }
  else { if (nsplit == 3) gemm_dispatch<false, 3>(p, splits, s); else gemm_dispatch<false, 1>(p, splits, s); }
  return launch_status("spv_gemm_bf16");
}

// ---------------------------------------------------------------------------------------------
// decoder / likelihood
// ---------------------------------------------------------------------------------------------
static int to_dec(const spv_dec_params* q, DecParams& p) {
  if (!q) return fail(SPV_ERR_ARG, "decoder: null params%s");
  p.X = q->X; p.ldx = q->ldx; p.rows = q->rows; p.col_off = q->col_off; p.count_is_u16 = q->count_is_u16;
  p.B = q->B; p.G = q->G; p.Bp = q->Bp; p.Gp = q->Gp;
  p.Wm_hi = q->Wm_hi; p.Wm_lo = q->Wm_lo; p.KMp = q->KMp; p.ksteps_m = q->ksteps_m;
  p.Am_hi = q->Am_hi; p.Am_lo = q->Am_lo;
  p.Wps_hi = q->Wps_hi; p.Wps_lo = q->Wps_lo; p.Aps_hi = q->Aps_hi; p.Aps_lo = q->Aps_lo;
  p.gene_tab = (const float4*)q->gene_tab; p.cnt_tab = (const float2*)q->cnt_tab;
  p.a_p = q->a_p; p.a_s = q->a_s; p.lse_p = q->lse_p; p.lse_s = q->lse_s; p.w_row = q->w_row;
  p.gene_splits = q->gene_splits; p.genes_per_split = q->genes_per_split;
  p.part_max_p = q->part_max_p; p.part_sum_p = q->part_sum_p; p.part_max_s = q->part_max_s; p.part_sum_s = q->part_sum_s;
  p.rec_part = q->rec_part; p.tp_part = q->tp_part; p.ts_part = q->ts_part; p.dtheta_part = q->dtheta_part;
  p.dL = q->dL; p.tP = q->tP; p.tS = q->tS; p.ldg = q->ldg; p.grads_f32 = q->grads_f32;
  if (p.B <= 0 || p.G <= 0 || p.Bp < p.B || p.Gp < p.G || (p.Bp % DEC_CELLS_PER_WG) || (p.Gp % 32))
    return fail(SPV_ERR_ARG, "decoder: Bp must be a multiple of 128 and Gp of 32%s");
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

extern "C" int spv_dec_lse(const spv_dec_params* q, const float* library, void* stream) {
  DecParams p;
  int rc = to_dec(q, p);
  if (rc != SPV_OK) return rc;
  if (!library || !p.part_max_p || !p.part_sum_p || !p.part_max_s || !p.part_sum_s || !p.lse_p || !p.lse_s || !p.a_p || !p.a_s)
    return fail(SPV_ERR_ARG, "spv_dec_lse: null pointer%s");
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dec_lse_kernel, dim3(p.Bp / DEC_CELLS_PER_WG, p.gene_splits), dim3(256), 0, s, p);
  hipLaunchKernelGGL(dec_lse_combine_kernel, dim3((p.Bp + 255) / 256), dim3(256), 0, s, p.part_max_p, p.part_sum_p,
                     p.part_max_s, p.part_sum_s, p.gene_splits, p.Bp, p.B, library, (float*)p.lse_p, (float*)p.lse_s,
                     (float*)p.a_p, (float*)p.a_s);
  return launch_status("spv_dec_lse");
}

template <int NSPLIT, bool TRAIN, typename GT>
static int nb_launch(const DecParams& p, hipStream_t s) {
  const size_t lds = (size_t)32 * (p.KMp + WM_PAD) * 2 * (NSPLIT == 3 ? 2 : 1) + 4 * 32 * sizeof(float);
  dim3 grid(p.Bp / DEC_CELLS_PER_WG, p.gene_splits);
  if constexpr (NSPLIT == 3) {
    hipLaunchKernelGGL((dec_nb_kernel<1, 3, TRAIN, GT>), grid, dim3(256), lds, s, p);
  } else {
    switch (p.ksteps_m) {
      case 17: hipLaunchKernelGGL((dec_nb_kernel<17, 1, TRAIN, GT>), grid, dim3(256), lds, s, p); break;
      case 18: hipLaunchKernelGGL((dec_nb_kernel<18, 1, TRAIN, GT>), grid, dim3(256), lds, s, p); break;
      case 19: hipLaunchKernelGGL((dec_nb_kernel<19, 1, TRAIN, GT>), grid, dim3(256), lds, s, p); break;
      case 20: hipLaunchKernelGGL((dec_nb_kernel<20, 1, TRAIN, GT>), grid, dim3(256), lds, s, p); break;
      default: return fail(SPV_ERR_UNSUPPORTED, "spv_dec_nb_fwd: ksteps_m must be 17..20 in bf16 mode%s");
    }
  }
  return SPV_OK;
}

extern "C" int spv_dec_nb_fwd(const spv_dec_params* q, int32_t nsplit, int32_t train, void* stream) {
  DecParams p;
  int rc = to_dec(q, p);
  if (rc != SPV_OK) return rc;
  if (!p.X || !p.Wm_hi || !p.Am_hi || !p.gene_tab || !p.cnt_tab || !p.a_p || !p.a_s || !p.w_row || !p.rec_part)
    return fail(SPV_ERR_ARG, "spv_dec_nb_fwd: null pointer%s");
  if (nsplit != 1 && nsplit != 3) return fail(SPV_ERR_ARG, "spv_dec_nb_fwd: nsplit must be 1 or 3%s");
  if (nsplit == 3 && (!p.Wm_lo || !p.Am_lo)) return fail(SPV_ERR_ARG, "spv_dec_nb_fwd: nsplit=3 needs lo images%s");
  if ((p.KMp % 16) || p.ksteps_m * 16 > p.KMp || p.ksteps_m <= 0) return fail(SPV_ERR_ARG, "spv_dec_nb_fwd: bad KMp / ksteps_m%s");
  if (train && (!p.dL || !p.tP || !p.tS || !p.tp_part || !p.ts_part || !p.dtheta_part || p.ldg < p.Gp || (p.ldg % 8)))
    return fail(SPV_ERR_ARG, "spv_dec_nb_fwd: training outputs missing%s");
  hipStream_t s = (hipStream_t)stream;
  if (train) {
    if (p.grads_f32) rc = (nsplit == 3) ? nb_launch<3, true, float>(p, s) : nb_launch<1, true, float>(p, s);
    else rc = (nsplit == 3) ? nb_launch<3, true, bf16_t>(p, s) : nb_launch<1, true, bf16_t>(p, s);
  } else {
    rc = (nsplit == 3) ? nb_launch<3, false, bf16_t>(p, s) : nb_launch<1, false, bf16_t>(p, s);
  }
  if (rc != SPV_OK) return rc;
  return launch_status("spv_dec_nb_fwd");
}

extern "C" int spv_dec_softmax_bwd(const spv_dec_params* q, const float* Tp, const float* Ts, void* stream) {
  DecParams p;
  int rc = to_dec(q, p);
  if (rc != SPV_OK) return rc;
  if (!Tp || !Ts || !p.tP || !p.tS || !p.lse_p || !p.lse_s) return fail(SPV_ERR_ARG, "spv_dec_softmax_bwd: null pointer%s");
  dim3 grid(p.Bp / DEC_CELLS_PER_WG, p.gene_splits);
  hipStream_t s = (hipStream_t)stream;
  if (p.grads_f32) hipLaunchKernelGGL(dec_softmax_bwd_kernel<float>, grid, dim3(256), 0, s, p, Tp, Ts);
  else hipLaunchKernelGGL(dec_softmax_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, p, Tp, Ts);
  return launch_status("spv_dec_softmax_bwd");
}
