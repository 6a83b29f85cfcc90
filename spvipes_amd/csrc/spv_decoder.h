// Decoder + negative-binomial-mixture likelihood kernels (A6-A8 of SURVEY.md 8a).
//
// Reference semantics (file:line into /root/reference/src/spVIPES):
//   LinearDecoderSPVIPE.forward  nn/networks.py:314-325
//       rate_k = exp(library) * softmax_G(BN_G(z_k W_k^T)),  k in {private, shared}
//       logits = cat(relu(BN(cat(z) W_a^T + b_a)), z) W_m^T + b_m
//   loss                         module/spVIPESmodule.py:817-824
//       rec_b = - sum_g log_mixture_nb(x = log1p(count), mu1 = rate_private, mu2 = rate_shared,
//                                      theta = exp(px_r), pi_logits = logits)
//
// How the work is laid out on the chip
//   * Training-mode BatchNorm over the batch of a *linear* map is itself affine in z, so the host
//     folds it into effective weights W'_g = (gamma_g / sigma_g) w_g and a bias c_g (batch mean
//     and variance of z W^T follow from mean(z) and cov(z)); the kernels only ever see
//     y_k[b,g] = z_k[b] . W'_k[g] + c_k[g].  The bias rides in an extra "ones" K column.
//   * Tiles are 32 genes (MFMA rows, accumulator registers) x 32 cells (MFMA columns, lanes):
//     per-cell reductions over genes (softmax statistics, rec_b, the row sums the softmax
//     backward needs) are in-lane accumulations across the gene loop; only the per-gene sum for
//     d px_r crosses lanes.
//   * The only [B,G] arrays in HBM are 16-bit: the mixing logits (one plain MFMA GEMM, f16) and,
//     in training, the three per-element gradients the backward GEMMs need (d/d logits,
//     d/d log mu1, d/d log mu2, bf16) so that the backward pass re-evaluates no transcendental.
//     ("fp32" precision mode keeps all four as fp32.)
//   * lgamma / digamma terms depend only on (count, gene): they come from a per-step table
//     tab[c][g] = { lgamma(x+theta_g) - lgamma(theta_g) - lgamma(x+1),  psi(x+theta_g) - psi(theta_g) },
//     x = log1p(c), built by nb_tables_kernel for c < NB_CMAX; larger counts are evaluated inline.
#pragma once
#include "spv_common.h"

namespace spv {

constexpr int NB_CMAX = 64;      // table rows (counts 0..63)
constexpr int DEC_KP = 16;       // K slots of the private regressor  (n_p + 1 bias <= 16)
constexpr int DEC_KS = 32;       // K slots of the shared regressor   (n_s + 1 bias <= 32)
constexpr int DEC_KPS = DEC_KP + DEC_KS;
constexpr int DEC_CELLS_PER_WG = 128;

struct DecParams {
  // counts
  const void* X; long ldx; const int* rows; int col_off; int count_is_u16;
  int B, G;          // logical cells in the minibatch / genes of the group
  int Bp, Gp;        // padded extents of the packed images (multiples of 128)
  // packed operands (bf16, zero padded)
  const void* logits; int n_gene_tiles; int logits_f32;               // tiled [Bp/32][Gp/32][64][16] mixing logits (f16 | f32)
  const bf16_t* Wps_hi; const bf16_t* Wps_lo;                         // [Gp][48]   W'_p|c_p|0.. W'_s|c_s|0..
  const bf16_t* Aps_hi; const bf16_t* Aps_lo;                         // [Bp][48]   z_p|1|0..   z_s|1|0..
  // per gene / per cell vectors
  const float4* gene_tab;       // [Gp] {theta, lt = log(theta+eps), lt + theta/(theta+eps), theta * lt / ln 2}
  const float2* cnt_tab;        // [NB_CMAX][Gp] {F, Psi}
  const float* a_p; const float* a_s;   // [Bp] library - lse_k
  const float* lse_p; const float* lse_s;
  const float* w_row;           // [Bp] weight of each cell in the scalar loss (0 for padding)
  // outputs
  int gene_splits; int genes_per_split;      // multiples of 32
  float* part_max_p; float* part_sum_p; float* part_max_s; float* part_sum_s;  // [splits][Bp]  (lse)
  float* rec_part; float* tp_part; float* ts_part;                             // [nb_splits][Bp]  (nb)
  float* dtheta_part;           // [Bp/64][Gp]  one partial row per 64-cell workgroup
  void* dL; void* tP; void* tS; int grads_f32;                                 // tiled like logits; bf16 or f32
  int nb_splits; int nb_genes_per_split;     // gene splits of the likelihood kernel (multiple of 32, <= NB_GSPL_MAX)
  int nb_cell_tiles;                         // 64-cell tiles a likelihood workgroup walks with ONE staged weight slice (>= 1)
};

// ---- per-gene tables ----------------------------------------------------------
__device__ __forceinline__ void nb_tables_body(const float* px_r, int G, int Gp, float4* gene_tab, float2* cnt_tab) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x;
  const int c = blockIdx.y;
  if (g >= Gp) return;
  if (g >= G) {
    if (c == 0) gene_tab[g] = make_float4(1.f, 0.f, 1.f, 0.f);
    cnt_tab[(long)c * Gp + g] = make_float2(0.f, 0.f);
    return;
  }
  const float theta = fast_exp(px_r[g]);  // px_r = exp(param), module/spVIPESmodule.py:758
  if (c == 0) {
    const float lt = fast_log(theta + SPV_EPS_NB);
    gene_tab[g] = make_float4(theta, lt, fmaf(theta, fast_rcp(theta + SPV_EPS_NB), lt), theta * lt * 1.4426950408889634f);   // (the likelihood kernel works in base-2 units)
    cnt_tab[g] = make_float2(0.f, 0.f);  // x = 0: lgamma terms cancel exactly
    return;
  }
  const float x = log1p_count((float)c);
  const LgammaDigamma t = lgamma_digamma(theta), xt = lgamma_digamma(x + theta), x1 = lgamma_digamma(x + 1.0f);
  cnt_tab[(long)c * Gp + g] = make_float2(xt.lg - t.lg - x1.lg, xt.dg - t.dg);
}
__global__ void nb_tables_kernel(const float* px_r, int G, int Gp, float4* gene_tab, float2* cnt_tab) { nb_tables_body(px_r, G, Gp, gene_tab, cnt_tab); }
// "pair" launches (here and below): both groups of a step in ONE grid, group = blockIdx.z -- workgroups outside a group's own grid extent
// leave at once, the others run the single-group body unchanged (same blockIdx.x / .y, same bits)
struct NbTabArgs { const float* px_r; int G, Gp; float4* gene_tab; float2* cnt_tab; };
__global__ void nb_tables_pair_kernel(NbTabArgs a0, NbTabArgs a1) {
  const NbTabArgs a = blockIdx.z ? a1 : a0;
  nb_tables_body(a.px_r, a.G, a.Gp, a.gene_tab, a.cnt_tab);   // (its own g >= Gp test covers the x extent)
}

// Count words of 4 consecutive genes g..g+3 (g % 4 == 0) of one cell, kept undecoded so that the load
// needs no wait until its values are used.  The access mode is uniform per launch:
enum { CNT_U16_ALIGNED = 0, CNT_U16_ANY = 1, CNT_F32 = 2 };
template <int CM> struct RawCounts;
template <> struct RawCounts<CNT_U16_ALIGNED> { u2v w; };          // one 8-byte load (matrix base, ld, col_off 4-aligned)
template <> struct RawCounts<CNT_U16_ANY> { unsigned short w[4]; };
template <> struct RawCounts<CNT_F32> { float w[4]; };

// branch-free: lanes (or genes) with nothing to read fetch element 0 of the matrix and decode() masks them
template <int CM>
__device__ __forceinline__ RawCounts<CM> load_counts4_raw(const DecParams& p, long row, int g, bool cell_ok) {
  RawCounts<CM> r;
  const long base = row * p.ldx + p.col_off + g;
  if constexpr (CM == CNT_U16_ALIGNED) {
    const bool full = cell_ok && (g + 4 <= p.G);
    r.w = *reinterpret_cast<const u2v*>(reinterpret_cast<const unsigned short*>(p.X) + (full ? base : 0));
  } else if constexpr (CM == CNT_U16_ANY) {
#pragma unroll
    for (int j = 0; j < 4; ++j) r.w[j] = reinterpret_cast<const unsigned short*>(p.X)[(cell_ok && g + j < p.G) ? base + j : 0];
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) r.w[j] = reinterpret_cast<const float*>(p.X)[(cell_ok && g + j < p.G) ? base + j : 0];
  }
  return r;
}
template <int CM>
__device__ __forceinline__ void decode_counts4(const DecParams& p, const RawCounts<CM>& r, long row, int g, bool cell_ok, float (&c)[4]) {
  if constexpr (CM == CNT_U16_ALIGNED) {
    const bool full = cell_ok && (g + 4 <= p.G);
    c[0] = full ? (float)(r.w[0] & 0xFFFFu) : 0.f; c[1] = full ? (float)(r.w[0] >> 16) : 0.f;
    c[2] = full ? (float)(r.w[1] & 0xFFFFu) : 0.f; c[3] = full ? (float)(r.w[1] >> 16) : 0.f;
    const bool partial = cell_ok && g < p.G && g + 4 > p.G;  // only where G % 4 != 0: the last genes of the group
    if (__builtin_expect(__any(partial), 0)) {
      if (partial) {
        const unsigned short* src = reinterpret_cast<const unsigned short*>(p.X) + row * p.ldx + p.col_off + g;
#pragma unroll
        for (int j = 0; j < 3; ++j) if (g + j < p.G) c[j] = (float)src[j];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) c[j] = (cell_ok && g + j < p.G) ? (float)r.w[j] : 0.f;
  }
}

// y_p / y_s tiles (32 genes x 32 cells) with split-bf16 operands straight from L2 (K = 16 / 32):
// A fragment = W' rows (natural [g][k], PsW below), B fragment = the wave's resident z fragments.
struct PsFrags { s8v hi[3], lo[3]; };
__device__ __forceinline__ void load_ps_cell_frags(const DecParams& p, int cell0, int lane, PsFrags& f) {
  const long off = (long)(cell0 + (lane & 31)) * DEC_KPS + 8 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    f.hi[s] = *reinterpret_cast<const s8v*>(p.Aps_hi + off + 16 * s);
    f.lo[s] = *reinterpret_cast<const s8v*>(p.Aps_lo + off + 16 * s);
  }
}
struct PsW { s8v hi[3], lo[3]; };  // gene-side fragments of one 32-gene tile (K slices 0: private, 1..2: shared)
__device__ __forceinline__ void load_ps_w(const DecParams& p, int g0, int lane, PsW& w) {
  const long off = (long)(g0 + (lane & 31)) * DEC_KPS + 8 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    w.hi[s] = *reinterpret_cast<const s8v*>(p.Wps_hi + off + 16 * s);
    w.lo[s] = *reinterpret_cast<const s8v*>(p.Wps_lo + off + 16 * s);
  }
}

constexpr float NB_LOG2E_C = 1.4426950408889634f;
// ---- pass 1: per-cell log-sum-exp over genes of y_p and y_s (softmax denominators) ------------
__device__ __forceinline__ void dec_lse_body(const DecParams& p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int cell0 = blockIdx.x * DEC_CELLS_PER_WG + 32 * wave;
  const int split = blockIdx.y;
  PsFrags cf;
  load_ps_cell_frags(p, cell0, lane, cf);
  float mp = -INFINITY, sp = 0.f, ms = -INFINITY, ss = 0.f;
  const int gbeg = split * p.genes_per_split;
  int gend = gbeg + p.genes_per_split;
  if (gend > p.Gp) gend = p.Gp;
  if (gend > ((p.G + 31) & ~31)) gend = (p.G + 31) & ~31;
  const int ntile = (gend - gbeg) >> 5;
  PsW wA;
  if (ntile > 0) load_ps_w(p, gbeg, lane, wA);
  for (int t = 0; t < ntile; ++t) {
    const int g0 = gbeg + 32 * t;
    f16v yp, ys;
#pragma unroll
    for (int q = 0; q < 16; ++q) { yp[q] = 0.f; ys[q] = 0.f; }
    yp = mfma32_split<3>(wA.hi[0], wA.lo[0], cf.hi[0], cf.lo[0], yp);
    ys = mfma32_split<3>(wA.hi[1], wA.lo[1], cf.hi[1], cf.lo[1], ys);
    ys = mfma32_split<3>(wA.hi[2], wA.lo[2], cf.hi[2], cf.lo[2], ys);
    PsW wB;
    load_ps_w(p, gbeg + 32 * min(t + 1, ntile - 1), lane, wB);  // next tile's fragments fly under this tile's exps
    // 2 exp per element is the kernel's arithmetic; everything around it is kept to one fma and half a max per element: the gene-validity
    // masks only exist in a tile that crosses G (wave-uniform test), the maxima go through v_max3, and exp(y - m) is exp2(fma(y, log2 e, -m log2 e))
    if (g0 + 32 > p.G) {
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const bool ok = g0 + crow(q, h) < p.G;
        yp[q] = ok ? yp[q] : -INFINITY;
        ys[q] = ok ? ys[q] : -INFINITY;
      }
    }
    float tmp = fmaxf(yp[0], yp[1]), tms = fmaxf(ys[0], ys[1]);
#pragma unroll
    for (int q = 2; q < 16; q += 2) {
      tmp = __builtin_fmaxf(__builtin_fmaxf(tmp, yp[q]), yp[q + 1]);   // (pairs fold into v_max3_f32)
      tms = __builtin_fmaxf(__builtin_fmaxf(tms, ys[q]), ys[q + 1]);
    }
    const float nmp = fmaxf(mp, tmp), nms = fmaxf(ms, tms);
    // a half-wave can see only masked genes in the last tile: keep exp() away from (-inf) - (-inf)
    const float smp = (nmp == -INFINITY) ? 0.f : nmp, sms = (nms == -INFINITY) ? 0.f : nms;
    sp *= fast_exp(mp - smp);
    ss *= fast_exp(ms - sms);
    const float np2 = -smp * NB_LOG2E_C, ns2 = -sms * NB_LOG2E_C;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      sp += __builtin_amdgcn_exp2f(fmaf(yp[q], NB_LOG2E_C, np2));
      ss += __builtin_amdgcn_exp2f(fmaf(ys[q], NB_LOG2E_C, ns2));
    }
    mp = nmp; ms = nms;
    wA = wB;
  }
  // merge the two lane halves (same cell, different genes)
  {
    const float omp = other_half(mp), osp = other_half(sp), oms = other_half(ms), oss = other_half(ss);
    const float nmp = fmaxf(mp, omp), nms = fmaxf(ms, oms);
    const float smp = (nmp == -INFINITY) ? 0.f : nmp, sms = (nms == -INFINITY) ? 0.f : nms;
    sp = sp * fast_exp(mp - smp) + osp * fast_exp(omp - smp);
    ss = ss * fast_exp(ms - sms) + oss * fast_exp(oms - sms);
    mp = nmp; ms = nms;
  }
  if (h == 0) {
    const long o = (long)split * p.Bp + cell0 + lane;
    p.part_max_p[o] = mp; p.part_sum_p[o] = sp;
    p.part_max_s[o] = ms; p.part_sum_s[o] = ss;
  }
}
__global__ __launch_bounds__(256) void dec_lse_kernel(DecParams p) { dec_lse_body(p); }
__global__ __launch_bounds__(256) void dec_lse_pair_kernel(DecParams p0, DecParams p1) {
  const DecParams p = blockIdx.z ? p1 : p0;
  if ((int)blockIdx.x * DEC_CELLS_PER_WG >= p.Bp || (int)blockIdx.y >= p.gene_splits) return;   // (workgroup-uniform)
  dec_lse_body(p);
}

// lse_k[b] = log sum_splits ...;  a_k[b] = library[b] - lse_k[b]   (log of exp(library) * softmax)
// block = 64 cells x 4 split groups: group y merges splits y, y+4, ... in order (online log-sum-exp), the four group
// results are merged in order
// (branch-free: with data-dependent branches between them the loads of the unrolled merge loop cannot be issued ahead of the chain)
__device__ __forceinline__ void lse_merge(float& m, float& s, float m2, float s2) {
  const float mn = fmaxf(m, m2);
  const float base = (mn == -INFINITY) ? 0.f : mn;   // both empty: keep exp() away from (-inf) - (-inf)
  s = s * __expf(m - base) + s2 * __expf(m2 - base);
  m = mn;
}
__device__ __forceinline__ void dec_lse_combine_body(const float* pmp, const float* psp, const float* pms, const float* pss, int splits,
                                                     int Bp, int B, const float* library, float* lse_p, float* lse_s, float* a_p, float* a_s) {
  __shared__ float s_m[2][4][64], s_s[2][4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int b = blockIdx.x * 64 + tx;
  float mp = -INFINITY, ms = -INFINITY, sp = 0.f, ss = 0.f;
  if (b < Bp) {
#pragma unroll 4
    for (int k = ty; k < splits; k += 4) {
      const long o = (long)k * Bp + b;
      lse_merge(mp, sp, pmp[o], psp[o]);
      lse_merge(ms, ss, pms[o], pss[o]);
    }
  }
  s_m[0][ty][tx] = mp; s_s[0][ty][tx] = sp; s_m[1][ty][tx] = ms; s_s[1][ty][tx] = ss;
  __syncthreads();
  if (ty == 0 && b < Bp) {
    for (int y = 1; y < 4; ++y) {
      lse_merge(mp, sp, s_m[0][y][tx], s_s[0][y][tx]);
      lse_merge(ms, ss, s_m[1][y][tx], s_s[1][y][tx]);
    }
    const float lp = mp + __logf(sp), ls = ms + __logf(ss);
    const float lib = (b < B) ? library[b] : 0.f;
    lse_p[b] = lp; lse_s[b] = ls;
    a_p[b] = lib - lp; a_s[b] = lib - ls;
  }
}
__global__ __launch_bounds__(256) void dec_lse_combine_kernel(const float* pmp, const float* psp, const float* pms, const float* pss, int splits,
                                                              int Bp, int B, const float* library, float* lse_p, float* lse_s, float* a_p,
                                                              float* a_s) {
  dec_lse_combine_body(pmp, psp, pms, pss, splits, Bp, B, library, lse_p, lse_s, a_p, a_s);
}
struct LseCombArgs { const float *pmp, *psp, *pms, *pss; int splits, Bp, B; const float* library; float *lse_p, *lse_s, *a_p, *a_s; };
__global__ __launch_bounds__(256) void dec_lse_combine_pair_kernel(LseCombArgs a0, LseCombArgs a1) {
  const LseCombArgs a = blockIdx.z ? a1 : a0;
  if ((int)blockIdx.x * 64 >= a.Bp) return;
  dec_lse_combine_body(a.pmp, a.psp, a.pms, a.pss, a.splits, a.Bp, a.B, a.library, a.lse_p, a.lse_s, a.a_p, a.a_s);
}

// Counts >= NB_CMAX fall outside the (count, gene) table: the hot loop looks them up clamped to the
// last row and this out-of-line helper returns the difference to the exact value,
//   { lgamma(x+theta) - lgamma(theta) - lgamma(x+1),  psi(x+theta) - psi(theta) } - table[NB_CMAX-1][g].
// Both terms enter the result additively, so the fix-up is a plain add, made once per chunk in a
// rolled loop behind a wave-uniform test (one copy of the lgamma code, outside the hot chain).
__device__ __forceinline__ float2 gamma_terms_fixup(const DecParams& p, int g, float c) {
  const float theta = p.gene_tab[g].x, x = log1p_count(c);
  const LgammaDigamma a = lgamma_digamma(theta), b = lgamma_digamma(x + theta), d = lgamma_digamma(x + 1.0f);
  const float2 t = p.cnt_tab[(long)(int)__builtin_amdgcn_fmed3f(c, 0.f, (float)(NB_CMAX - 1)) * p.Gp + g];   // the table row the main path used for this count
  return make_float2(b.lg - a.lg - d.lg - t.x, b.dg - a.dg - t.y);
}

// ---- pass 2: NB-mixture log-likelihood, its row sums and (TRAIN) its per-element gradients ----
// The mixing logits arrive precomputed (one plain MFMA GEMM, f16 or f32 [Bp][ld]); keeping that
// K = 292 contraction out of this kernel frees ~100 registers per lane, and this kernel is bound by
// VALU/transcendental issue, which wants waves, not registers.
template <typename GT>
__device__ __forceinline__ void store4(void* base, size_t off, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<bf16_t>(void* base, size_t off, const float (&v)[4]) {
  u2v w;
  w[0] = pack2bf(v[0], v[1]);
  w[1] = pack2bf(v[2], v[3]);
  *reinterpret_cast<u2v*>(reinterpret_cast<bf16_t*>(base) + off) = w;
}
template <>
__device__ __forceinline__ void store4<float>(void* base, size_t off, const float (&v)[4]) {
  *reinterpret_cast<f4v*>(reinterpret_cast<float*>(base) + off) = f4v{v[0], v[1], v[2], v[3]};
}

template <typename LT>
__device__ __forceinline__ void load4(const void* base, size_t off, float (&v)[4]);
template <>
__device__ __forceinline__ void load4<_Float16>(const void* base, size_t off, float (&v)[4]) {
  typedef __attribute__((ext_vector_type(4))) _Float16 h4v;
  const h4v x = *reinterpret_cast<const h4v*>(reinterpret_cast<const _Float16*>(base) + off);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = (float)x[j];
}
template <>
__device__ __forceinline__ void load4<bf16_t>(const void* base, size_t off, float (&v)[4]) {
  const u2v x = *reinterpret_cast<const u2v*>(reinterpret_cast<const bf16_t*>(base) + off);
  v[0] = bf2f(x[0] & 0xFFFF); v[1] = bf2f(x[0] >> 16); v[2] = bf2f(x[1] & 0xFFFF); v[3] = bf2f(x[1] >> 16);
}
template <>
__device__ __forceinline__ void load4<float>(const void* base, size_t off, float (&v)[4]) {
  const f4v x = *reinterpret_cast<const f4v*>(reinterpret_cast<const float*>(base) + off);
#pragma unroll
  for (int j = 0; j < 4; ++j) v[j] = x[j];
}

// the same with a 32-bit BYTE offset: base (uniform, SGPRs) + zero-extended VGPR offset is an addressing mode of the
// global loads / stores, so no 64-bit address has to be computed per access
template <typename T>
__device__ __forceinline__ void store4b(void* base, unsigned elem_off, const float (&v)[4]) {
  store4<T>(reinterpret_cast<char*>(base) + (size_t)(elem_off * (unsigned)sizeof(T)), 0, v);
}
template <typename T>
__device__ __forceinline__ void load4b(const void* base, unsigned elem_off, float (&v)[4]) {
  load4<T>(reinterpret_cast<const char*>(base) + (size_t)(elem_off * (unsigned)sizeof(T)), 0, v);
}

// fp32-mode gradient arrays: a bf16 hi plane followed by a bf16 lo plane (v = hi + lo to ~2^-17), each [Bp][Gp] in
// accumulator-tile order -- exactly the two operand images the split-bf16 GEMMs of the backward pass consume, so no
// separate split pass over the fp32 values is needed (same bytes as the fp32 array they replace)
struct split_t {};
template <typename GT>
__device__ __forceinline__ void store_grad4(void* base, unsigned elem_off, unsigned plane, const float (&v)[4]) {
  if constexpr (sizeof(GT) == 2) {
    store4b<bf16_t>(base, elem_off, v);
  } else {
    unsigned h0, l0, h1, l1;
    split2bf(v[0], v[1], h0, l0);
    split2bf(v[2], v[3], h1, l1);
    const u2v wh = {h0, h1}, wl = {l0, l1};
    *reinterpret_cast<u2v*>(reinterpret_cast<char*>(base) + (size_t)(elem_off * 2u)) = wh;
    *reinterpret_cast<u2v*>(reinterpret_cast<char*>(base) + (size_t)((elem_off + plane) * 2u)) = wl;
  }
}

// sum of v[q] over the 32 lanes of a wave half for all 16 q at once ("transposing" butterfly,
// 16 shuffles instead of 80): afterwards lane l holds the total of q = 8 b4 + 4 b3 + 2 b2 + b1
// (b_i = bit i of l), duplicated in lanes l and l ^ 1.
__device__ __forceinline__ float half_sum16(const float (&v)[16], int lane) {
  float a[8];
  const bool u4 = lane & 16;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float send = u4 ? v[i] : v[i + 8], keep = u4 ? v[i + 8] : v[i];
    a[i] = keep + __shfl_xor(send, 16, 64);
  }
  float b[4];
  const bool u3 = lane & 8;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float send = u3 ? a[i] : a[i + 4], keep = u3 ? a[i + 4] : a[i];
    b[i] = keep + __shfl_xor(send, 8, 64);
  }
  float c[2];
  const bool u2 = lane & 4;
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const float send = u2 ? b[i] : b[i + 2], keep = u2 ? b[i + 2] : b[i];
    c[i] = keep + __shfl_xor(send, 4, 64);
  }
  const bool u1 = lane & 2;
  const float send = u1 ? c[0] : c[1], keep = u1 ? c[1] : c[0];
  float d = keep + __shfl_xor(send, 2, 64);
  d += __shfl_xor(d, 1, 64);
  return d;
}

// 16x16x32 MFMA formulation: a wave owns 16 cells and walks the genes of its split in chunks of 16; one chunk is
// exactly one accumulator (lane l: cell l & 15, genes 4 (l >> 4) + j), so nothing but the chunk's own 4 elements
// per lane is live -- no 32-register tile to carry, no rotation -- and the next chunk's counts / logits are
// fetched while this one is evaluated.  The [cells][genes] arrays stay in the 32x32 accumulator-tile order.
typedef __attribute__((ext_vector_type(4))) float f4acc;
constexpr float NB_LN2 = 0.6931471805599453f, NB_LOG2E = 1.4426950408889634f;
constexpr int NB_GSPL_MAX = 160;     // genes per split: their regressor weights (hi/lo) and gene table live in LDS
constexpr int NB_WPITCH = 56;        // LDS row pitch of the weight slice in bf16 (112 B: conflict-free 16-B row reads)
constexpr int NB_CELLS_PER_WG = 64;  // 4 waves x one 16-cell tile

#ifndef SPV_NB_ABLATE
#define SPV_NB_ABLATE 0   // dev only (tools/probes/nb_bench.hip): 1 no gradient stores, 2 table gather from row 0 only, 4 no logits load, 8 no d-theta reduction, 16 one count row for all
#endif
#ifdef SPV_NB_STAMPS   // dev only: s_memtime stamps of workgroup phases (tools/probes/nb_bench.hip); never defined in the library build
__device__ unsigned long long* g_nb_stamps;
#define NB_STAMP(i) do { if (threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_nb_stamps[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8 + (i)] = t_; } } while (0)
#else
#define NB_STAMP(i) do { } while (0)
#endif
#ifndef SPV_NB_OCC
#define SPV_NB_OCC 3   // waves per SIMD the likelihood kernel's register budget is sized for (tools/probes/nb_bench.hip sweeps it)
#endif
template <bool TRAIN, typename GT, typename LT, int CM>
__device__ __forceinline__ void dec_nb_body(const DecParams& p) {
  __shared__ __attribute__((aligned(16))) bf16_t s_whi[NB_GSPL_MAX * NB_WPITCH], s_wlo[NB_GSPL_MAX * NB_WPITCH];
  __shared__ float4 s_gt[NB_GSPL_MAX];
  __shared__ float s_dth[TRAIN ? 4 : 1][NB_GSPL_MAX];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int c16 = lane & 15, gq = lane >> 4;              // cell within the wave's 16, gene group within a chunk
  const int split = blockIdx.y;
  const int gbeg = split * p.nb_genes_per_split;
  int gend = gbeg + p.nb_genes_per_split;
  if (gend > p.Gp) gend = p.Gp;
  if (gend > ((p.G + 15) & ~15)) gend = (p.G + 15) & ~15;
  const int ng = gend - gbeg;
  NB_STAMP(0);
  const int nct = p.nb_cell_tiles;   // cell tiles of this workgroup: blockIdx.x * nct .. + nct - 1
  if (ng <= 0) {  // a split made of padding genes only: its partials are zeros
    for (int ct = 0; ct < nct; ++ct) {
      const int cell = (blockIdx.x * nct + ct) * NB_CELLS_PER_WG + tid;
      if (cell < p.Bp) {
        const long o = (long)split * p.Bp + cell;
        p.rec_part[o] = 0.f;
        if constexpr (TRAIN) { p.tp_part[o] = 0.f; p.ts_part[o] = 0.f; }
      }
    }
    return;
  }
  // ---- stage this split's regressor weights [ng][48] (hi, lo) and gene table in LDS -------------------------
  for (int i = tid; i < ng * 6; i += 256) {
    const int row = i / 6, ch = i % 6;
    *reinterpret_cast<u4v*>(s_whi + row * NB_WPITCH + 8 * ch) = *reinterpret_cast<const u4v*>(p.Wps_hi + (long)(gbeg + row) * DEC_KPS + 8 * ch);
    *reinterpret_cast<u4v*>(s_wlo + row * NB_WPITCH + 8 * ch) = *reinterpret_cast<const u4v*>(p.Wps_lo + (long)(gbeg + row) * DEC_KPS + 8 * ch);
  }
  for (int i = tid; i < ng; i += 256) s_gt[i] = p.gene_tab[gbeg + i];
  __syncthreads();
  NB_STAMP(1);
  const int nchunks = ng >> 4;

  // The staged weight slice serves nct consecutive 64-cell tiles (the staging + its barrier is ~15 % of a one-tile workgroup's
  // lifetime: tools/probes/nb_bench.hip, phase stamps); the per-gene d-theta sums of the tiles add up in s_dth.
  for (int ct = 0; ct < nct; ++ct) {
    const int tile16 = (blockIdx.x * nct + ct) * (NB_CELLS_PER_WG / 16) + wave;   // 16-cell tile of this wave
    if (tile16 * 16 >= p.Bp) break;   // (wave-uniform: a trailing workgroup with fewer tiles)
    const int cell = tile16 * 16 + c16;
    const int cell_tile = tile16 >> 1, chh = tile16 & 1;   // 32-cell storage tile and which half of it
    const bool cell_ok = cell < p.B;
    // resident cell-side fragments: B[k = 8 gq + i][col = cell]
    const long aoff = (long)cell * DEC_KPS + 8 * gq;
    s8v bp_hi = *reinterpret_cast<const s8v*>(p.Aps_hi + aoff), bp_lo = *reinterpret_cast<const s8v*>(p.Aps_lo + aoff);
    if (gq >= 2) {  // the private regressor only has K = 16: its K = 32 MFMA sees zeros beyond
#pragma unroll
      for (int i = 0; i < 8; ++i) { bp_hi[i] = 0; bp_lo[i] = 0; }
    }
    const s8v bs_hi = *reinterpret_cast<const s8v*>(p.Aps_hi + aoff + DEC_KP), bs_lo = *reinterpret_cast<const s8v*>(p.Aps_lo + aoff + DEC_KP);
    const float ap2 = p.a_p[cell] * NB_LOG2E, as2 = p.a_s[cell] * NB_LOG2E;   // a_k = library - lse_k, in base-2 units
    const float w = p.w_row[cell];
    const long row_of_cell = ((SPV_NB_ABLATE & 16) != 0) ? 0 : (cell_ok ? (p.rows ? (long)p.rows[cell] : (long)cell) : 0);
    float rec = 0.f, tp_sum = 0.f, ts_sum = 0.f;
    // storage offset of this lane inside a 32x32 tile for gene half gh: qq = 2 gh + (gq >> 1), h = gq & 1
    const int lane_st = (16 * chh + c16 + 32 * (gq & 1)) * 4 + (gq >> 1) * 256;
    // 32-bit element offsets (Bp * Gp < 2^31 is checked by the host): with the array base in SGPRs the loads / stores take
    // a 32-bit VGPR offset and the per-chunk 64-bit address arithmetic disappears
    const unsigned tile_row = (unsigned)cell_tile * (unsigned)p.n_gene_tiles;
    const unsigned plane = (unsigned)p.Bp * (unsigned)p.Gp;   // elements per plane of a hi/lo gradient array
    auto tile_off = [&](int g0) { return (tile_row + (unsigned)(g0 >> 5)) * 1024u + (unsigned)(((g0 >> 4) & 1) * 512 + lane_st); };
    // (count, gene) table row = the count clamped to [0, NB_CMAX - 1] with one v_med3_f32: a negative value in an fp32-stored
    // matrix (which the host rejects for resident data sets) can then never index outside the table
    auto gather_tab = [&](const float (&c)[4], int g0, float2 (&tab)[4]) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        unsigned eo = __umul24((unsigned)(int)__builtin_amdgcn_fmed3f(c[j], 0.f, (float)(NB_CMAX - 1)), (unsigned)p.Gp) + (unsigned)(g0 + 4 * gq + j);   // (Gp < 2^24)
        if constexpr ((SPV_NB_ABLATE & 2) != 0) eo = (unsigned)(g0 + 4 * gq + j);
        tab[j] = *reinterpret_cast<const float2*>(reinterpret_cast<const char*>(p.cnt_tab) + (size_t)(eo * 8u));
      }
    };

    // software pipeline: counts two chunks ahead (undecoded), logits and (count, gene) table rows one chunk ahead;
    // every load is issued before the stores of the chunk in flight, so its wait never drains those stores
    RawCounts<CM> rawA = load_counts4_raw<CM>(p, row_of_cell, gbeg + 4 * gq, cell_ok), rawB = rawA, rawC = rawA;
    float ellA[4];
    float2 tabA[4];
    load4b<LT>(p.logits, tile_off(gbeg), ellA);
    if (nchunks > 1) rawB = load_counts4_raw<CM>(p, row_of_cell, gbeg + 16 + 4 * gq, cell_ok);
    {
      float c0[4];
      decode_counts4<CM>(p, rawA, row_of_cell, gbeg + 4 * gq, cell_ok, c0);
      gather_tab(c0, gbeg, tabA);
    }

    NB_STAMP(2);
    for (int c = 0; c < nchunks; ++c) {
      if (c == 1) NB_STAMP(3);
      const int g0 = gbeg + 16 * c;
      const unsigned toff = tile_off(g0);
      // ---- y_p, y_s for this chunk: A[row = gene][k = 8 gq + i] from the LDS slice ------------------------------
      const int wrow = (16 * c + c16) * NB_WPITCH + 8 * gq;
      const s8v wp_hi = *reinterpret_cast<const s8v*>(s_whi + wrow), wp_lo = *reinterpret_cast<const s8v*>(s_wlo + wrow);
      const s8v ws_hi = *reinterpret_cast<const s8v*>(s_whi + wrow + DEC_KP), ws_lo = *reinterpret_cast<const s8v*>(s_wlo + wrow + DEC_KP);
      f4acc yp = {0.f, 0.f, 0.f, 0.f}, ys = {0.f, 0.f, 0.f, 0.f};
      yp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp_hi, bp_lo, yp, 0, 0, 0);
      yp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp_lo, bp_hi, yp, 0, 0, 0);
      yp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wp_hi, bp_hi, yp, 0, 0, 0);
      ys = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws_hi, bs_lo, ys, 0, 0, 0);
      ys = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws_lo, bs_hi, ys, 0, 0, 0);
      ys = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ws_hi, bs_hi, ys, 0, 0, 0);
      // ---- issue the loads of the following chunks (past the end: the last chunk again, unused) ------------------
      const int g1 = gbeg + 16 * min(c + 1, nchunks - 1), g2 = gbeg + 16 * min(c + 2, nchunks - 1);
      float ellB[4];
      float2 tabB[4];
      rawC = load_counts4_raw<CM>(p, row_of_cell, g2 + 4 * gq, cell_ok);
      if constexpr ((SPV_NB_ABLATE & 4) != 0) { ellB[0] = ellA[1]; ellB[1] = ellA[2]; ellB[2] = ellA[3]; ellB[3] = ellA[0]; }
      else load4b<LT>(p.logits, tile_off(g1), ellB);
      {
        float cB[4];
        decode_counts4<CM>(p, rawB, row_of_cell, g1 + 4 * gq, cell_ok, cB);
        gather_tab(cB, g1, tabB);
      }
      float cntA[4];
      decode_counts4<CM>(p, rawA, row_of_cell, g0 + 4 * gq, cell_ok, cntA);
      const int gq0 = g0 + 4 * gq;  // first of this lane's 4 consecutive genes
      float o_dl[4], o_tp[4], o_ts[4], dth4[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int g = gq0 + j;
        const bool ok = cell_ok && (g < p.G);
        const float4 gt = s_gt[g - gbeg];
        // The hardware exp / log are base 2: the kernel keeps L_k = log2(theta + mu_k + eps) and log2(mu_k + eps) in base 2
        // and folds the ln 2 into the few places that need natural units (the kernel is bound by VALU issue):
        //   nb_k = theta (lt - ln S_k) + x (ln e_k - ln S_k) = ln2 * (theta lt / ln2 + x log2 e_k - (theta + x) log2 S_k)
        const float theta = gt.x, dlt = gt.z, thlt2 = gt.w;
        const float cj = cntA[j];
        const float x = log1p_count(cj);
        const float F = tabA[j].x, Psi = tabA[j].y;
        const float el_ = ellA[j];
        const float mu1 = __builtin_amdgcn_exp2f(fmaf(yp[j], NB_LOG2E, ap2)), mu2 = __builtin_amdgcn_exp2f(fmaf(ys[j], NB_LOG2E, as2));
        const float e1 = mu1 + SPV_EPS_NB, e2 = mu2 + SPV_EPS_NB;
        const float S1 = theta + e1, S2 = theta + e2;   // theta + mu + eps (one add fewer per component than the left-to-right sum)
        const float L1 = __builtin_amdgcn_logf(S1), L2 = __builtin_amdgcn_logf(S2);   // base 2
        const float thx = theta + x;
        // nb_k and the mixing logit stay in base-2 units (nb_k / ln 2) until the one multiply that forms logp
        const float el2 = el_ * NB_LOG2E;
        const float nb1 = fmaf(x, __builtin_amdgcn_logf(e1), fmaf(-thx, L1, thlt2));
        const float v2 = fmaf(x, __builtin_amdgcn_logf(e2), fmaf(-thx, L2, thlt2 - el2));
        const float d = nb1 - v2, M = fmaxf(nb1, v2);
        const float ed = __builtin_amdgcn_exp2f(-fabsf(d)), el = __builtin_amdgcn_exp2f(-fabsf(el2));
        const float iol = fast_rcp(1.0f + el);
        const float logp = fmaf(NB_LN2, __builtin_amdgcn_logf((1.0f + ed) * iol) + (M - fmaxf(-el2, 0.f)), F);
        rec -= ok ? logp : 0.f;
        if constexpr (TRAIN) {
          // d nb_k / d y_k = mu_k (x / e_k - (theta + x) / S_k);  d nb_k / d theta = [lt + theta / (theta + eps)] - ln S_k - (theta + x) / S_k
          // (the bracket is the gene table's z); the row weight is folded into the responsibilities once
          const float iod = fast_rcp(1.0f + ed);
          const float r1 = (d >= 0.f) ? iod : ed * iod;
          const float sig = (el_ <= 0.f) ? iol : el * iol;  // sigmoid(-logit)
          const float q1 = thx * fast_rcp(S1), q2 = thx * fast_rcp(S2);
          const float g1 = mu1 * fmaf(x, fast_rcp(e1), -q1);
          const float g2 = mu2 * fmaf(x, fast_rcp(e2), -q2);
          const float dn1 = fmaf(-NB_LN2, L1, dlt) - q1;
          const float dn2 = fmaf(-NB_LN2, L2, dlt) - q2;
          const float wk = ok ? -w : 0.f;  // loss = sum_b w_b * (-sum_g logp)
          const float wr1 = wk * r1, wr2 = wk - wr1;   // w r_1, w r_2 (r_2 = 1 - r_1)
          o_dl[j] = fmaf(wk, sig, -wr2);
          o_tp[j] = wr1 * g1;
          o_ts[j] = wr2 * g2;
          tp_sum += o_tp[j];
          ts_sum += o_ts[j];
          dth4[j] = fmaf(wr1, dn1, fmaf(wr2, dn2, wk * Psi));
        }
      }
      // counts the table does not cover: beyond its last row, or (fp32-stored matrices only) not integral -- the
      // reference evaluates its lgamma terms at any float
      bool off_table = fmaxf(fmaxf(cntA[0], cntA[1]), fmaxf(cntA[2], cntA[3])) >= (float)NB_CMAX;
      if constexpr (CM == CNT_F32)
        off_table = off_table || cntA[0] != floorf(cntA[0]) || cntA[1] != floorf(cntA[1]) || cntA[2] != floorf(cntA[2]) || cntA[3] != floorf(cntA[3]);
      if (__builtin_expect(__any(off_table), 0)) {
#pragma unroll 1
        for (int j = 0; j < 4; ++j) {
          const float cj = (j == 0) ? cntA[0] : (j == 1) ? cntA[1] : (j == 2) ? cntA[2] : cntA[3];
          if ((cj >= (float)NB_CMAX || cj != floorf(cj)) && cell_ok && gq0 + j < p.G) {
            const float2 fx = gamma_terms_fixup(p, gq0 + j, cj);
            rec -= fx.x;
            if constexpr (TRAIN) {
              const float dv = -w * fx.y;
              dth4[0] += (j == 0) ? dv : 0.f; dth4[1] += (j == 1) ? dv : 0.f;
              dth4[2] += (j == 2) ? dv : 0.f; dth4[3] += (j == 3) ? dv : 0.f;
            }
          }
        }
      }
      if constexpr (TRAIN) {
        if constexpr ((SPV_NB_ABLATE & 1) != 0) {
#pragma unroll
          for (int j = 0; j < 4; ++j) asm volatile("" ::"v"(o_dl[j]), "v"(o_tp[j]), "v"(o_ts[j]));
        } else {
        store_grad4<GT>(p.dL, toff, plane, o_dl);
        store_grad4<GT>(p.tP, toff, plane, o_tp);
        store_grad4<GT>(p.tS, toff, plane, o_ts);
        }
        // per-gene sums over this wave's 16 cells (lanes sharing gq): 4 values x 16 lanes -> lane keeps gene 2*b3 + b2
        if constexpr ((SPV_NB_ABLATE & 8) != 0) { asm volatile("" ::"v"(dth4[0]), "v"(dth4[1]), "v"(dth4[2]), "v"(dth4[3])); } else {
        const bool u3 = lane & 8, u2 = lane & 4;
        const float a0 = (u3 ? dth4[2] : dth4[0]) + __shfl_xor(u3 ? dth4[0] : dth4[2], 8, 64);
        const float a1 = (u3 ? dth4[3] : dth4[1]) + __shfl_xor(u3 ? dth4[1] : dth4[3], 8, 64);
        float s = (u2 ? a1 : a0) + __shfl_xor(u2 ? a0 : a1, 4, 64);
        s += __shfl_xor(s, 2, 64);
        s += __shfl_xor(s, 1, 64);
        if ((lane & 3) == 0) {   // (each entry belongs to exactly one lane of this wave: no race across the cell tiles)
          float* slot = &s_dth[wave][16 * c + 4 * gq + 2 * ((lane >> 3) & 1) + ((lane >> 2) & 1)];
          *slot = (ct == 0) ? s : *slot + s;
        }
        }
      }

#pragma unroll
      for (int j = 0; j < 4; ++j) { ellA[j] = ellB[j]; tabA[j] = tabB[j]; }
      rawA = rawB; rawB = rawC;
    }
    NB_STAMP(4);
    // the four gene groups of a cell sit in lanes c16, c16+16, c16+32, c16+48
    rec += __shfl_xor(rec, 16, 64); rec += __shfl_xor(rec, 32, 64);
    if constexpr (TRAIN) {
      tp_sum += __shfl_xor(tp_sum, 16, 64); tp_sum += __shfl_xor(tp_sum, 32, 64);
      ts_sum += __shfl_xor(ts_sum, 16, 64); ts_sum += __shfl_xor(ts_sum, 32, 64);
    }
    if (gq == 0) {
      const long o = (long)split * p.Bp + cell;
      p.rec_part[o] = rec;
      if constexpr (TRAIN) { p.tp_part[o] = tp_sum; p.ts_part[o] = ts_sum; }
    }
  }
  if constexpr (TRAIN) {  // per-gene d theta over the workgroup's 64 cells: the four waves' partials in wave order
    __syncthreads();
    for (int i = tid; i < ng; i += 256)
      p.dtheta_part[(long)blockIdx.x * p.Gp + gbeg + i] = ((s_dth[0][i] + s_dth[1][i]) + s_dth[2][i]) + s_dth[3][i];
  }
  NB_STAMP(5);
}
template <bool TRAIN, typename GT, typename LT, int CM>
__global__ __launch_bounds__(256, SPV_NB_OCC) void dec_nb_kernel(DecParams p) { dec_nb_body<TRAIN, GT, LT, CM>(p); }
template <bool TRAIN, typename GT, typename LT, int CM>
__global__ __launch_bounds__(256, SPV_NB_OCC) void dec_nb_pair_kernel(DecParams p0, DecParams p1) {
  const DecParams p = blockIdx.z ? p1 : p0;
  const int ctiles = (p.Bp + NB_CELLS_PER_WG - 1) / NB_CELLS_PER_WG;
  if ((int)blockIdx.x >= (ctiles + p.nb_cell_tiles - 1) / p.nb_cell_tiles || (int)blockIdx.y >= p.nb_splits) return;   // (workgroup-uniform)
  dec_nb_body<TRAIN, GT, LT, CM>(p);
}

// ---- materialising path: the decoder outputs the reference's generative() returns ---------------------------------
//   px_scale_k = softmax_G(y_k),  px_rate_k = exp(library) * px_scale_k   (nn/networks.py:314-320),  mixing logits
// (:322-325) as plain row-major fp32 [B][ld].  Off the hot path (the fused likelihood never stores them): same y_k
// evaluation as dec_lse_kernel (split-bf16 K = 16 / 32 MFMAs), needs lse_k / a_k from spv_dec_lse and the tile-ordered
// logits from spv_dec_logits.  Lane = cell, registers = genes: every lane writes 16-byte pieces of its own row.
template <typename LT>
__global__ __launch_bounds__(256) void dec_materialize_kernel(DecParams p, float* scale_p, float* scale_s, float* rate_p, float* rate_s,
                                                              float* logits_out, long ld) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
  const int cell_tile = blockIdx.x * (DEC_CELLS_PER_WG / 32) + wave;
  const int cell0 = cell_tile * 32, cell = cell0 + r;
  PsFrags cf;
  load_ps_cell_frags(p, cell0, lane, cf);
  const float lp = p.lse_p[cell], ls = p.lse_s[cell], ap = p.a_p[cell], as = p.a_s[cell];
  const int gbeg = blockIdx.y * p.genes_per_split;
  int gend = gbeg + p.genes_per_split;
  if (gend > p.Gp) gend = p.Gp;
  if (gend > ((p.G + 31) & ~31)) gend = (p.G + 31) & ~31;
  for (int g0 = gbeg; g0 < gend; g0 += 32) {
    PsW w;
    load_ps_w(p, g0, lane, w);
    f16v yp, ys;
#pragma unroll
    for (int q = 0; q < 16; ++q) { yp[q] = 0.f; ys[q] = 0.f; }
    yp = mfma32_split<3>(w.hi[0], w.lo[0], cf.hi[0], cf.lo[0], yp);
    ys = mfma32_split<3>(w.hi[1], w.lo[1], cf.hi[1], cf.lo[1], ys);
    ys = mfma32_split<3>(w.hi[2], w.lo[2], cf.hi[2], cf.lo[2], ys);
    const size_t tbase = ((size_t)cell_tile * p.n_gene_tiles + (g0 >> 5)) * 1024 + lane * 4;
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      float el[4];
      load4<LT>(p.logits, tbase + 256 * qq, el);
      const int g = g0 + 8 * qq + 4 * h;
      if (cell >= p.B) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        if (g + j >= p.G) continue;
        const int q = 4 * qq + j;
        const size_t o = (size_t)cell * ld + g + j;
        scale_p[o] = __expf(yp[q] - lp); scale_s[o] = __expf(ys[q] - ls);
        rate_p[o] = __expf(yp[q] + ap);  rate_s[o] = __expf(ys[q] + as);
        logits_out[o] = el[j];
      }
    }
  }
}

// ---- backward helper: finish the softmax backward in place -------------------------------------
//   d/dy_k[b,g] = t_k[b,g] - softmax_k[b,g] * T_k[b],   T_k[b] = sum_g t_k[b,g]
// (t_k came out of dec_nb_kernel with a_k = library - lse_k held fixed; the second term is the
// derivative through lse_k).  Re-evaluates y_k with the K = 16/32 split MFMAs and one exp each.
template <typename GT> struct Raw4;
template <> struct Raw4<bf16_t> { typedef u2v type; };
template <> struct Raw4<split_t> { typedef u4v type; };   // hi words, lo words
template <typename GT>
__device__ __forceinline__ typename Raw4<GT>::type load4_raw(const void* base, long off, long plane) {
  if constexpr (sizeof(typename Raw4<GT>::type) == 8) {
    return *reinterpret_cast<const u2v*>(reinterpret_cast<const bf16_t*>(base) + off);
  } else {
    const u2v h = *reinterpret_cast<const u2v*>(reinterpret_cast<const bf16_t*>(base) + off);
    const u2v l = *reinterpret_cast<const u2v*>(reinterpret_cast<const bf16_t*>(base) + off + plane);
    return u4v{h[0], h[1], l[0], l[1]};
  }
}
template <typename GT>
__device__ __forceinline__ void store4_grad(void* base, long off, long plane, const float (&v)[4]) {
  if constexpr (sizeof(typename Raw4<GT>::type) == 8) {
    store4<bf16_t>(base, (size_t)off, v);
  } else {
    unsigned h0, l0, h1, l1;
    split2bf(v[0], v[1], h0, l0);
    split2bf(v[2], v[3], h1, l1);
    const u2v wh = {h0, h1}, wl = {l0, l1};
    *reinterpret_cast<u2v*>(reinterpret_cast<bf16_t*>(base) + off) = wh;
    *reinterpret_cast<u2v*>(reinterpret_cast<bf16_t*>(base) + off + plane) = wl;
  }
}
__device__ __forceinline__ void decode4(const u2v& x, float (&v)[4]) {
  v[0] = bf2f(x[0] & 0xFFFF); v[1] = bf2f(x[0] >> 16); v[2] = bf2f(x[1] & 0xFFFF); v[3] = bf2f(x[1] >> 16);
}
__device__ __forceinline__ void decode4(const u4v& x, float (&v)[4]) {   // hi + lo
  v[0] = bf2f(x[0] & 0xFFFF) + bf2f(x[2] & 0xFFFF); v[1] = bf2f(x[0] >> 16) + bf2f(x[2] >> 16);
  v[2] = bf2f(x[1] & 0xFFFF) + bf2f(x[3] & 0xFFFF); v[3] = bf2f(x[1] >> 16) + bf2f(x[3] >> 16);
}

// Software pipelined over 32-gene tiles: the next tile's fragments and gradient words are requested before the
// current tile's stores are issued, so waiting for them never has to drain those stores (vmcnt is in order).
//
// FUSE (bf16 gradients; split-bf16 "fp32" gradient words: the same contraction on hi / lo pairs, three MFMAs): the corrected tile sits in registers in MFMA accumulator layout [gene][cell]; read as the
// B operand of two more 32x32x16 MFMAs per head (k-slots = the lane's 2 x 4 consecutive genes, the A operand = the
// transposed regressor slice W'^T[k][gene] staged in LDS with the same slot order) it yields the gradient reaching the
// latents through the two rate heads,  dz_p[cell][k] = sum_gene t'_P[gene][cell] W'_p[gene][k]  (and dz_s), accumulated
// over the split's genes and written as one partial slab per split -- the two [B,G] x [G,16|32] GEMMs of the backward
// pass and their re-read of t'_P / t'_S disappear.
constexpr int SMB_SUB = 160;            // genes of the transposed slice resident in LDS at a time (320 before the weight-gradient fusion took its share of LDS: 2 workgroups per CU must fit)
constexpr int SMB_PITCH = SMB_SUB + 4;  // bf16 per row: 648 B = 162 dwords = 2 x 81: the 32 rows an 8-byte fragment read touches start on 32 distinct even banks
                                        // (the first choice, 656 B = 4 x 41 dwords, put rows r and r + 16 on the same banks: SQ_LDS_BANK_CONFLICT 3.7 M per launch)

// WRITE = false: read-only variant (FUSE only): the latent gradient alone, t_P / t_S stay uncorrected.  The backward pass's
// critical chain (latent gradient -> trunk / PoE / encoder backward) then waits for a pass that only READS the two gradient
// arrays; the in-place correction the regressor weight-gradient GEMMs need runs beside that chain on the side stream.
//
// HEADS (FUSE, read-only): the same corrected tile also yields the two regressor weight gradients,
//   d W'_p[gene][k] = sum_cell t'_P[gene][cell] z_p[cell][k]   (and d W'_s),   z = the latent operand image [cell][48] ( | 1 column = d c),
// a contraction over CELLS: every wave parks its 32-cell x 32-gene tile (bf16, cells on rows) in LDS, one barrier, and two of the four
// waves (alternating by tile parity) run the 128-cell contraction of one head each with transposed LDS reads, the latent image of the
// workgroup's cells resident in registers as the B operand.  One partial [G][16] / [G][32] slab per 128-cell workgroup row, summed in
// order by spv_reduce_slabs.  With it nothing re-reads t_P / t_S after this pass: no write-back (168 MB per group) and no
// spv_dec_heads_wgrad pass (another 168 MB read) -- the [B, G] traffic of the decoder backward drops from 924 to ~600 MB per group.
constexpr int SMB_ZPITCH = 96;   // bf16 per row of the latent image in LDS: 48 dwords, a k-major pitch whose transposed reads are conflict free (kmajor_pitch)
constexpr int SMB_TILE_ELEMS = DEC_CELLS_PER_WG * 32;   // one head's parked tile: [128 cells][32 genes] bf16 (k-major for the transposed reads)
template <typename GT, bool FUSE, bool WRITE = true, bool HEADS = false>
__global__ __launch_bounds__(256, 2) void dec_softmax_bwd_kernel(DecParams p, const float* Tp, const float* Ts, float* dz_part, float* dwp_part, float* dws_part) {
  typedef typename Raw4<GT>::type raw_t;
  static_assert(!HEADS || (FUSE && !WRITE), "the weight-gradient fusion rides on the read-only fused pass");
  const long plane = (long)p.Bp * p.Gp;
  constexpr bool SPLITG = FUSE && sizeof(raw_t) == sizeof(u4v);   // split-bf16 gradient words ("fp32" mode): the fused contraction runs on hi / lo pairs
  __shared__ __attribute__((aligned(16))) bf16_t s_wT[FUSE ? DEC_KPS * SMB_PITCH : 8];
  __shared__ __attribute__((aligned(16))) bf16_t s_wT_lo[SPLITG ? DEC_KPS * SMB_PITCH : 8];
  __shared__ __attribute__((aligned(16))) bf16_t s_tile[HEADS ? 2 * 2 * SMB_TILE_ELEMS : 8];   // [tile parity][head][cell][gene]
  __shared__ __attribute__((aligned(16))) bf16_t s_z[HEADS ? DEC_CELLS_PER_WG * SMB_ZPITCH : 8];  // the workgroup's rows of the latent image [cell][48 | pad]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
  const int cell_tile = blockIdx.x * (DEC_CELLS_PER_WG / 32) + wave;
  const int cell0 = cell_tile * 32;
  const int cell = cell0 + r;
  const int split = blockIdx.y;
  PsFrags cf;
  load_ps_cell_frags(p, cell0, lane, cf);
  const float lp = p.lse_p[cell], ls = p.lse_s[cell];
  const float tpb = Tp[cell], tsb = Ts[cell];
  const int gbeg = split * p.genes_per_split;
  int gend = gbeg + p.genes_per_split;
  if (gend > p.Gp) gend = p.Gp;
  if (gend > ((p.G + 31) & ~31)) gend = (p.G + 31) & ~31;
  const int ntile = (gend - gbeg) >> 5;
  f16v accP, accS;
#pragma unroll
  for (int q = 0; q < 16; ++q) { accP[q] = 0.f; accS[q] = 0.f; }
  // HEADS: this wave's role in the weight-gradient contraction (waves 0 / 2: private head on even / odd tiles, waves 1 / 3: shared head); its
  // B operand z[cell][column of the head] comes out of the LDS copy of the workgroup's 128 latent-image rows by transposed reads
  const int head_role = wave & 1;
  if constexpr (HEADS) {
    const int cb0 = blockIdx.x * DEC_CELLS_PER_WG;
    for (int i = threadIdx.x; i < DEC_CELLS_PER_WG * (DEC_KPS / 8); i += 256) {   // (Bp is a multiple of 128: in bounds; rows >= B hold zeros)
      const int row = i / (DEC_KPS / 8), c8 = (i % (DEC_KPS / 8)) * 8;
      *reinterpret_cast<u4v*>(s_z + row * SMB_ZPITCH + c8) = *reinterpret_cast<const u4v*>(p.Aps_hi + (long)(cb0 + row) * DEC_KPS + c8);
    }
    // (made visible by the first barrier of the tile loop, which every wave passes before the first contraction)
  }
  if (ntile > 0) {   // (block-uniform)
    const long trow = (long)cell_tile * p.n_gene_tiles;
    PsW wA;
    // two register stages for the gradient words, used alternately (the loop below is unrolled by two, so the roles are static):
    // read-modify-write variants request tile t + 1 at the top of tile t; the read-only HEADS variant has nothing to drain and keeps
    // TWO tiles in flight -- tile t + 2 is requested into tile t's registers as soon as they have been decoded (with one barrier per
    // tile the four waves cannot drift apart, and a single tile of prefetch left the pass latency bound at 2.1 TB/s)
    raw_t rp0[4], rs0[4], rp1[4], rs1[4];
    load_ps_w(p, gbeg, lane, wA);
    // HEADS (bf16 words): 16-byte loads -- a lane pair (2i, 2i + 1) fetches the pair's 16 bytes of word group qq = 2 pr (even lane) and
    // 2 pr + 1 (odd lane) and the halves are swapped between the two lanes when the tile is decoded (`unswap`): half the load
    // instructions, each 1 KiB per wave instead of 512 B (8-byte accesses run at 0.54-0.70 of the 16-byte rate: MI355X_MICROARCH.md)
    auto request = [&](int t, raw_t (&rp)[4], raw_t (&rs)[4]) {
      const long tb = (trow + ((gbeg + 32 * min(t, ntile - 1)) >> 5)) * 1024 + lane * 4;
      if constexpr (HEADS) {
        const long tb2 = tb - (lane & 1) * 4 + (lane & 1) * 256;   // the pair's first word, in group 2 pr + (lane & 1)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const u4v a = *reinterpret_cast<const u4v*>(reinterpret_cast<const bf16_t*>(p.tP) + tb2 + 512 * pr);
          const u4v b = *reinterpret_cast<const u4v*>(reinterpret_cast<const bf16_t*>(p.tS) + tb2 + 512 * pr);
          rp[2 * pr] = u2v{a[0], a[1]}; rp[2 * pr + 1] = u2v{a[2], a[3]};   // (still swapped: see unswap)
          rs[2 * pr] = u2v{b[0], b[1]}; rs[2 * pr + 1] = u2v{b[2], b[3]};
        }
      } else {
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) { rp[qq] = load4_raw<GT>(p.tP, tb + 256 * qq, plane); rs[qq] = load4_raw<GT>(p.tS, tb + 256 * qq, plane); }
      }
    };
    // after a HEADS request: r[2 pr] = the pair's EVEN-lane words and r[2 pr + 1] its ODD-lane words, of group 2 pr on even lanes and of
    // group 2 pr + 1 on odd lanes; each lane keeps its own words of its group and trades the other half with its neighbour
    auto unswap = [&](raw_t (&r4)[4]) {
      if constexpr (HEADS) {
        const bool odd = lane & 1;
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) {
          const u2v lo = r4[2 * pr], hi = r4[2 * pr + 1];
          const u2v send = odd ? lo : hi;   // the neighbour's words
          const u2v recv = u2v{(unsigned)__shfl_xor((int)send[0], 1, 64), (unsigned)__shfl_xor((int)send[1], 1, 64)};
          r4[2 * pr] = odd ? recv : lo;       // own words of group 2 pr
          r4[2 * pr + 1] = odd ? hi : recv;   // own words of group 2 pr + 1
        }
      }
    };
    request(0, rp0, rs0);
    if constexpr (HEADS) request(1, rp1, rs1);
    auto tile = [&](const int t, raw_t (&rpA)[4], raw_t (&rsA)[4], raw_t (&rpB)[4], raw_t (&rsB)[4]) {
      const int g0 = gbeg + 32 * t, gn = gbeg + 32 * min(t + 1, ntile - 1);
      if constexpr (FUSE) {
        if (t % (SMB_SUB / 32) == 0) {  // (re)stage W'^T for the next SMB_SUB genes: s_wT[k][gene - g0]
          lds_barrier();
          const int ng = min(SMB_SUB, gend - g0);
          for (int i = threadIdx.x; i < ng * (DEC_KPS / 8); i += 256) {
            const int gl = i / (DEC_KPS / 8), c8 = (i % (DEC_KPS / 8)) * 8;
            const s8v w = *reinterpret_cast<const s8v*>(p.Wps_hi + (long)(g0 + gl) * DEC_KPS + c8);
#pragma unroll
            for (int j = 0; j < 8; ++j) s_wT[(c8 + j) * SMB_PITCH + gl] = (bf16_t)w[j];
            if constexpr (SPLITG) {
              const s8v wl = *reinterpret_cast<const s8v*>(p.Wps_lo + (long)(g0 + gl) * DEC_KPS + c8);
#pragma unroll
              for (int j = 0; j < 8; ++j) s_wT_lo[(c8 + j) * SMB_PITCH + gl] = (bf16_t)wl[j];
            }
          }
          lds_barrier();
        }
      }
      f16v yp, ys;
#pragma unroll
      for (int q = 0; q < 16; ++q) { yp[q] = 0.f; ys[q] = 0.f; }
      yp = mfma32_split<3>(wA.hi[0], wA.lo[0], cf.hi[0], cf.lo[0], yp);
      ys = mfma32_split<3>(wA.hi[1], wA.lo[1], cf.hi[1], cf.lo[1], ys);
      ys = mfma32_split<3>(wA.hi[2], wA.lo[2], cf.hi[2], cf.lo[2], ys);
      PsW wB;
      load_ps_w(p, gn, lane, wB);
      const long tbase = (trow + (g0 >> 5)) * 1024 + lane * 4;
      if constexpr (!HEADS) request(t + 1, rpB, rsB);
      // corrected values as packed bf16 pairs straight away (register [qq] = genes 8 qq + 4 h + {0,1 | 2,3}): what the MFMA operands
      // and the parked tiles hold, and half the registers of 2 x 16 floats
      unsigned cpk[8], csk[8];
      unsigned cpl[SPLITG ? 8 : 1], csl[SPLITG ? 8 : 1];   // ... and, for split words, the bf16 of what the first rounding left
      unswap(rpA);
      unswap(rsA);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int g = g0 + 8 * qq + 4 * h;
        float vp[4], vs[4];
        decode4(rpA[qq], vp);
        decode4(rsA[qq], vs);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = 4 * qq + j;
          const bool ok = (g + j < p.G) && (cell < p.B);
          vp[j] = ok ? vp[j] - fast_exp(yp[q] - lp) * tpb : 0.f;
          vs[j] = ok ? vs[j] - fast_exp(ys[q] - ls) * tsb : 0.f;
        }
        cpk[2 * qq] = pack2bf(vp[0], vp[1]); cpk[2 * qq + 1] = pack2bf(vp[2], vp[3]);
        csk[2 * qq] = pack2bf(vs[0], vs[1]); csk[2 * qq + 1] = pack2bf(vs[2], vs[3]);
        if constexpr (SPLITG) {
          auto lo2 = [](unsigned hi2, float a, float b) { return pack2bf(a - __uint_as_float(hi2 << 16), b - __uint_as_float(hi2 & 0xffff0000u)); };
          cpl[2 * qq] = lo2(cpk[2 * qq], vp[0], vp[1]); cpl[2 * qq + 1] = lo2(cpk[2 * qq + 1], vp[2], vp[3]);
          csl[2 * qq] = lo2(csk[2 * qq], vs[0], vs[1]); csl[2 * qq + 1] = lo2(csk[2 * qq + 1], vs[2], vs[3]);
        }
        if constexpr (WRITE) {
          store4_grad<GT>(p.tP, tbase + 256 * qq, plane, vp);
          store4_grad<GT>(p.tS, tbase + 256 * qq, plane, vs);
        }
      }
      if constexpr (HEADS) request(t + 2, rpA, rsA);   // this tile's registers are free again: the tile after next goes into them
      if constexpr (FUSE) {
        const int gl = (g0 - gbeg) % SMB_SUB;   // offset of this tile inside the staged slice
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          // B operand: this lane's genes 16m + 4h + {0..3} and 16m + 8 + 4h + {0..3} (registers 8m .. 8m + 7), as stored (bf16)
          const u4v bPw = u4v{cpk[4 * m], cpk[4 * m + 1], cpk[4 * m + 2], cpk[4 * m + 3]}, bSw = u4v{csk[4 * m], csk[4 * m + 1], csk[4 * m + 2], csk[4 * m + 3]};
          const s8v bP = *reinterpret_cast<const s8v*>(&bPw), bS = *reinterpret_cast<const s8v*>(&bSw);
          // A operand: row k = lane & 31 of W'^T at the same genes
          const bf16_t* wr = s_wT + gl + 16 * m + 4 * h;
          const u2v p0 = *reinterpret_cast<const u2v*>(wr + (r & 15) * SMB_PITCH), p1 = *reinterpret_cast<const u2v*>(wr + (r & 15) * SMB_PITCH + 8);
          const u2v s0 = *reinterpret_cast<const u2v*>(wr + (DEC_KP + r) * SMB_PITCH), s1 = *reinterpret_cast<const u2v*>(wr + (DEC_KP + r) * SMB_PITCH + 8);
          u4v ap = u4v{p0[0], p0[1], p1[0], p1[1]};
          if (r >= DEC_KP) ap = u4v{0u, 0u, 0u, 0u};   // the private head has 16 rows
          const u4v as4 = u4v{s0[0], s0[1], s1[0], s1[1]};
          if constexpr (SPLITG) {   // hi * lo, lo * hi, hi * hi (small terms first, as spv_gemm.h's mfma32_split)
            const u4v bPl = u4v{cpl[4 * m], cpl[4 * m + 1], cpl[4 * m + 2], cpl[4 * m + 3]}, bSl = u4v{csl[4 * m], csl[4 * m + 1], csl[4 * m + 2], csl[4 * m + 3]};
            const bf16_t* wl = s_wT_lo + gl + 16 * m + 4 * h;
            const u2v lp0 = *reinterpret_cast<const u2v*>(wl + (r & 15) * SMB_PITCH), lp1 = *reinterpret_cast<const u2v*>(wl + (r & 15) * SMB_PITCH + 8);
            const u2v ls0 = *reinterpret_cast<const u2v*>(wl + (DEC_KP + r) * SMB_PITCH), ls1 = *reinterpret_cast<const u2v*>(wl + (DEC_KP + r) * SMB_PITCH + 8);
            u4v apl = u4v{lp0[0], lp0[1], lp1[0], lp1[1]};
            if (r >= DEC_KP) apl = u4v{0u, 0u, 0u, 0u};
            const u4v asl = u4v{ls0[0], ls0[1], ls1[0], ls1[1]};
            accP = mfma32(*reinterpret_cast<const s8v*>(&ap), *reinterpret_cast<const s8v*>(&bPl), accP);
            accP = mfma32(*reinterpret_cast<const s8v*>(&apl), bP, accP);
            accS = mfma32(*reinterpret_cast<const s8v*>(&as4), *reinterpret_cast<const s8v*>(&bSl), accS);
            accS = mfma32(*reinterpret_cast<const s8v*>(&asl), bS, accS);
          }
          accP = mfma32(*reinterpret_cast<const s8v*>(&ap), bP, accP);
          accS = mfma32(*reinterpret_cast<const s8v*>(&as4), bS, accS);
          if constexpr (HEADS) {   // park the tile: row = this wave's cell, the lane's two 4-gene chunks of this half
            // 64-byte rows: a row's eight 8-byte chunks are stored XOR-swizzled with (row >> 1) & 7, so that the 16 lanes of a store
            // (16 consecutive rows, one chunk each) and the 4 rows x 4 chunks of a transposed read both land on distinct banks
            // (unswizzled: SQ_LDS_BANK_CONFLICT 10.0 M cycles per launch, 4-way on every store)
            const int row = wave * 32 + r, sw = (row >> 1) & 7;
            bf16_t* tp_ = s_tile + ((t & 1) * 2 + 0) * SMB_TILE_ELEMS + row * 32;
            bf16_t* ts_ = tp_ + SMB_TILE_ELEMS;
            const int ca = 4 * (((4 * m + h)) ^ sw), cb = 4 * (((4 * m + 2 + h)) ^ sw);
            *reinterpret_cast<u2v*>(tp_ + ca) = u2v{bPw[0], bPw[1]};
            *reinterpret_cast<u2v*>(tp_ + cb) = u2v{bPw[2], bPw[3]};
            *reinterpret_cast<u2v*>(ts_ + ca) = u2v{bSw[0], bSw[1]};
            *reinterpret_cast<u2v*>(ts_ + cb) = u2v{bSw[2], bSw[3]};
          }
        }
      }
      if constexpr (HEADS) {
        // all four waves' tiles of this gene tile are in LDS after the barrier; the buffer of parity t & 1 is next written two tiles
        // later, behind the NEXT barrier, by which time its readers (before that barrier in program order) are done: one barrier per tile
        lds_barrier();
        if ((wave >> 1) == (t & 1)) {   // (wave-uniform) this tile's two working waves
          const bf16_t* img = s_tile + ((t & 1) * 2 + head_role) * SMB_TILE_ELEMS;
          f16v accW;
#pragma unroll
          for (int q = 0; q < 16; ++q) accW[q] = 0.f;
#pragma unroll
          for (int ks = 0; ks < 8; ++ks) {  // A[gene][cell] and B[cell][column], k = cells: both by transposed reads of k-major images
            // A: the parked tile, chunk-swizzled as stored (frag_kmajor's addressing with the XOR applied per row)
            const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3;
            const int row0 = 16 * ks + 8 * h + q4, row1 = row0 + 4, chunk = 4 * (gi & 1) + p4;
            const s4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(img + row0 * 32 + 4 * (chunk ^ ((row0 >> 1) & 7))));
            const s4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(img + row1 * 32 + 4 * (chunk ^ ((row1 >> 1) & 7))));
            const s8v af = s8v{a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            accW = mfma32(af, frag_kmajor(s_z, SMB_ZPITCH, head_role ? DEC_KP : 0, 16 * ks, lane), accW);
          }
          // accW[q]: row = gene g0 + crow(q, h), column = r of this head
          const int nk = head_role ? DEC_KS : DEC_KP;
          float* out = (head_role ? dws_part : dwp_part) + (long)blockIdx.x * p.G * nk;
          if (r < nk) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
              const int g = g0 + crow(q, h);
              if (g < p.G) out[(long)g * nk + r] = accW[q];
            }
          }
        }
      }
      wA = wB;
    };
    for (int t = 0; t < ntile; t += 2) {   // (block-uniform trip structure: the barriers inside `tile` are reached by every wave)
      tile(t, rp0, rs0, rp1, rs1);
      if (t + 1 < ntile) tile(t + 1, rp1, rs1, rp0, rs0);
    }
  }
  if constexpr (FUSE) {   // acc[q]: row = crow(q, h) = k, column = lane & 31 = cell; one slab per split, zeros for empty splits
    float* out = dz_part + ((long)split * p.Bp + cell) * DEC_KPS;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int k = crow(q, h);
      if (k < DEC_KP) out[k] = accP[q];
      out[DEC_KP + k] = accS[q];
    }
  }
}

// ---- one-pass decoder backward, second form (round 3): the same arithmetic as dec_softmax_bwd_kernel<bf16_t, true, false, true> with an
// instruction stream that is IDENTICAL for every wave and every tile, so that hipcc's s_waitcnt counts are exact.
//
// What was wrong with the first form (found in its disassembly): the weight-gradient contraction ran on two of the four waves per tile
// (wave-uniform branch), and those waves stored their [32 genes x 16 | 32] block with 16 global stores inside that branch.  vmcnt retires
// in order and counts stores; at a control-flow join the compiler has to assume the path WITHOUT the stores, so the wait for the next
// tile's weight fragments became vmcnt(4) -- which, on a wave that had just stored, also waits for its two tiles of prefetched gradient
// words and for 12 of the 16 stores.  Every tile started with a full HBM round trip (2.3 TB/s).  The periodic re-staging of W'^T had a
// vmcnt(0) of its own every five tiles.
//
// Here: (1) every wave takes part in every tile's contraction -- wave w owns head w & 1 and gene half w >> 1 and computes its
// workgroup's 128 cells with v_mfma_f32_16x16x32_bf16: the tile's six [16 genes x 16 columns] blocks (private | shared low | shared high
// columns, two gene halves) are dealt two per wave (8 MFMAs and 8 stores each; two waves repeat a block, bit for bit, so that nobody's
// stream differs); (2) ONE partial slab [Bp / 128][Gp][48] = [d W'_p | d W'_s], stored unconditionally (padding gene rows exist);
// (3) W'^T is staged one 32-gene tile at a time, double buffered, its 16-byte global loads issued at the top of the previous tile by
// 192 threads: one load per tile, no drain.  Per tile and wave the vector-memory stream is: 6 fragment loads, 1 staging load, 4 tile
// loads (two tiles ahead), 8 stores -- in that order, always.
constexpr int HB_WT_PITCH = 36;   // bf16 per row of a staged W'^T tile [48][32 | 4 pad]: 18 dwords, 32 rows on 32 distinct even banks
__device__ __forceinline__ void dec_heads_bwd_body(const DecParams& p, const float* Tp, const float* Ts, float* dz_part, float* dw_part) {
  __shared__ __attribute__((aligned(16))) bf16_t s_wT[2 * DEC_KPS * HB_WT_PITCH];                 // [tile parity][k][gene of the tile]
  __shared__ __attribute__((aligned(16))) bf16_t s_tile[2 * 2 * SMB_TILE_ELEMS];                  // [tile parity][head][cell][gene], chunk-swizzled
  __shared__ __attribute__((aligned(16))) bf16_t s_zh[2 * DEC_CELLS_PER_WG * 32];                 // [head][cell][32 columns] of the latent image (private: 16 + zeros)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
  const int cell_tile = blockIdx.x * (DEC_CELLS_PER_WG / 32) + wave;
  const int cell0 = cell_tile * 32;
  const int cell = cell0 + r;
  const int split = blockIdx.y;
  PsFrags cf;
  load_ps_cell_frags(p, cell0, lane, cf);
  const float lp = p.lse_p[cell], ls = p.lse_s[cell];
  const float tpb = Tp[cell], tsb = Ts[cell];
  const int gbeg = split * p.genes_per_split;
  int gend = gbeg + p.genes_per_split;
  if (gend > p.Gp) gend = p.Gp;
  if (gend > ((p.G + 31) & ~31)) gend = (p.G + 31) & ~31;
  const int ntile = (gend - gbeg) >> 5;
  f16v accP, accS;
#pragma unroll
  for (int q = 0; q < 16; ++q) { accP[q] = 0.f; accS[q] = 0.f; }
  if (ntile > 0) {   // (block-uniform)
    // ---- the workgroup's 128 rows of the latent image, one 32-column image per head ----------------------------------------------
    {
      const int cb0 = blockIdx.x * DEC_CELLS_PER_WG;
      for (int i = threadIdx.x; i < 2 * DEC_CELLS_PER_WG * 4; i += 256) {   // 8-element chunks: 4 per row and head
        const int hd = i / (DEC_CELLS_PER_WG * 4), row = (i / 4) % DEC_CELLS_PER_WG, c8 = (i % 4) * 8;
        u4v v = u4v{0u, 0u, 0u, 0u};
        if (hd == 1 || c8 < DEC_KP) v = *reinterpret_cast<const u4v*>(p.Aps_hi + (long)(cb0 + row) * DEC_KPS + (hd ? DEC_KP : 0) + c8);
        *reinterpret_cast<u4v*>(s_zh + (hd * DEC_CELLS_PER_WG + row) * 32 + c8) = v;
      }
    }
    // ---- W'^T staging: thread i < 192 owns gene i / 6 of a tile and the 8 columns 8 (i % 6) .. ------------------------------------
    // 192 pieces (32 genes x 6 chunks of 8 columns) for 256 threads: threads 192.. repeat the pieces 0..63 (same value to the same LDS
    // word: harmless).  NO branch around either half: behind an `if (thread < 192)` hipcc sinks the load into the branch, next to the
    // LDS stores, and drains the whole queue for it (s_waitcnt vmcnt(0)) once per tile
    const int st_i = (threadIdx.x < 192) ? (int)threadIdx.x : (int)threadIdx.x - 192;
    const int st_g = st_i / (DEC_KPS / 8), st_c = (st_i % (DEC_KPS / 8)) * 8;
    auto wt_load = [&](int t) -> u4v {
      const int g0 = gbeg + 32 * min(t, ntile - 1);
      return *reinterpret_cast<const u4v*>(p.Wps_hi + (long)(g0 + st_g) * DEC_KPS + st_c);
    };
    auto wt_store = [&](int t, const u4v& w) {
      bf16_t* dst = s_wT + (t & 1) * DEC_KPS * HB_WT_PITCH + st_c * HB_WT_PITCH + st_g;
#pragma unroll
      for (int j = 0; j < 4; ++j) { dst[(2 * j) * HB_WT_PITCH] = (bf16_t)(w[j] & 0xFFFFu); dst[(2 * j + 1) * HB_WT_PITCH] = (bf16_t)(w[j] >> 16); }
    };
    const long trow = (long)cell_tile * p.n_gene_tiles;
    PsW wA;
    u2v rp0[4], rs0[4], rp1[4], rs1[4];
    load_ps_w(p, gbeg, lane, wA);
    auto request = [&](int t, u2v (&rp)[4], u2v (&rs)[4]) {   // 16-byte loads of the lane pair's words, unswapped at decode (see the first form)
      const long tb = (trow + ((gbeg + 32 * min(t, ntile - 1)) >> 5)) * 1024 + lane * 4;
      const long tb2 = tb - (lane & 1) * 4 + (lane & 1) * 256;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const u4v a = *reinterpret_cast<const u4v*>(reinterpret_cast<const bf16_t*>(p.tP) + tb2 + 512 * pr);
        const u4v b = *reinterpret_cast<const u4v*>(reinterpret_cast<const bf16_t*>(p.tS) + tb2 + 512 * pr);
        rp[2 * pr] = u2v{a[0], a[1]}; rp[2 * pr + 1] = u2v{a[2], a[3]};
        rs[2 * pr] = u2v{b[0], b[1]}; rs[2 * pr + 1] = u2v{b[2], b[3]};
      }
    };
    auto unswap = [&](u2v (&r4)[4]) {
      const bool odd = lane & 1;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const u2v lo = r4[2 * pr], hi = r4[2 * pr + 1];
        const u2v send = odd ? lo : hi;
        const u2v recv = u2v{(unsigned)__shfl_xor((int)send[0], 1, 64), (unsigned)__shfl_xor((int)send[1], 1, 64)};
        r4[2 * pr] = odd ? recv : lo;
        r4[2 * pr + 1] = odd ? hi : recv;
      }
    };
    wt_store(0, wt_load(0));
    request(0, rp0, rs0);
    request(1, rp1, rs1);
    lds_barrier();   // latent images and the first W'^T tile are in LDS
    // this wave's share of every tile's weight-gradient contraction: two of the six [16 genes x 16 columns] blocks
    // (block b = private | shared columns 0..15 | shared columns 16..31, gene half gh; role = 3 gh + b): wave w takes roles w and
    // (w + 4) % 6 -- waves 2 and 3 repeat roles 0 and 1 as their second one (same operands, same bits, same address: a harmless duplicate
    // that keeps the instruction stream of the four waves identical)
    const int role0 = wave, role1 = (wave + 4) % 6;
    float* const wslab = dw_part + (long)blockIdx.x * p.Gp * DEC_KPS;
    // No validity masks on the elements: cells beyond the batch hold t = 0 and T = 0 (the likelihood pass writes them with weight 0), so
    // their corrected values are exact zeros; padding genes (>= G) get t' = -softmax * T != 0, but they meet W' rows that are zero in
    // the packed image (nothing reaches the latent gradient) and their weight-gradient rows are never read by the reduction.
    auto tile = [&](const int t, u2v (&rpA)[4], u2v (&rsA)[4]) {
      const int g0 = gbeg + 32 * t, gn = gbeg + 32 * min(t + 1, ntile - 1);
      f16v yp, ys;
#pragma unroll
      for (int q = 0; q < 16; ++q) { yp[q] = 0.f; ys[q] = 0.f; }
      yp = mfma32_split<3>(wA.hi[0], wA.lo[0], cf.hi[0], cf.lo[0], yp);
      ys = mfma32_split<3>(wA.hi[1], wA.lo[1], cf.hi[1], cf.lo[1], ys);
      ys = mfma32_split<3>(wA.hi[2], wA.lo[2], cf.hi[2], cf.lo[2], ys);
      PsW wB;
      load_ps_w(p, gn, lane, wB);                 // 6 loads: next tile's W' fragments
      const u4v wst = wt_load(t + 1);             // 1 load : next tile's W'^T piece
      unsigned cpk[8], csk[8];
      unswap(rpA);
      unswap(rsA);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        float vp[4], vs[4];
        decode4(rpA[qq], vp);
        decode4(rsA[qq], vs);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = 4 * qq + j;
          vp[j] = vp[j] - fast_exp(yp[q] - lp) * tpb;
          vs[j] = vs[j] - fast_exp(ys[q] - ls) * tsb;
        }
        cpk[2 * qq] = pack2bf(vp[0], vp[1]); cpk[2 * qq + 1] = pack2bf(vp[2], vp[3]);
        csk[2 * qq] = pack2bf(vs[0], vs[1]); csk[2 * qq + 1] = pack2bf(vs[2], vs[3]);
      }
      request(t + 2, rpA, rsA);                   // 4 loads: the tile after next, into the registers just decoded
      const bf16_t* wT = s_wT + (t & 1) * DEC_KPS * HB_WT_PITCH;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const u4v bPw = u4v{cpk[4 * m], cpk[4 * m + 1], cpk[4 * m + 2], cpk[4 * m + 3]}, bSw = u4v{csk[4 * m], csk[4 * m + 1], csk[4 * m + 2], csk[4 * m + 3]};
        const s8v bP = *reinterpret_cast<const s8v*>(&bPw), bS = *reinterpret_cast<const s8v*>(&bSw);
        const bf16_t* wr = wT + 16 * m + 4 * h;
        const u2v p0 = *reinterpret_cast<const u2v*>(wr + (r & 15) * HB_WT_PITCH), p1 = *reinterpret_cast<const u2v*>(wr + (r & 15) * HB_WT_PITCH + 8);
        const u2v s0 = *reinterpret_cast<const u2v*>(wr + (DEC_KP + r) * HB_WT_PITCH), s1 = *reinterpret_cast<const u2v*>(wr + (DEC_KP + r) * HB_WT_PITCH + 8);
        u4v ap = u4v{p0[0], p0[1], p1[0], p1[1]};
        if (r >= DEC_KP) ap = u4v{0u, 0u, 0u, 0u};   // the private head has 16 rows
        const u4v as4 = u4v{s0[0], s0[1], s1[0], s1[1]};
        accP = mfma32(*reinterpret_cast<const s8v*>(&ap), bP, accP);
        accS = mfma32(*reinterpret_cast<const s8v*>(&as4), bS, accS);
        // park the tile (64-byte rows, chunks XOR-swizzled with (row >> 1) & 7: conflict-free stores and transposed reads)
        const int row = wave * 32 + r, sw = (row >> 1) & 7;
        bf16_t* tp_ = s_tile + ((t & 1) * 2 + 0) * SMB_TILE_ELEMS + row * 32;
        bf16_t* ts_ = tp_ + SMB_TILE_ELEMS;
        const int ca = 4 * ((4 * m + h) ^ sw), cb = 4 * ((4 * m + 2 + h) ^ sw);
        *reinterpret_cast<u2v*>(tp_ + ca) = u2v{bPw[0], bPw[1]};
        *reinterpret_cast<u2v*>(tp_ + cb) = u2v{bPw[2], bPw[3]};
        *reinterpret_cast<u2v*>(ts_ + ca) = u2v{bSw[0], bSw[1]};
        *reinterpret_cast<u2v*>(ts_ + cb) = u2v{bSw[2], bSw[3]};
      }
      wt_store(t + 1, wst);   // buffer (t + 1) & 1 was last read during tile t - 1, i.e. before the previous barrier
      lds_barrier();          // the four parked tiles of gene tile t and the next W'^T tile are visible
      // ---- this wave's two weight-gradient blocks: D[16 genes][16 columns] = sum over 128 cells, v_mfma_f32_16x16x32_bf16 ---------------
      {
        const int kq = 8 * (lane >> 4) + ((lane & 15) >> 2), p4 = lane & 3;   // row inside a 32-cell k-step this lane addresses; its 4-column group
        const int b0 = role0 % 3, g0h = role0 / 3, b1 = role1 % 3, g1h = role1 / 3;
        const bf16_t* img0 = s_tile + ((t & 1) * 2 + (b0 ? 1 : 0)) * SMB_TILE_ELEMS;
        const bf16_t* img1 = s_tile + ((t & 1) * 2 + (b1 ? 1 : 0)) * SMB_TILE_ELEMS;
        const bf16_t* z0 = s_zh + (b0 ? DEC_CELLS_PER_WG * 32 + 16 * (b0 - 1) : 0) + 4 * p4;   // latent image of the block's head, its 16 columns
        const bf16_t* z1 = s_zh + (b1 ? DEC_CELLS_PER_WG * 32 + 16 * (b1 - 1) : 0) + 4 * p4;
        f4acc d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int row0 = 32 * ks + kq, row1 = row0 + 4, s0 = (row0 >> 1) & 7, s1 = (row1 >> 1) & 7;
          const s4v a00 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(img0 + row0 * 32 + 4 * ((4 * g0h + p4) ^ s0)));
          const s4v a01 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(img0 + row1 * 32 + 4 * ((4 * g0h + p4) ^ s1)));
          const s4v a10 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(img1 + row0 * 32 + 4 * ((4 * g1h + p4) ^ s0)));
          const s4v a11 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(img1 + row1 * 32 + 4 * ((4 * g1h + p4) ^ s1)));
          const s4v b00 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(z0 + row0 * 32));
          const s4v b01 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(z0 + row1 * 32));
          const s4v b10 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(z1 + row0 * 32));
          const s4v b11 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(z1 + row1 * 32));
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s8v{a00[0], a00[1], a00[2], a00[3], a01[0], a01[1], a01[2], a01[3]},
                                                       s8v{b00[0], b00[1], b00[2], b00[3], b01[0], b01[1], b01[2], b01[3]}, d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(s8v{a10[0], a10[1], a10[2], a10[3], a11[0], a11[1], a11[2], a11[3]},
                                                       s8v{b10[0], b10[1], b10[2], b10[3], b11[0], b11[1], b11[2], b11[3]}, d1, 0, 0, 0);
        }
        // d[j]: gene g0 + 16 gh + 4 (lane >> 4) + j, column 16 b + (lane & 15) of the 48-column slab; rows < Gp always: 8 unconditional stores
        float* o0 = wslab + (long)(g0 + 16 * g0h + 4 * (lane >> 4)) * DEC_KPS + 16 * b0 + (lane & 15);
        float* o1 = wslab + (long)(g0 + 16 * g1h + 4 * (lane >> 4)) * DEC_KPS + 16 * b1 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o0[j * DEC_KPS] = d0[j]; o1[j * DEC_KPS] = d1[j]; }
      }
      wA = wB;
    };
    for (int t = 0; t < ntile; t += 2) {   // (block-uniform trip structure: every wave reaches every barrier)
      tile(t, rp0, rs0);
      if (t + 1 < ntile) tile(t + 1, rp1, rs1);
    }
  }
  // acc[q]: row = crow(q, h) = k, column = lane & 31 = cell; one slab per split, zeros for empty splits
  float* out = dz_part + ((long)split * p.Bp + cell) * DEC_KPS;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int k = crow(q, h);
    if (k < DEC_KP) out[k] = accP[q];
    out[DEC_KP + k] = accS[q];
  }
}
__global__ __launch_bounds__(256, 2) void dec_heads_bwd_kernel(DecParams p, const float* Tp, const float* Ts, float* dz_part, float* dw_part) {
  dec_heads_bwd_body(p, Tp, Ts, dz_part, dw_part);
}
struct HeadsBwdArgs { DecParams p; const float *Tp, *Ts; float *dz_part, *dw_part; };
__global__ __launch_bounds__(256, 2) void dec_heads_bwd_pair_kernel(HeadsBwdArgs a0, HeadsBwdArgs a1) {
  const HeadsBwdArgs a = blockIdx.z ? a1 : a0;
  if ((int)blockIdx.x * DEC_CELLS_PER_WG >= a.p.Bp || (int)blockIdx.y >= a.p.gene_splits) return;   // (workgroup-uniform)
  dec_heads_bwd_body(a.p, a.Tp, a.Ts, a.dz_part, a.dw_part);
}


// ---- one-pass decoder backward for split-bf16 gradient words ("fp32" precision mode, round 4) -----------------------------------------
// dec_heads_bwd_kernel's structure (every wave in every tile's contraction, one barrier per tile, an instruction stream that is the same for
// every wave and tile) on the hi / lo planes the likelihood kernel writes in this mode: t = hi + lo (~2^-17), the corrected value is split
// again, and every product runs as hi * lo + lo * hi + hi * hi (the arithmetic of dec_softmax_bwd_kernel<split_t, true> for the latent
// gradient and of the split-bf16 GEMMs for the two regressor weight gradients).  It replaces, per group, the in-place softmax fix (reads AND
// writes both planes of t_P / t_S: 4 x [B, G] x 2 B each way) and the two register-staged weight-gradient GEMMs that re-read them: one
// read of the four planes instead of two reads and one write.
// LDS (dynamic, 112 128 B: one workgroup per CU): W'^T tiles [parity][hi | lo][48][36], parked tiles [parity][head][hi | lo][128 cells][32 genes]
// (chunk-swizzled as in the bf16 kernel), latent images [head][hi | lo][128][32].  Per tile and wave the vector-memory stream is
// 6 fragment loads, 2 staging loads, 8 tile loads (two tiles ahead), 8 stores -- in that order, always.
constexpr int HBS_WT_ELEMS = DEC_KPS * HB_WT_PITCH;             // one plane of one staged W'^T tile
constexpr int HBS_LDS_BYTES = (2 * 2 * HBS_WT_ELEMS + 2 * 2 * 2 * SMB_TILE_ELEMS + 2 * 2 * DEC_CELLS_PER_WG * 32) * 2;
__global__ __launch_bounds__(256, 1) void dec_heads_bwd_split_kernel(DecParams p, const float* Tp, const float* Ts, float* dz_part, float* dw_part) {
  extern __shared__ __attribute__((aligned(16))) unsigned char hbs_smem[];
  bf16_t* const s_wT = reinterpret_cast<bf16_t*>(hbs_smem);                   // [parity][hl][k][gene of the tile]
  bf16_t* const s_tile = s_wT + 2 * 2 * HBS_WT_ELEMS;                          // [parity][head][hl][cell][gene], chunk-swizzled
  bf16_t* const s_zh = s_tile + 2 * 2 * 2 * SMB_TILE_ELEMS;                    // [head][hl][cell][32 columns] of the latent image
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
  const int cell_tile = blockIdx.x * (DEC_CELLS_PER_WG / 32) + wave;
  const int cell0 = cell_tile * 32;
  const int cell = cell0 + r;
  const int split = blockIdx.y;
  const long plane = (long)p.Bp * p.Gp;
  PsFrags cf;
  load_ps_cell_frags(p, cell0, lane, cf);
  const float lp = p.lse_p[cell], ls = p.lse_s[cell];
  const float tpb = Tp[cell], tsb = Ts[cell];
  const int gbeg = split * p.genes_per_split;
  int gend = gbeg + p.genes_per_split;
  if (gend > p.Gp) gend = p.Gp;
  if (gend > ((p.G + 31) & ~31)) gend = (p.G + 31) & ~31;
  const int ntile = (gend - gbeg) >> 5;
  f16v accP, accS;
#pragma unroll
  for (int q = 0; q < 16; ++q) { accP[q] = 0.f; accS[q] = 0.f; }
  if (ntile > 0) {   // (block-uniform)
    {   // the workgroup's 128 rows of the latent image: one 32-column image per head and plane
      const int cb0 = blockIdx.x * DEC_CELLS_PER_WG;
      for (int i = threadIdx.x; i < 2 * 2 * DEC_CELLS_PER_WG * 4; i += 256) {   // 8-element chunks: 4 per row, head and plane
        const int hd = i / (2 * DEC_CELLS_PER_WG * 4), hl = (i / (DEC_CELLS_PER_WG * 4)) & 1, row = (i / 4) % DEC_CELLS_PER_WG, c8 = (i % 4) * 8;
        const bf16_t* src = hl ? p.Aps_lo : p.Aps_hi;
        u4v v = u4v{0u, 0u, 0u, 0u};
        if (hd == 1 || c8 < DEC_KP) v = *reinterpret_cast<const u4v*>(src + (long)(cb0 + row) * DEC_KPS + (hd ? DEC_KP : 0) + c8);
        *reinterpret_cast<u4v*>(s_zh + ((hd * 2 + hl) * DEC_CELLS_PER_WG + row) * 32 + c8) = v;
      }
    }
    // W'^T staging (both planes): thread i < 192 owns gene i / 6 of a tile and the 8 columns 8 (i % 6) ..; threads 192.. repeat pieces 0..63
    const int st_i = (threadIdx.x < 192) ? (int)threadIdx.x : (int)threadIdx.x - 192;
    const int st_g = st_i / (DEC_KPS / 8), st_c = (st_i % (DEC_KPS / 8)) * 8;
    auto wt_store = [&](int t, int hl, const u4v& w) {
      bf16_t* dst = s_wT + ((t & 1) * 2 + hl) * HBS_WT_ELEMS + st_c * HB_WT_PITCH + st_g;
#pragma unroll
      for (int j = 0; j < 4; ++j) { dst[(2 * j) * HB_WT_PITCH] = (bf16_t)(w[j] & 0xFFFFu); dst[(2 * j + 1) * HB_WT_PITCH] = (bf16_t)(w[j] >> 16); }
    };
    auto wt_off = [&](int t) { return (long)(gbeg + 32 * min(t, ntile - 1) + st_g) * DEC_KPS + st_c; };
    const long trow = (long)cell_tile * p.n_gene_tiles;
    PsW wA;
    u2v rp0[4], rs0[4], rp1[4], rs1[4];       // hi planes, two register stages
    u2v rpl0[4], rsl0[4], rpl1[4], rsl1[4];   // lo planes
    load_ps_w(p, gbeg, lane, wA);
    auto request = [&](int t, u2v (&rp)[4], u2v (&rs)[4], u2v (&rpl)[4], u2v (&rsl)[4]) {   // 16-byte loads of the lane pair's words (see dec_heads_bwd_kernel)
      const long tb = (trow + ((gbeg + 32 * min(t, ntile - 1)) >> 5)) * 1024 + lane * 4;
      const long tb2 = tb - (lane & 1) * 4 + (lane & 1) * 256;
      const bf16_t* tP = reinterpret_cast<const bf16_t*>(p.tP);
      const bf16_t* tS = reinterpret_cast<const bf16_t*>(p.tS);
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const u4v a = *reinterpret_cast<const u4v*>(tP + tb2 + 512 * pr);
        const u4v b = *reinterpret_cast<const u4v*>(tS + tb2 + 512 * pr);
        const u4v al = *reinterpret_cast<const u4v*>(tP + plane + tb2 + 512 * pr);
        const u4v bl = *reinterpret_cast<const u4v*>(tS + plane + tb2 + 512 * pr);
        rp[2 * pr] = u2v{a[0], a[1]}; rp[2 * pr + 1] = u2v{a[2], a[3]};
        rs[2 * pr] = u2v{b[0], b[1]}; rs[2 * pr + 1] = u2v{b[2], b[3]};
        rpl[2 * pr] = u2v{al[0], al[1]}; rpl[2 * pr + 1] = u2v{al[2], al[3]};
        rsl[2 * pr] = u2v{bl[0], bl[1]}; rsl[2 * pr + 1] = u2v{bl[2], bl[3]};
      }
    };
    auto unswap = [&](u2v (&r4)[4]) {
      const bool odd = lane & 1;
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        const u2v lo = r4[2 * pr], hi = r4[2 * pr + 1];
        const u2v send = odd ? lo : hi;
        const u2v recv = u2v{(unsigned)__shfl_xor((int)send[0], 1, 64), (unsigned)__shfl_xor((int)send[1], 1, 64)};
        r4[2 * pr] = odd ? recv : lo;
        r4[2 * pr + 1] = odd ? hi : recv;
      }
    };
    wt_store(0, 0, *reinterpret_cast<const u4v*>(p.Wps_hi + wt_off(0)));
    wt_store(0, 1, *reinterpret_cast<const u4v*>(p.Wps_lo + wt_off(0)));
    request(0, rp0, rs0, rpl0, rsl0);
    request(1, rp1, rs1, rpl1, rsl1);
    lds_barrier();   // latent images and the first W'^T tile are in LDS
    const int role0 = wave, role1 = (wave + 4) % 6;   // this wave's two [16 genes x 16 columns] blocks of every tile (dec_heads_bwd_kernel)
    float* const wslab = dw_part + (long)blockIdx.x * p.Gp * DEC_KPS;
    auto lo2 = [](unsigned hi2, float a, float b) { return pack2bf(a - __uint_as_float(hi2 << 16), b - __uint_as_float(hi2 & 0xffff0000u)); };
    auto tile = [&](const int t, u2v (&rpA)[4], u2v (&rsA)[4], u2v (&rplA)[4], u2v (&rslA)[4]) {
      const int g0 = gbeg + 32 * t, gn = gbeg + 32 * min(t + 1, ntile - 1);
      f16v yp, ys;
#pragma unroll
      for (int q = 0; q < 16; ++q) { yp[q] = 0.f; ys[q] = 0.f; }
      yp = mfma32_split<3>(wA.hi[0], wA.lo[0], cf.hi[0], cf.lo[0], yp);
      ys = mfma32_split<3>(wA.hi[1], wA.lo[1], cf.hi[1], cf.lo[1], ys);
      ys = mfma32_split<3>(wA.hi[2], wA.lo[2], cf.hi[2], cf.lo[2], ys);
      PsW wB;
      load_ps_w(p, gn, lane, wB);                                                        // 6 loads: next tile's W' fragments
      const u4v wsth = *reinterpret_cast<const u4v*>(p.Wps_hi + wt_off(t + 1));          // 2 loads: next tile's W'^T pieces
      const u4v wstl = *reinterpret_cast<const u4v*>(p.Wps_lo + wt_off(t + 1));
      unsigned cpk[8], csk[8], cpl[8], csl[8];   // corrected values: hi words and the bf16 of what the first rounding left
      unswap(rpA); unswap(rsA); unswap(rplA); unswap(rslA);
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        float vp[4], vs[4], wp[4], wsv[4];
        decode4(rpA[qq], vp); decode4(rplA[qq], wp);
        decode4(rsA[qq], vs); decode4(rslA[qq], wsv);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int q = 4 * qq + j;
          vp[j] = (vp[j] + wp[j]) - fast_exp(yp[q] - lp) * tpb;
          vs[j] = (vs[j] + wsv[j]) - fast_exp(ys[q] - ls) * tsb;
        }
        cpk[2 * qq] = pack2bf(vp[0], vp[1]); cpk[2 * qq + 1] = pack2bf(vp[2], vp[3]);
        csk[2 * qq] = pack2bf(vs[0], vs[1]); csk[2 * qq + 1] = pack2bf(vs[2], vs[3]);
        cpl[2 * qq] = lo2(cpk[2 * qq], vp[0], vp[1]); cpl[2 * qq + 1] = lo2(cpk[2 * qq + 1], vp[2], vp[3]);
        csl[2 * qq] = lo2(csk[2 * qq], vs[0], vs[1]); csl[2 * qq + 1] = lo2(csk[2 * qq + 1], vs[2], vs[3]);
      }
      request(t + 2, rpA, rsA, rplA, rslA);       // 8 loads: the tile after next, into the registers just decoded
      const bf16_t* wTh = s_wT + ((t & 1) * 2 + 0) * HBS_WT_ELEMS;
      const bf16_t* wTl = wTh + HBS_WT_ELEMS;
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const u4v bPw = u4v{cpk[4 * m], cpk[4 * m + 1], cpk[4 * m + 2], cpk[4 * m + 3]}, bSw = u4v{csk[4 * m], csk[4 * m + 1], csk[4 * m + 2], csk[4 * m + 3]};
        const u4v bPlw = u4v{cpl[4 * m], cpl[4 * m + 1], cpl[4 * m + 2], cpl[4 * m + 3]}, bSlw = u4v{csl[4 * m], csl[4 * m + 1], csl[4 * m + 2], csl[4 * m + 3]};
        const s8v bP = *reinterpret_cast<const s8v*>(&bPw), bS = *reinterpret_cast<const s8v*>(&bSw);
        const s8v bPl = *reinterpret_cast<const s8v*>(&bPlw), bSl = *reinterpret_cast<const s8v*>(&bSlw);
        auto afrag = [&](const bf16_t* wT, int row, bool zero) {
          const bf16_t* wr = wT + 16 * m + 4 * h + row * HB_WT_PITCH;
          const u2v x0 = *reinterpret_cast<const u2v*>(wr), x1 = *reinterpret_cast<const u2v*>(wr + 8);
          u4v a4 = u4v{x0[0], x0[1], x1[0], x1[1]};
          if (zero) a4 = u4v{0u, 0u, 0u, 0u};
          return *reinterpret_cast<const s8v*>(&a4);
        };
        const s8v ap = afrag(wTh, r & 15, r >= DEC_KP), apl = afrag(wTl, r & 15, r >= DEC_KP);   // the private head has 16 rows
        const s8v as8 = afrag(wTh, DEC_KP + r, false), asl = afrag(wTl, DEC_KP + r, false);
        accP = mfma32(ap, bPl, accP);   // small terms first, as mfma32_split
        accP = mfma32(apl, bP, accP);
        accP = mfma32(ap, bP, accP);
        accS = mfma32(as8, bSl, accS);
        accS = mfma32(asl, bS, accS);
        accS = mfma32(as8, bS, accS);
        // park the tile's two planes (64-byte rows, chunks XOR-swizzled with (row >> 1) & 7: conflict-free stores and transposed reads)
        const int row = wave * 32 + r, sw = (row >> 1) & 7;
        bf16_t* tph = s_tile + (((t & 1) * 2 + 0) * 2 + 0) * SMB_TILE_ELEMS + row * 32;
        bf16_t* tpl = tph + SMB_TILE_ELEMS;
        bf16_t* tsh = tph + 2 * SMB_TILE_ELEMS;
        bf16_t* tsl = tph + 3 * SMB_TILE_ELEMS;
        const int ca = 4 * ((4 * m + h) ^ sw), cb = 4 * ((4 * m + 2 + h) ^ sw);
        *reinterpret_cast<u2v*>(tph + ca) = u2v{bPw[0], bPw[1]};   *reinterpret_cast<u2v*>(tph + cb) = u2v{bPw[2], bPw[3]};
        *reinterpret_cast<u2v*>(tpl + ca) = u2v{bPlw[0], bPlw[1]}; *reinterpret_cast<u2v*>(tpl + cb) = u2v{bPlw[2], bPlw[3]};
        *reinterpret_cast<u2v*>(tsh + ca) = u2v{bSw[0], bSw[1]};   *reinterpret_cast<u2v*>(tsh + cb) = u2v{bSw[2], bSw[3]};
        *reinterpret_cast<u2v*>(tsl + ca) = u2v{bSlw[0], bSlw[1]}; *reinterpret_cast<u2v*>(tsl + cb) = u2v{bSlw[2], bSlw[3]};
      }
      wt_store(t + 1, 0, wsth);   // buffer (t + 1) & 1 was last read during tile t - 1, i.e. before the previous barrier
      wt_store(t + 1, 1, wstl);
      lds_barrier();              // the parked planes of gene tile t and the next W'^T tile are visible
      {   // this wave's two weight-gradient blocks: D[16 genes][16 columns] = sum over 128 cells, three v_mfma_f32_16x16x32_bf16 per 32 cells
        const int kq = 8 * (lane >> 4) + ((lane & 15) >> 2), p4 = lane & 3;
        const int b0 = role0 % 3, g0h = role0 / 3, b1 = role1 % 3, g1h = role1 / 3;
        const bf16_t* img0 = s_tile + (((t & 1) * 2 + (b0 ? 1 : 0)) * 2) * SMB_TILE_ELEMS;   // hi plane; lo plane SMB_TILE_ELEMS further
        const bf16_t* img1 = s_tile + (((t & 1) * 2 + (b1 ? 1 : 0)) * 2) * SMB_TILE_ELEMS;
        const bf16_t* z0 = s_zh + ((b0 ? 1 : 0) * 2) * DEC_CELLS_PER_WG * 32 + (b0 ? 16 * (b0 - 1) : 0) + 4 * p4;   // hi plane of the block's head, its 16 columns
        const bf16_t* z1 = s_zh + ((b1 ? 1 : 0) * 2) * DEC_CELLS_PER_WG * 32 + (b1 ? 16 * (b1 - 1) : 0) + 4 * p4;
        constexpr int ZL = DEC_CELLS_PER_WG * 32;   // lo plane of a latent image
        f4acc d0 = {0.f, 0.f, 0.f, 0.f}, d1 = {0.f, 0.f, 0.f, 0.f};
        auto tr2 = [&](const bf16_t* a, const bf16_t* b) {
          const s4v x = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)a), y = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)b);
          return s8v{x[0], x[1], x[2], x[3], y[0], y[1], y[2], y[3]};
        };
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int row0 = 32 * ks + kq, row1 = row0 + 4, s0 = (row0 >> 1) & 7, s1 = (row1 >> 1) & 7;
          const int o00 = row0 * 32 + 4 * ((4 * g0h + p4) ^ s0), o01 = row1 * 32 + 4 * ((4 * g0h + p4) ^ s1);
          const int o10 = row0 * 32 + 4 * ((4 * g1h + p4) ^ s0), o11 = row1 * 32 + 4 * ((4 * g1h + p4) ^ s1);
          const s8v a0h = tr2(img0 + o00, img0 + o01), a0l = tr2(img0 + SMB_TILE_ELEMS + o00, img0 + SMB_TILE_ELEMS + o01);
          const s8v a1h = tr2(img1 + o10, img1 + o11), a1l = tr2(img1 + SMB_TILE_ELEMS + o10, img1 + SMB_TILE_ELEMS + o11);
          const s8v b0h = tr2(z0 + row0 * 32, z0 + row1 * 32), b0l = tr2(z0 + ZL + row0 * 32, z0 + ZL + row1 * 32);
          const s8v b1h = tr2(z1 + row0 * 32, z1 + row1 * 32), b1l = tr2(z1 + ZL + row0 * 32, z1 + ZL + row1 * 32);
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, b0l, d0, 0, 0, 0);
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0l, b0h, d0, 0, 0, 0);
          d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0h, b0h, d0, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, b1l, d1, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1l, b1h, d1, 0, 0, 0);
          d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1h, b1h, d1, 0, 0, 0);
        }
        float* o0 = wslab + (long)(g0 + 16 * g0h + 4 * (lane >> 4)) * DEC_KPS + 16 * b0 + (lane & 15);
        float* o1 = wslab + (long)(g0 + 16 * g1h + 4 * (lane >> 4)) * DEC_KPS + 16 * b1 + (lane & 15);
#pragma unroll
        for (int j = 0; j < 4; ++j) { o0[j * DEC_KPS] = d0[j]; o1[j * DEC_KPS] = d1[j]; }
      }
      wA = wB;
    };
    for (int t = 0; t < ntile; t += 2) {   // (block-uniform trip structure: every wave reaches every barrier)
      tile(t, rp0, rs0, rpl0, rsl0);
      if (t + 1 < ntile) tile(t + 1, rp1, rs1, rpl1, rsl1);
    }
  }
  float* out = dz_part + ((long)split * p.Bp + cell) * DEC_KPS;
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int k = crow(q, h);
    if (k < DEC_KP) out[k] = accP[q];
    out[DEC_KP + k] = accS[q];
  }
}

}  // namespace spv
