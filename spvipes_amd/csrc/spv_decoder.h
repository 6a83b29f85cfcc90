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
//   * No [B,G] fp32 intermediate ever exists in HBM.  The forward pass emits, per (cell, gene), the
//     three gradients the backward GEMMs need (d/d logits, d/d log mu1, d/d log mu2) as bf16/fp32
//     so that the backward pass re-evaluates no transcendental.
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
  const bf16_t* Wm_hi; const bf16_t* Wm_lo; int KMp; int ksteps_m;   // [Gp][KMp]  mixture weights | bias
  const bf16_t* Am_hi; const bf16_t* Am_lo;                           // [Bp][KMp]  relu(BN(.)) | z | 1
  const bf16_t* Wps_hi; const bf16_t* Wps_lo;                         // [Gp][48]   W'_p|c_p|0.. W'_s|c_s|0..
  const bf16_t* Aps_hi; const bf16_t* Aps_lo;                         // [Bp][48]   z_p|1|0..   z_s|1|0..
  // per gene / per cell vectors
  const float4* gene_tab;       // [Gp] {theta, log(theta+eps), 1/(theta+eps), 0}
  const float2* cnt_tab;        // [NB_CMAX][Gp] {F, Psi}
  const float* a_p; const float* a_s;   // [Bp] library - lse_k
  const float* lse_p; const float* lse_s;
  const float* w_row;           // [Bp] weight of each cell in the scalar loss (0 for padding)
  // outputs
  int gene_splits; int genes_per_split;      // multiples of 32
  float* part_max_p; float* part_sum_p; float* part_max_s; float* part_sum_s;  // [splits][Bp]  (lse)
  float* rec_part; float* tp_part; float* ts_part;                             // [splits][Bp]  (nb)
  float* dtheta_part;           // [Bp/128][Gp]
  void* dL; void* tP; void* tS; long ldg; int grads_f32;                       // [Bp][Gp] bf16 or f32
};

// ---- per-gene tables ----------------------------------------------------------
__global__ void nb_tables_kernel(const float* px_r, int G, int Gp, float4* gene_tab, float2* cnt_tab) {
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
    gene_tab[g] = make_float4(theta, fast_log(theta + SPV_EPS_NB), fast_rcp(theta + SPV_EPS_NB), 0.f);
    cnt_tab[g] = make_float2(0.f, 0.f);  // x = 0: lgamma terms cancel exactly
    return;
  }
  const float x = log1p_count((float)c);
  const LgammaDigamma t = lgamma_digamma(theta), xt = lgamma_digamma(x + theta), x1 = lgamma_digamma(x + 1.0f);
  cnt_tab[(long)c * Gp + g] = make_float2(xt.lg - t.lg - x1.lg, xt.dg - t.dg);
}

// counts of the 16 accumulator rows of this lane: cell = cell0 + (lane & 31), genes g0 + crow(q, h)
__device__ __forceinline__ void load_counts16(const DecParams& p, int cell, int g0, int h, bool cell_ok, float (&c)[16]) {
#pragma unroll
  for (int q = 0; q < 16; ++q) c[q] = 0.f;
  if (!cell_ok) return;
  const long row = p.rows ? (long)p.rows[cell] : (long)cell;
#pragma unroll
  for (int qq = 0; qq < 4; ++qq) {
    const int g = g0 + 8 * qq + 4 * h;  // 4 consecutive genes: registers 4qq .. 4qq+3
    if (p.count_is_u16) {
      const unsigned short* src = reinterpret_cast<const unsigned short*>(p.X) + row * p.ldx + p.col_off + g;
      if (g + 4 <= p.G && (reinterpret_cast<uintptr_t>(src) & 7) == 0) {
        const u2v raw = *reinterpret_cast<const u2v*>(src);
        c[4 * qq + 0] = (float)(raw[0] & 0xFFFFu); c[4 * qq + 1] = (float)(raw[0] >> 16);
        c[4 * qq + 2] = (float)(raw[1] & 0xFFFFu); c[4 * qq + 3] = (float)(raw[1] >> 16);
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (g + j < p.G) c[4 * qq + j] = (float)src[j];
      }
    } else {
      const float* src = reinterpret_cast<const float*>(p.X) + row * p.ldx + p.col_off + g;
      if (g + 4 <= p.G && (reinterpret_cast<uintptr_t>(src) & 15) == 0) {
        const f4v raw = *reinterpret_cast<const f4v*>(src);
#pragma unroll
        for (int j = 0; j < 4; ++j) c[4 * qq + j] = raw[j];
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) if (g + j < p.G) c[4 * qq + j] = src[j];
      }
    }
  }
}

// y_p / y_s tiles (32 genes x 32 cells) with split-bf16 operands straight from L2 (K = 16 / 32):
// A fragment = W' rows (natural [g][k]), B fragment = the wave's resident z fragments.
struct PsFrags { s8v hi[3], lo[3]; };
__device__ __forceinline__ void load_ps_cell_frags(const DecParams& p, int cell0, int lane, PsFrags& f) {
  const long off = (long)(cell0 + (lane & 31)) * DEC_KPS + 8 * (lane >> 5);
#pragma unroll
  for (int s = 0; s < 3; ++s) {
    f.hi[s] = *reinterpret_cast<const s8v*>(p.Aps_hi + off + 16 * s);
    f.lo[s] = *reinterpret_cast<const s8v*>(p.Aps_lo + off + 16 * s);
  }
}
__device__ __forceinline__ void ps_tiles(const DecParams& p, int g0, int lane, const PsFrags& cf, f16v& yp, f16v& ys) {
  const long off = (long)(g0 + (lane & 31)) * DEC_KPS + 8 * (lane >> 5);
#pragma unroll
  for (int q = 0; q < 16; ++q) { yp[q] = 0.f; ys[q] = 0.f; }
  s8v a_hi = *reinterpret_cast<const s8v*>(p.Wps_hi + off), a_lo = *reinterpret_cast<const s8v*>(p.Wps_lo + off);
  yp = mfma32_split<3>(a_hi, a_lo, cf.hi[0], cf.lo[0], yp);
#pragma unroll
  for (int s = 1; s < 3; ++s) {
    a_hi = *reinterpret_cast<const s8v*>(p.Wps_hi + off + 16 * s);
    a_lo = *reinterpret_cast<const s8v*>(p.Wps_lo + off + 16 * s);
    ys = mfma32_split<3>(a_hi, a_lo, cf.hi[s], cf.lo[s], ys);
  }
}

// ---- pass 1: per-cell log-sum-exp over genes of y_p and y_s (softmax denominators) ------------
__global__ __launch_bounds__(256) void dec_lse_kernel(DecParams p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5;
  const int cell0 = blockIdx.x * DEC_CELLS_PER_WG + 32 * wave;
  const int split = blockIdx.y;
  PsFrags cf;
  load_ps_cell_frags(p, cell0, lane, cf);
  float mp = -INFINITY, sp = 0.f, ms = -INFINITY, ss = 0.f;
  const int gbeg = split * p.genes_per_split;
  int gend = gbeg + p.genes_per_split;
  if (gend > p.Gp) gend = p.Gp;
  for (int g0 = gbeg; g0 < gend; g0 += 32) {
    if (g0 >= p.G) break;
    f16v yp, ys;
    ps_tiles(p, g0, lane, cf, yp, ys);
    float tmp = -INFINITY, tms = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const bool ok = g0 + crow(q, h) < p.G;
      yp[q] = ok ? yp[q] : -INFINITY;
      ys[q] = ok ? ys[q] : -INFINITY;
      tmp = fmaxf(tmp, yp[q]);
      tms = fmaxf(tms, ys[q]);
    }
    const float nmp = fmaxf(mp, tmp), nms = fmaxf(ms, tms);
    // a half-wave can see only masked genes in the last tile: keep exp() away from (-inf) - (-inf)
    const float smp = (nmp == -INFINITY) ? 0.f : nmp, sms = (nms == -INFINITY) ? 0.f : nms;
    sp *= fast_exp(mp - smp);
    ss *= fast_exp(ms - sms);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      sp += fast_exp(yp[q] - smp);
      ss += fast_exp(ys[q] - sms);
    }
    mp = nmp; ms = nms;
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

// lse_k[b] = log sum_splits ...;  a_k[b] = library[b] - lse_k[b]   (log of exp(library) * softmax)
__global__ void dec_lse_combine_kernel(const float* pmp, const float* psp, const float* pms, const float* pss, int splits,
                                       int Bp, int B, const float* library, float* lse_p, float* lse_s, float* a_p, float* a_s) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= Bp) return;
  float mp = -INFINITY, ms = -INFINITY;
  for (int s = 0; s < splits; ++s) { mp = fmaxf(mp, pmp[(long)s * Bp + b]); ms = fmaxf(ms, pms[(long)s * Bp + b]); }
  float sp = 0.f, ss = 0.f;
  for (int s = 0; s < splits; ++s) {
    const float m1 = pmp[(long)s * Bp + b], m2 = pms[(long)s * Bp + b];
    if (m1 != -INFINITY) sp += psp[(long)s * Bp + b] * __expf(m1 - mp);
    if (m2 != -INFINITY) ss += pss[(long)s * Bp + b] * __expf(m2 - ms);
  }
  const float lp = mp + __logf(sp), ls = ms + __logf(ss);
  const float lib = (b < B) ? library[b] : 0.f;
  lse_p[b] = lp; lse_s[b] = ls;
  a_p[b] = lib - lp; a_s[b] = lib - ls;
}

// ---- pass 2: NB-mixture log-likelihood, its row sums and (TRAIN) its per-element gradients ----
template <typename GT>
__device__ __forceinline__ void store4(void* base, long ld, int cell, int g, const float (&v)[4]);
template <>
__device__ __forceinline__ void store4<bf16_t>(void* base, long ld, int cell, int g, const float (&v)[4]) {
  u2v w;
  w[0] = (unsigned)f2bf(v[0]) | ((unsigned)f2bf(v[1]) << 16);
  w[1] = (unsigned)f2bf(v[2]) | ((unsigned)f2bf(v[3]) << 16);
  *reinterpret_cast<u2v*>(reinterpret_cast<bf16_t*>(base) + (long)cell * ld + g) = w;
}
template <>
__device__ __forceinline__ void store4<float>(void* base, long ld, int cell, int g, const float (&v)[4]) {
  *reinterpret_cast<f4v*>(reinterpret_cast<float*>(base) + (long)cell * ld + g) = f4v{v[0], v[1], v[2], v[3]};
}

constexpr int WM_PAD = 8;  // LDS pitch of the mixture-weight tile = KMp + 8 (conflict-free 16-B row reads)

template <int KSTEPS, int NSPLIT, bool TRAIN, typename GT>
__global__ __launch_bounds__(256) void dec_nb_kernel(DecParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int pitch = p.KMp + WM_PAD;
  bf16_t* sW = reinterpret_cast<bf16_t*>(smem_raw);              // [32][pitch] hi
  bf16_t* sW_lo = sW + 32 * pitch;                               // [32][pitch] lo (NSPLIT == 3)
  float* sdth = reinterpret_cast<float*>(sW + 32 * pitch * (NSPLIT == 3 ? 2 : 1));  // [4 waves][32 genes]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, r = lane & 31;
  const int cell0 = blockIdx.x * DEC_CELLS_PER_WG + 32 * wave;
  const int cell = cell0 + r;
  const bool cell_ok = cell < p.B;
  const int split = blockIdx.y;
  const int ksteps = (NSPLIT == 1) ? KSTEPS : p.ksteps_m;

  // resident cell-side fragments
  PsFrags cf;
  load_ps_cell_frags(p, cell0, lane, cf);
  s8v bm[(NSPLIT == 1) ? KSTEPS : 1];
  if constexpr (NSPLIT == 1) {
#pragma unroll
    for (int s = 0; s < KSTEPS; ++s) bm[s] = *reinterpret_cast<const s8v*>(p.Am_hi + (long)cell * p.KMp + 16 * s + 8 * h);
  }
  const float ap = p.a_p[cell], as_ = p.a_s[cell];
  const float w = p.w_row[cell];
  float rec = 0.f, tp_sum = 0.f, ts_sum = 0.f;

  const int gbeg = split * p.genes_per_split;
  int gend = gbeg + p.genes_per_split;
  if (gend > p.Gp) gend = p.Gp;
  for (int g0 = gbeg; g0 < gend; g0 += 32) {
    if (g0 >= p.G) break;  // uniform over the workgroup
    __syncthreads();
    // stage the mixture-weight tile: 32 rows x KMp bf16, 16-byte chunks
    {
      const int cpr = p.KMp / 8;
      for (int c = tid; c < 32 * cpr; c += 256) {
        const int row = c / cpr, cc = c % cpr;
        *reinterpret_cast<u4v*>(sW + row * pitch + 8 * cc) = *reinterpret_cast<const u4v*>(p.Wm_hi + (long)(g0 + row) * p.KMp + 8 * cc);
        if constexpr (NSPLIT == 3)
          *reinterpret_cast<u4v*>(sW_lo + row * pitch + 8 * cc) = *reinterpret_cast<const u4v*>(p.Wm_lo + (long)(g0 + row) * p.KMp + 8 * cc);
      }
    }
    float cnt[16];
    load_counts16(p, cell, g0, h, cell_ok, cnt);
    f16v yp, ys;
    ps_tiles(p, g0, lane, cf, yp, ys);
    __syncthreads();
    f16v lg;
#pragma unroll
    for (int q = 0; q < 16; ++q) lg[q] = 0.f;
    if constexpr (NSPLIT == 1) {
#pragma unroll
      for (int s = 0; s < KSTEPS; ++s) lg = mfma32(frag_natural(sW, pitch, 0, 16 * s, lane), bm[s], lg);
    } else {
      for (int s = 0; s < ksteps; ++s) {
        const long bo = (long)cell * p.KMp + 16 * s + 8 * h;
        const s8v b_hi = *reinterpret_cast<const s8v*>(p.Am_hi + bo), b_lo = *reinterpret_cast<const s8v*>(p.Am_lo + bo);
        lg = mfma32_split<3>(frag_natural(sW, pitch, 0, 16 * s, lane), frag_natural(sW_lo, pitch, 0, 16 * s, lane), b_hi, b_lo, lg);
      }
    }

    float dth[16];
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      float o_dl[4], o_tp[4], o_ts[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = 4 * qq + j;
        const int g = g0 + crow(q, h);
        const bool ok = cell_ok && (g < p.G);
        const float4 gt = p.gene_tab[g];
        const float theta = gt.x, lt = gt.y, ith = gt.z;
        const float c = cnt[q];
        const float x = log1p_count(c);
        float F = 0.f, Psi = 0.f;
        if (c > 0.f) {
          if (c < (float)NB_CMAX) {
            const float2 t = p.cnt_tab[(long)(int)c * p.Gp + g];
            F = t.x; Psi = t.y;
          } else {
            const LgammaDigamma a = lgamma_digamma(theta), b = lgamma_digamma(x + theta), d = lgamma_digamma(x + 1.0f);
            F = b.lg - a.lg - d.lg; Psi = b.dg - a.dg;
          }
        }
        const float ell = lg[q];
        const float mu1 = fast_exp(yp[q] + ap), mu2 = fast_exp(ys[q] + as_);
        const float S1 = theta + mu1 + SPV_EPS_NB, S2 = theta + mu2 + SPV_EPS_NB;
        const float L1 = fast_log(S1), L2 = fast_log(S2);
        const float e1 = mu1 + SPV_EPS_NB, e2 = mu2 + SPV_EPS_NB;
        const float nb1 = theta * (lt - L1) + x * (fast_log(e1) - L1);
        const float v2 = theta * (lt - L2) + x * (fast_log(e2) - L2) - ell;
        const float d = nb1 - v2, M = fmaxf(nb1, v2);
        const float ed = fast_exp(-fabsf(d)), el = fast_exp(-fabsf(ell));
        const float iol = fast_rcp(1.0f + el);
        const float logp = M - fmaxf(-ell, 0.f) + fast_log((1.0f + ed) * iol) + F;
        rec -= ok ? logp : 0.f;
        if constexpr (TRAIN) {
          const float iod = fast_rcp(1.0f + ed);
          const float r1 = (d >= 0.f) ? iod : ed * iod, r2 = 1.0f - r1;
          const float sig = (ell <= 0.f) ? iol : el * iol;  // sigmoid(-ell)
          const float iS1 = fast_rcp(S1), iS2 = fast_rcp(S2);
          const float g1 = x * mu1 * fast_rcp(e1) - (theta + x) * mu1 * iS1;
          const float g2 = x * mu2 * fast_rcp(e2) - (theta + x) * mu2 * iS2;
          const float t1 = r1 * g1, t2 = r2 * g2;
          const float dn1 = (lt - L1) + theta * (ith - iS1) - x * iS1;
          const float dn2 = (lt - L2) + theta * (ith - iS2) - x * iS2;
          const float wk = ok ? -w : 0.f;  // loss = sum_b w_b * (-sum_g logp)
          o_dl[j] = wk * (sig - r2);
          o_tp[j] = wk * t1;
          o_ts[j] = wk * t2;
          tp_sum += o_tp[j];
          ts_sum += o_ts[j];
          dth[q] = wk * (r1 * dn1 + r2 * dn2 + Psi);
        }
      }
      if constexpr (TRAIN) {
        if (cell < p.Bp) {
          const int g = g0 + 8 * qq + 4 * h;
          store4<GT>(p.dL, p.ldg, cell, g, o_dl);
          store4<GT>(p.tP, p.ldg, cell, g, o_tp);
          store4<GT>(p.tS, p.ldg, cell, g, o_ts);
        }
      }
    }
    if constexpr (TRAIN) {
      // per-gene sum over this wave's 32 cells, then over the 4 waves (fixed order: deterministic)
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const float s = half_sum(dth[q]);
        if (r == 0) sdth[wave * 32 + crow(q, h)] = s;
      }
      __syncthreads();
      if (tid < 32) {
        const float s = (sdth[tid] + sdth[32 + tid]) + (sdth[64 + tid] + sdth[96 + tid]);
        p.dtheta_part[(long)blockIdx.x * p.Gp + g0 + tid] = s;
      }
    }
  }
  rec += other_half(rec);
  if constexpr (TRAIN) { tp_sum += other_half(tp_sum); ts_sum += other_half(ts_sum); }
  if (h == 0) {
    const long o = (long)split * p.Bp + cell;
    p.rec_part[o] = rec;
    if constexpr (TRAIN) { p.tp_part[o] = tp_sum; p.ts_part[o] = ts_sum; }
  }
}

// ---- backward helper: finish the softmax backward in place -------------------------------------
//   d/dy_k[b,g] = t_k[b,g] - softmax_k[b,g] * T_k[b],   T_k[b] = sum_g t_k[b,g]
// (t_k came out of dec_nb_kernel with a_k = library - lse_k held fixed; the second term is the
// derivative through lse_k).  Re-evaluates y_k with the K = 16/32 split MFMAs and one exp each.
template <typename GT>
__global__ __launch_bounds__(256) void dec_softmax_bwd_kernel(DecParams p, const float* Tp, const float* Ts) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
  const int cell0 = blockIdx.x * DEC_CELLS_PER_WG + 32 * wave;
  const int cell = cell0 + r;
  const int split = blockIdx.y;
  PsFrags cf;
  load_ps_cell_frags(p, cell0, lane, cf);
  const float lp = p.lse_p[cell], ls = p.lse_s[cell];
  const float tpb = Tp[cell], tsb = Ts[cell];
  const int gbeg = split * p.genes_per_split;
  int gend = gbeg + p.genes_per_split;
  if (gend > p.Gp) gend = p.Gp;
  for (int g0 = gbeg; g0 < gend; g0 += 32) {
    if (g0 >= p.G) break;
    f16v yp, ys;
    ps_tiles(p, g0, lane, cf, yp, ys);
#pragma unroll
    for (int qq = 0; qq < 4; ++qq) {
      const int g = g0 + 8 * qq + 4 * h;
      float vp[4], vs[4];
      if constexpr (sizeof(GT) == 2) {
        const u2v a = *reinterpret_cast<const u2v*>(reinterpret_cast<const bf16_t*>(p.tP) + (long)cell * p.ldg + g);
        const u2v b = *reinterpret_cast<const u2v*>(reinterpret_cast<const bf16_t*>(p.tS) + (long)cell * p.ldg + g);
        vp[0] = bf2f(a[0] & 0xFFFF); vp[1] = bf2f(a[0] >> 16); vp[2] = bf2f(a[1] & 0xFFFF); vp[3] = bf2f(a[1] >> 16);
        vs[0] = bf2f(b[0] & 0xFFFF); vs[1] = bf2f(b[0] >> 16); vs[2] = bf2f(b[1] & 0xFFFF); vs[3] = bf2f(b[1] >> 16);
      } else {
        const f4v a = *reinterpret_cast<const f4v*>(reinterpret_cast<const float*>(p.tP) + (long)cell * p.ldg + g);
        const f4v b = *reinterpret_cast<const f4v*>(reinterpret_cast<const float*>(p.tS) + (long)cell * p.ldg + g);
#pragma unroll
        for (int j = 0; j < 4; ++j) { vp[j] = a[j]; vs[j] = b[j]; }
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int q = 4 * qq + j;
        const bool ok = (g + j < p.G) && (cell < p.B);
        vp[j] = ok ? vp[j] - fast_exp(yp[q] - lp) * tpb : 0.f;
        vs[j] = ok ? vs[j] - fast_exp(ys[q] - ls) * tsb : 0.f;
      }
      store4<GT>(p.tP, p.ldg, cell, g, vp);
      store4<GT>(p.tS, p.ldg, cell, g, vs);
    }
  }
}

}  // namespace spv
