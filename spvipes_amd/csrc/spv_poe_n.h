// N-group Product of Experts over cluster ("component") matched cells -- the N-expert generalisation of the reference's
// _product_of_experts (module/spVIPESmodule.py:573-581) that SURVEY.md 8(d) prescribes for BASELINE config 4 (3 groups,
// cluster-matched PoE).  THROUGHPUT ONLY: the reference cannot run more than two groups (data/prepare_adatas.py:94-95,
// spVIPESmodule.py:283-286, 723-726), so there is nothing to be bit-compatible with; tests/test_gpu_three_groups.py checks the
// kernels against a plain torch restatement of the definition below.
//
// Definition.  Cell i of group g with component c = comp_g[i]; for every OTHER group h that has cells of component c in
// its minibatch the expert is the component mean of h's shared-encoder statistics,
//     m_h[c] = mean_{j in h, comp_h[j] = c} (loc_h[j], logvar_h[j]),
// and with the always-on N(0, 1) prior expert (spVIPESmodule.py:346, 576)
//     prec = 1 + exp(-logvar_g[i]) + sum_h exp(-mlogvar_h[c]),     num = loc_g[i] exp(-logvar_g[i]) + sum_h mloc_h[c] exp(-mlogvar_h[c])
//     loc* = num / prec,  logvar* = -log prec,  scale* = exp(logvar* / 2);  the draw and the KL use scale*.clamp(min = 1e-6)
// (for two groups and a uniform transport plan inside a component this is what _cluster_based_poe's plan-weighted experts
// reduce to, :211-229).  All reductions run in a fixed order (per-segment partial sums in LDS, segments summed in index
// order): no atomics, bit-reproducible.
#pragma once
#include "spv_common.h"
#include "../../include/spvipes_hip.h"

namespace spv {

constexpr int PN_SEG = SPV_POE_COMP_SEG;    // cell segments of the component reductions
constexpr int PN_CMAX = SPV_POE_COMP_CMAX;  // component codes < 64

typedef spv_poe_comp_args PoeN;

__device__ __forceinline__ float pn_max32(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float pn_sum32(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// component code of a cell, or -1 when the stored value is not an integral code in [0, ncomp) (negative, too large, NaN): such a
// cell has no component -- it contributes to no component mean and receives no expert (its posterior is fused with the prior
// alone), exactly what the two-group cluster path does with a component that the other group lacks.  The host validates the
// codes once per data set (train.Trainer); this keeps a raw module call with bad codes from indexing outside the tables.
__device__ __forceinline__ int pn_code(float v, int ncomp) {
  return (v >= 0.f && v < (float)ncomp) ? (int)v : -1;
}

// K1: per (segment, group) partial sums of (loc | logvar | 1) per component; thread d owns column d, cells in order
__global__ __launch_bounds__(128) void pn_stats_kernel(PoeN a) {
  __shared__ float acc[PN_CMAX][2 * 32 + 1];
  const int g = blockIdx.y, seg = blockIdx.x, d = threadIdx.x, W = 2 * a.n + 1;
  if (d >= W) return;
  for (int c = 0; c < a.ncomp; ++c) acc[c][d] = 0.f;
  const int per = (a.B[g] + PN_SEG - 1) / PN_SEG, lo = seg * per, hi = min(a.B[g], lo + per);
  for (int b = lo; b < hi; ++b) {
    const int c = pn_code(a.comp[g][b], a.ncomp);
    if (c < 0) continue;
    acc[c][d] += (d < 2 * a.n) ? a.stats[g][(long)b * a.ld[g] + d] : 1.0f;
  }
  float* out = a.part + ((long)(g * PN_SEG + seg) * a.ncomp) * W;
  for (int c = 0; c < a.ncomp; ++c) out[(long)c * W + d] = acc[c][d];
}

// K2: mean[g][c] = (mean loc | mean logvar | count)
__global__ __launch_bounds__(256) void pn_means_kernel(PoeN a) {
  const int g = blockIdx.y, W = 2 * a.n + 1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= a.ncomp * W) return;
  const int c = idx / W, d = idx % W;
  float s = 0.f, cnt = 0.f;
  for (int seg = 0; seg < PN_SEG; ++seg) {
    const float* p = a.part + ((long)(g * PN_SEG + seg) * a.ncomp + c) * W;
    s += p[d];
    cnt += p[W - 1];
  }
  a.mean[((long)g * a.ncomp + c) * W + d] = (d == W - 1) ? cnt : s / fmaxf(cnt, 1.0f);
}

// K3: fusion, draw, softmax, KL.  block = 8 cells x 32 latent dimensions
__global__ __launch_bounds__(256) void pn_fuse_fwd_kernel(PoeN a) {
  const int g = blockIdx.y, n = a.n, W = 2 * n + 1;
  const int b = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  const bool ok = b < a.B[g] && d < n;
  float z = -INFINITY, klt = 0.f, jl = 0.f, jv = 0.f, sc = 0.f;
  const long i = (long)b * n + d;
  if (ok) {
    const int c = pn_code(a.comp[g][b], a.ncomp);
    const float* own = a.stats[g] + (long)b * a.ld[g];
    const float inv = expf(-own[n + d]);
    float prec = 1.0f + inv, num = own[d] * inv;
    for (int h = 0; h < a.ngroups; ++h) {
      if (h == g || c < 0) continue;
      const float* m = a.mean + ((long)h * a.ncomp + c) * W;
      if (m[W - 1] > 0.f) {
        const float w = expf(-m[n + d]);
        prec += w;
        num += m[d] * w;
      }
    }
    const float J = 1.0f / prec;
    jl = num * J; jv = logf(J); sc = sqrtf(J);
    const float sq = fmaxf(sc, 1e-6f);
    z = jl + sq * a.eps[g][i];
    klt = 0.5f * (sq * sq + jl * jl - 1.0f - logf(sq * sq));
  }
  const float mx = pn_max32(z);
  const float e = ok ? expf(z - mx) : 0.f;
  const float sum = pn_sum32(e), kl = pn_sum32(klt);
  if (ok) { a.loc[g][i] = jl; a.logvar[g][i] = jv; a.scale[g][i] = sc; a.logz[g][i] = z; a.theta[g][i] = e / sum; }
  if (b < a.B[g] && d == 0) a.kl[g][b] = kl;
}

// K4: per cell: gradient wrt prec and num (kept for the component reduction) and the cell's own statistics
__global__ __launch_bounds__(256) void pn_cell_bwd_kernel(PoeN a) {
  const int g = blockIdx.y, n = a.n;
  const int b = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  if (b >= a.B[g] || d >= n) return;
  const long i = (long)b * n + d;
  const float jl = a.loc[g][i], sc = a.scale[g][i];
  const bool live = sc >= 1e-6f;
  const float sq = fmaxf(sc, 1e-6f);
  const float gk = ld_or_zero(a.g_kl[g], b), gz = ld_or_zero(a.g_logz[g], i);
  const float Gl = ld_or_zero(a.g_loc[g], i) + gz + gk * jl;
  const float Gs = ld_or_zero(a.g_scale[g], i) + (live ? gz * a.eps[g][i] + gk * (sq - 1.0f / sq) : 0.f);
  const float Gv = ld_or_zero(a.g_logvar[g], i) + 0.5f * sc * Gs;
  const float J = sc * sc;                       // 1 / prec
  const float dN = Gl * J, dP = -J * (Gl * jl + Gv);   // d / d num,  d / d prec
  a.dpn[g][(long)b * 2 * n + d] = dP;
  a.dpn[g][(long)b * 2 * n + n + d] = dN;
  const float* own = a.stats[g] + (long)b * a.ld[g];
  const float inv = expf(-own[n + d]);
  float* ds = a.d_stats[g] + (long)b * a.ld[g];
  ds[d] = dN * inv;
  ds[n + d] = -inv * (dN * own[d] + dP);
}

// K5: per (segment, target group h): d mean_h[c] = sum over the cells of the OTHER groups with component c, in order
__global__ __launch_bounds__(64) void pn_comp_bwd_kernel(PoeN a) {
  __shared__ float acc[PN_CMAX][2 * 32];
  const int h = blockIdx.y, seg = blockIdx.x, d = threadIdx.x, n = a.n, W = 2 * n + 1;
  if (d >= n) return;
  for (int c = 0; c < a.ncomp; ++c) { acc[c][d] = 0.f; acc[c][n + d] = 0.f; }
  for (int g = 0; g < a.ngroups; ++g) {
    if (g == h) continue;
    const int per = (a.B[g] + PN_SEG - 1) / PN_SEG, lo = seg * per, hi = min(a.B[g], lo + per);
    for (int b = lo; b < hi; ++b) {
      const int c = pn_code(a.comp[g][b], a.ncomp);
      if (c < 0) continue;
      const float* m = a.mean + ((long)h * a.ncomp + c) * W;
      if (m[W - 1] <= 0.f) continue;
      const float w = expf(-m[n + d]);
      const float dP = a.dpn[g][(long)b * 2 * n + d], dN = a.dpn[g][(long)b * 2 * n + n + d];
      acc[c][d] += dN * w;                       // d / d mloc
      acc[c][n + d] += -w * (dN * m[d] + dP);    // d / d mlogvar
    }
  }
  float* out = a.part + ((long)(h * PN_SEG + seg) * a.ncomp) * W;
  for (int c = 0; c < a.ncomp; ++c) { out[(long)c * W + d] = acc[c][d]; out[(long)c * W + n + d] = acc[c][n + d]; }
}

// K6: every cell of group h receives d mean_h[comp] / count_h[comp]
__global__ __launch_bounds__(256) void pn_apply_bwd_kernel(PoeN a) {
  const int h = blockIdx.y, n = a.n, W = 2 * n + 1;
  const int b = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  if (b >= a.B[h] || d >= n) return;
  const int c = pn_code(a.comp[h][b], a.ncomp);
  if (c < 0) return;
  const float cnt = a.mean[((long)h * a.ncomp + c) * W + W - 1];
  float sl = 0.f, sv = 0.f;
  for (int seg = 0; seg < PN_SEG; ++seg) {
    const float* p = a.part + ((long)(h * PN_SEG + seg) * a.ncomp + c) * W;
    sl += p[d];
    sv += p[n + d];
  }
  float* ds = a.d_stats[h] + (long)b * a.ld[h];
  ds[d] += sl / cnt;
  ds[n + d] += sv / cnt;
}

}  // namespace spv
