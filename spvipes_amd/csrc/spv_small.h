// Small dense primitives for the [B, <=256] parts of the step (encoder tails, decoder trunk):
// fp32 VALU kernels, LDS tiled, batched over up to SPV_MAXP independent problems per launch
// (the two groups x {private, shared} encoders run as ONE launch per stage).
//
// They replace the stock torch ops of nn/networks.py:120-129 (fc2, mu/logvar heads + BatchNorm1d),
// nn/networks.py:322-323 (sigmoid_decoder trunk) and their autograd: a step needs ~35 of these
// launches instead of ~900 eager kernels.  fp32 throughout: the latent means the north-star
// checks come out of these layers.
#pragma once
#include "../../include/spvipes_hip.h"
#include "spv_common.h"

namespace spv {


typedef spv_linear_prob LinearProb;    // field meanings: include/spvipes_hip.h
typedef spv_linear_batch LinearBatch;

__device__ __forceinline__ unsigned hash_u32(unsigned long long x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdULL; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ULL; x ^= x >> 33;
  return (unsigned)x;
}
// keep-probability test of inverted dropout, counter-based (seed, problem, element) -> reproducible
// (32-bit PCG-style mixing: the 64-bit finaliser above costs ~3 us in the epilogue of the four-encoder fc2 launch, 16 draws per lane)
__device__ __forceinline__ bool dropout_keep(unsigned long long seed, int prob, long idx, float p) {
  const unsigned key = ((unsigned)seed ^ ((unsigned)(seed >> 32) * 0x9E3779B9u)) + (unsigned)(prob + 1) * 0x85EBCA6Bu;
  unsigned x = (unsigned)idx * 747796405u + key;
  x = ((x >> ((x >> 28) + 4u)) ^ x) * 277803737u;
  x ^= x >> 22;
  x = x * 0x2C1B3C6Du + key;
  x ^= x >> 15;
  return (float)(x >> 8) * (1.0f / 16777216.0f) >= p;
}

// ---- small dense layers on the fp32 matrix pipe ------------------------------------------------------
// v_mfma_f32_32x32x2_f32 (exact fp32 products and sums) with both operands read straight from global/L2 into
// registers: no LDS staging, no barriers.  One wave owns a 32 x 32 output tile; lane l supplies row/col l % 32
// and the contraction indices c0 + 8 (l / 32) + j, j < 8, of a 16-deep step (any bijection of the 16 indices
// onto the 8 x 2 MFMA k-slots is valid as long as A and B agree).
__device__ __forceinline__ f16v mfma_f32(float x, float w, const f16v& c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(x, w, c, 0, 0, 0); }

// v[j] = p[j] for j < lim, else 0 (lim may be <= 0).  vec: p is 16-byte aligned and lim is >= 8 or <= 0.
// Branch-free: out-of-range elements are read from `safe` (any valid aligned address) and zeroed afterwards.
// (VEC is a template parameter on purpose: with a run-time flag the two forms sit in the arms of a branch, and hipcc then waits for
// every load INSIDE its arm -- s_waitcnt vmcnt(0) right behind the load -- so that nothing is in flight under the MFMAs of the step:
// tools/probes/linear_probe.py measured 0.5 us per 16-deep step, twice the eight MFMAs' time.)
template <bool VEC>
__device__ __forceinline__ void load8(const float* p, const float* safe, int lim, float (&v)[8]) {
  if constexpr (VEC) {
    const float* s = (lim >= 8) ? p : safe;
    const f4v a = *reinterpret_cast<const f4v*>(s), b = *reinterpret_cast<const f4v*>(s + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) { v[j] = (lim >= 8) ? a[j] : 0.f; v[4 + j] = (lim >= 8) ? b[j] : 0.f; }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float t = *((j < lim) ? p + j : safe); v[j] = (j < lim) ? t : 0.f; }
  }
}
// v[j] = p[(c0 + j) * ld] for c0 + j < lim, else 0: a column walk (coalesced across the lanes of a wave)
__device__ __forceinline__ void load8_strided(const float* p, long ld, int c0, int lim, float (&v)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) { const bool ok = c0 + j < lim; const float t = p[ok ? (long)(c0 + j) * ld : 0]; v[j] = ok ? t : 0.f; }
}
__device__ __forceinline__ bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }
// (Round 4, measured and taken back: FOUR 16-deep steps of operands in flight instead of one -- tools/probes/graph_chain_probe.py prices a
// launch of these kernels at 2.5 us + 0.55 us per step, 2.6 x the eight MFMAs' time, i.e. one L2 round trip per step -- made every shape
// SLOWER (4 x [4096 x 128] x [128 x 128]: 12.1 -> 16.1 us per dependent launch, [128 x 64] x [64 x 64]: 5.8 -> 8.0): hipcc hoists the
// zero-selects of load8 behind the loads, so the first MFMA waits for the whole prologue, and steps beyond K still issue their loads.)

// Y = act(X W^T + b): workgroup = 2 x 2 waves = 64 rows x 64 columns
template <bool VX, bool VW>
__device__ __forceinline__ void linear_fwd_body(const LinearBatch& a) {
  const LinearProb& q = a.p[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
  const int b0 = blockIdx.x * 64 + 32 * (wave >> 1), n0 = blockIdx.y * 64 + 32 * (wave & 1);
  const int rows_out = (q.img_hi != nullptr && q.img_rows > a.B) ? q.img_rows : a.B;   // image rows beyond the batch are zero filled
  if (n0 >= q.N || b0 >= rows_out) return;  // wave-uniform
  const int row = b0 + r, col = n0 + r;
  const float* xr = q.X + (long)min(row, a.B - 1) * q.ldx;   // rows / columns beyond the problem only feed outputs never stored
  const float* wr = q.W + (long)min(col, q.N - 1) * q.K;
  f16v acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float xa[8], wb[8];
  load8<VX>(xr + 8 * h, xr, q.K - 8 * h, xa);
  load8<VW>(wr + 8 * h, wr, q.K - 8 * h, wb);
  for (int k0 = 0; k0 < q.K; k0 += 16) {
    float xn[8], wn[8];
    const int kn = k0 + 16 + 8 * h;   // next step's elements fly under this step's eight MFMAs
    load8<VX>(xr + kn, xr, q.K - kn, xn);
    load8<VW>(wr + kn, wr, q.K - kn, wn);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = mfma_f32(xa[j], wb[j], acc);
#pragma unroll
    for (int j = 0; j < 8; ++j) { xa[j] = xn[j]; wb[j] = wn[j]; }
  }
  if (col >= q.N) return;
  const float bias = q.bias ? q.bias[col] : 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int b = b0 + crow(i, h);
    if (b >= rows_out) continue;
    if (b >= a.B) {   // (only with an image: its padding rows)
      q.img_hi[(long)b * q.ld_img + col] = 0;
      if (q.img_lo != nullptr) q.img_lo[(long)b * q.ld_img + col] = 0;
      continue;
    }
    float v = acc[i] + bias;
    if (a.relu) v = fmaxf(v, 0.f);
    if (a.drop_p > 0.f) {
      const bool keep = q.keep ? (q.keep[(long)b * q.N + col] != 0.f)   // injected mask (nn/networks.py:121 with a recorded draw)
                               : dropout_keep(a.seed + (a.seed_ptr ? *a.seed_ptr : 0ull), blockIdx.z, (long)b * q.N + col, a.drop_p);
      v = keep ? v * (1.0f / (1.0f - a.drop_p)) : 0.f;
    }
    q.Y[(long)b * q.ldy + col] = v;
    if (q.img_hi != nullptr) {   // the same value as split-bf16 words into a packed operand image (spv_linear_prob.img_hi)
      bf16_t hi, lo;
      split_bf16(v, hi, lo);
      q.img_hi[(long)b * q.ld_img + col] = hi;
      if (q.img_lo != nullptr) q.img_lo[(long)b * q.ld_img + col] = lo;
    }
  }
}
__global__ __launch_bounds__(256) void linear_fwd_kernel(LinearBatch a) {
  const LinearProb& q = a.p[blockIdx.z];
  const bool vx = ((q.ldx & 3) == 0) && aligned16(q.X) && ((q.K & 7) == 0);   // (block-uniform)
  const bool vw = aligned16(q.W) && ((q.K & 7) == 0);
  if (vx && vw) linear_fwd_body<true, true>(a);
  else if (vw) linear_fwd_body<false, true>(a);
  else if (vx) linear_fwd_body<true, false>(a);
  else linear_fwd_body<false, false>(a);
}

__device__ __forceinline__ float masked_dy(const LinearBatch& a, const LinearProb& q, int b, int n) {
  float g = q.dY[(long)b * q.lddy + n];
  if (a.relu || a.drop_p > 0.f) g = (q.Y[(long)b * q.ldy + n] > 0.f) ? g * (a.drop_p > 0.f ? 1.0f / (1.0f - a.drop_p) : 1.0f) : 0.f;
  return g;
}
// mask (relu / dropout) of 8 loaded gradient values given the 8 matching forward outputs
__device__ __forceinline__ void apply_mask8(const LinearBatch& a, float (&g)[8], const float (&y)[8]) {
  const float sc = (a.drop_p > 0.f) ? 1.0f / (1.0f - a.drop_p) : 1.0f;
#pragma unroll
  for (int j = 0; j < 8; ++j) g[j] = (y[j] > 0.f) ? g[j] * sc : 0.f;
}

// dX (+)= mask(dY) W : rows = cells, columns = k, contraction over the N outputs of the layer
template <bool VEC, bool MASKED>   // (compile-time for the same reason as load8's VEC: no load may sit in the arm of a run-time branch)
__device__ __forceinline__ void linear_dgrad_body(const LinearBatch& a) {
  const LinearProb& q = a.p[blockIdx.z];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
  const int b0 = blockIdx.x * 64 + 32 * (wave >> 1), k0 = blockIdx.y * 64 + 32 * (wave & 1);
  if (k0 >= q.K || b0 >= a.B) return;
  const int row = b0 + r, col = k0 + r;
  const float* gr = q.dY + (long)min(row, a.B - 1) * q.lddy;
  const float* yr = MASKED ? q.Y + (long)min(row, a.B - 1) * q.ldy : gr;
  const float* wc = q.W + min(col, q.K - 1);
  const float* wc2 = (q.W2 ? q.W2 : q.W) + min(col, q.K - 1);   // rows >= nw2 of the weight come from a second matrix (spv_linear_prob.W2)
  const int nw2 = q.W2 ? q.n_w2 : q.N;
  f16v acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float ga[8], wb[8];
  auto fetch = [&](int n, float (&g)[8], float (&w)[8]) {
    load8<VEC>(gr + n, gr, q.N - n, g);
    if constexpr (MASKED) { float y[8]; load8<VEC>(yr + n, yr, q.N - n, y); apply_mask8(a, g, y); }
#pragma unroll
    for (int j = 0; j < 8; ++j) {   // (column walk over the weight rows n + j; unconditional loads from a selected address)
      const int r = n + j;
      const bool ok = r < q.N;
      const float t = *((r < nw2) ? wc + (long)(ok ? r : 0) * q.K : wc2 + (long)(ok ? r - nw2 : 0) * q.K);
      w[j] = ok ? t : 0.f;
    }
  };
  fetch(8 * h, ga, wb);
  for (int n0 = 0; n0 < q.N; n0 += 16) {
    float gn[8], wn[8];
    fetch(n0 + 16 + 8 * h, gn, wn);
#pragma unroll
    for (int j = 0; j < 8; ++j) acc = mfma_f32(ga[j], wb[j], acc);
#pragma unroll
    for (int j = 0; j < 8; ++j) { ga[j] = gn[j]; wb[j] = wn[j]; }
  }
  if (col >= q.K) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int b = b0 + crow(i, h);
    if (b >= a.B) continue;
    float* dst = q.dX + (long)b * q.lddx + col;
    *dst = a.accumulate ? *dst + acc[i] : acc[i];
  }
}
__global__ __launch_bounds__(256) void linear_dgrad_kernel(LinearBatch a) {
  const LinearProb& q = a.p[blockIdx.z];
  const bool masked = a.relu || a.drop_p > 0.f;
  const bool vec = ((q.lddy & 3) == 0) && aligned16(q.dY) && ((q.N & 7) == 0) && (!masked || (((q.ldy & 3) == 0) && aligned16(q.Y)));
  if (masked) { if (vec) linear_dgrad_body<true, true>(a); else linear_dgrad_body<false, true>(a); }
  else { if (vec) linear_dgrad_body<true, false>(a); else linear_dgrad_body<false, false>(a); }
}

// dW = mask(dY)^T X, db = colsum(mask(dY)): rows = n, columns = k, contraction over the batch.  The batch is cut into
// WG_SLICES slices (one workgroup each; its four waves take a quarter each and are added in wave order); partial
// tiles go to `wpart` and are summed in slice order by linear_wgrad_reduce_kernel.
constexpr int WG_SLICES = 16;

// DIRECT (minibatches of at most WG_DIRECT_MAX_B rows): ONE slice per problem holds every row, and its workgroup writes dW / db itself --
// no partials, no linear_wgrad_reduce_kernel launch (at 128 cells a launch is mostly fixed cost, and 14 of the 16 slices were empty)
constexpr int WG_DIRECT_MAX_B = 256;
template <bool MASKED, bool DIRECT>
__device__ __forceinline__ void linear_wgrad_body(const LinearBatch& a, float* wpart, long prob_stride) {
  constexpr int NSL = DIRECT ? 1 : WG_SLICES;
  const int prob = blockIdx.z / NSL, slice = blockIdx.z % NSL;
  const LinearProb& q = a.p[prob];
  const int n0 = blockIdx.x * 32, k0 = blockIdx.y * 32;
  if (n0 >= q.N || k0 >= q.K) return;  // block-uniform
  __shared__ float s_acc[3][16][64], s_b[3][32];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
  constexpr bool masked = MASKED;
  const int per = ((a.B + NSL * 64 - 1) / (NSL * 64)) * 64;   // rows per slice, a multiple of 64 (16 per wave and step)
  const int bbeg = slice * per + wave * (per / 4), bend = min(bbeg + per / 4, a.B);
  const float* gc = q.dY + min(n0 + r, q.N - 1);
  const float* yc = masked ? q.Y + min(n0 + r, q.N - 1) : gc;
  const float* xc = q.X + min(k0 + r, q.K - 1);
  f16v acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  float bsum = 0.f;
  float ga[8], xb[8];
  auto fetch = [&](int b, float (&g)[8], float (&x)[8]) {
    load8_strided(gc, q.lddy, b, bend, g);
    if constexpr (MASKED) { float y[8]; load8_strided(yc, q.ldy, b, bend, y); apply_mask8(a, g, y); }
    load8_strided(xc, q.ldx, b, bend, x);
  };
  fetch(bbeg + 8 * h, ga, xb);
  for (int b0 = bbeg; b0 < bend; b0 += 16) {
    float gn[8], xn[8];
    fetch(b0 + 16 + 8 * h, gn, xn);
#pragma unroll
    for (int j = 0; j < 8; ++j) { acc = mfma_f32(ga[j], xb[j], acc); bsum += ga[j]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) { ga[j] = gn[j]; xb[j] = xn[j]; }
  }
  bsum += __shfl_xor(bsum, 32, 64);
  if (wave > 0) {
#pragma unroll
    for (int i = 0; i < 16; ++i) s_acc[wave - 1][i][lane] = acc[i];
    if (h == 0) s_b[wave - 1][r] = bsum;
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = ((acc[i] + s_acc[0][i][lane]) + s_acc[1][i][lane]) + s_acc[2][i][lane];
  bsum = ((bsum + s_b[0][r]) + s_b[1][r]) + s_b[2][r];
  const int k = k0 + r;
  if constexpr (DIRECT) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int n = n0 + crow(i, h);
      if (n < q.N && k < q.K) q.dW[(long)n * q.K + k] = acc[i];
    }
    if (blockIdx.y == 0 && h == 0 && n0 + r < q.N && q.db) q.db[n0 + r] = bsum;
    return;
  }
  // partial layout per problem: [slice][N][K + 1] (last column = bias partial)
  float* base = wpart + prob * prob_stride + (long)slice * q.N * (q.K + 1);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int n = n0 + crow(i, h);
    if (n < q.N && k < q.K) base[(long)n * (q.K + 1) + k] = acc[i];
  }
  if (blockIdx.y == 0 && h == 0 && n0 + r < q.N) base[(long)(n0 + r) * (q.K + 1) + q.K] = bsum;
}
__global__ __launch_bounds__(256) void linear_wgrad_kernel(LinearBatch a, float* wpart, long prob_stride) {
  if (a.relu || a.drop_p > 0.f) linear_wgrad_body<true, false>(a, wpart, prob_stride);
  else linear_wgrad_body<false, false>(a, wpart, prob_stride);
}
__global__ __launch_bounds__(256) void linear_wgrad_direct_kernel(LinearBatch a) {
  if (a.relu || a.drop_p > 0.f) linear_wgrad_body<true, true>(a, nullptr, 0);
  else linear_wgrad_body<false, true>(a, nullptr, 0);
}

__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(LinearBatch a, const float* wpart, long prob_stride) {
  const LinearProb& q = a.p[blockIdx.y];
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const long tot = (long)q.N * (q.K + 1);
  if (i >= tot) return;
  const float* src = wpart + blockIdx.y * prob_stride + i;
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < WG_SLICES; ++k) s += src[(long)k * tot];
  const int n = (int)(i / (q.K + 1)), c = (int)(i % (q.K + 1));
  if (c < q.K) q.dW[(long)n * q.K + c] = s;
  else if (q.db) q.db[n] = s;
}

// ---------------------------------------------------------------------------------------------
// BatchNorm1d over the batch dimension, batched problems, N <= 256 columns
// ---------------------------------------------------------------------------------------------
typedef spv_bn_prob BnProb;
typedef spv_bn_batch BnBatch;
constexpr int BN_ROWS = 32;   // rows per workgroup (B = 4096 -> 128 workgroups per problem)

// column c = tid % NC, row slot = tid / NC (NC = N rounded up to a power of two): every thread strides over the
// block's rows; the per-slot partials are then summed in slot order (deterministic)
__device__ __forceinline__ int pow2_at_least(int n) { int p = 1; while (p < n) p <<= 1; return p; }

__global__ __launch_bounds__(256) void bn_stats_kernel(BnBatch a) {
  const BnProb& q = a.p[blockIdx.y];
  __shared__ float s_acc[256], s_mean[256];
  const int NC = pow2_at_least(q.N), RS = 256 / NC;  // N <= 256
  const int c = threadIdx.x % NC, slot = threadIdx.x / NC;
  const int b0 = blockIdx.x * BN_ROWS, b1 = min(b0 + BN_ROWS, a.B);
  float s = 0.f;
  if (c < q.N) {
#pragma unroll 4
    for (int b = b0 + slot; b < b1; b += RS) s += q.X[(long)b * q.ldx + c];
  }
  s_acc[threadIdx.x] = s;
  __syncthreads();
  if (threadIdx.x < q.N) {
    float t = 0.f;
    for (int k = 0; k < RS; ++k) t += s_acc[k * NC + threadIdx.x];
    s_mean[threadIdx.x] = t / (float)(b1 - b0);
  }
  __syncthreads();
  float m2 = 0.f;
  if (c < q.N) {
    const float mean = s_mean[c];
#pragma unroll 4
    for (int b = b0 + slot; b < b1; b += RS) { const float d = q.X[(long)b * q.ldx + c] - mean; m2 += d * d; }
  }
  s_acc[threadIdx.x] = m2;
  __syncthreads();
  if (threadIdx.x < q.N) {
    float t = 0.f;
    for (int k = 0; k < RS; ++k) t += s_acc[k * NC + threadIdx.x];
    q.part[((long)blockIdx.x * q.N + threadIdx.x) * 2] = s_mean[threadIdx.x];
    q.part[((long)blockIdx.x * q.N + threadIdx.x) * 2 + 1] = t;
  }
}

// per column: combine the per-block (mean, M2) partials (Chan's parallel formula, two passes over the partials),
// update the running statistics (torch: UNBIASED variance) and leave (mean, 1/sqrt(var + eps)) in stats.
// block = 64 columns x 4 partial groups; group y takes blocks y, y+4, ... in order, groups are added in order.
__global__ __launch_bounds__(256) void bn_finalize_kernel(BnBatch a) {
  const BnProb& q = a.p[blockIdx.y];
  __shared__ float s_p[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  const bool ok = j < q.N;
  const int nblk = (a.B + BN_ROWS - 1) / BN_ROWS;
  float mean = 0.f, var = 1.f;
  if (a.training) {
    float tot = 0.f;
    if (ok)
#pragma unroll 8
      for (int k = ty; k < nblk; k += 4) tot += q.part[((long)k * q.N + j) * 2] * (float)(min((k + 1) * BN_ROWS, a.B) - k * BN_ROWS);
    s_p[ty][tx] = tot;
    __syncthreads();
    mean = (((s_p[0][tx] + s_p[1][tx]) + s_p[2][tx]) + s_p[3][tx]) / (float)a.B;
    __syncthreads();
    float m2 = 0.f;
    if (ok)
#pragma unroll 8
      for (int k = ty; k < nblk; k += 4) {
        const float cnt = (float)(min((k + 1) * BN_ROWS, a.B) - k * BN_ROWS), d = q.part[((long)k * q.N + j) * 2] - mean;
        m2 += q.part[((long)k * q.N + j) * 2 + 1] + cnt * d * d;
      }
    s_p[ty][tx] = m2;
    __syncthreads();
    var = (((s_p[0][tx] + s_p[1][tx]) + s_p[2][tx]) + s_p[3][tx]) / (float)a.B;
    if (ok && ty == 0) {
      q.running_mean[j] = (1.f - a.momentum) * q.running_mean[j] + a.momentum * mean;
      q.running_var[j] = (1.f - a.momentum) * q.running_var[j] + a.momentum * var * ((float)a.B / fmaxf((float)a.B - 1.f, 1.f));
    }
  } else if (ok) {
    mean = q.running_mean[j];
    var = q.running_var[j];
  }
  if (ok && ty == 0) { q.stats[2 * j] = mean; q.stats[2 * j + 1] = rsqrtf(var + a.eps); }
}

// (mean, 1 / sqrt(var + eps)) of column j of problem q from the per-block partials, by a whole workgroup: the threads with
// `group` in 0..3 sum blocks group, group + 4, ... in order, the four sums are added in order -- the association of
// bn_finalize_kernel, so a launch that folds the finalisation into its consumer (N <= 64: a problem's partials are a few KB) produces
// the same bits.  EVERY thread of the workgroup calls it (three barriers); threads with group < 0 only read the result.  `col` < 64 is
// the LDS column of (q, j) in this workgroup.  `writer`: this thread also updates the running statistics and stores q.stats (one
// thread per column, in exactly one workgroup per problem).
__device__ __forceinline__ void bn_block_stats(const BnBatch& a, const BnProb& q, int j, bool col_ok, int group, int col, bool writer,
                                               float (*s_p)[64], float& mean, float& rstd) {
  const int nblk = (a.B + BN_ROWS - 1) / BN_ROWS;
  float m = 0.f, var = 1.f;
  if (a.training) {   // (uniform)
    float tot = 0.f;
    if (col_ok && group >= 0)
#pragma unroll 8
      for (int k = group; k < nblk; k += 4) tot += q.part[((long)k * q.N + j) * 2] * (float)(min((k + 1) * BN_ROWS, a.B) - k * BN_ROWS);
    if (group >= 0) s_p[group][col] = tot;
    __syncthreads();
    m = (((s_p[0][col] + s_p[1][col]) + s_p[2][col]) + s_p[3][col]) / (float)a.B;
    __syncthreads();
    float m2 = 0.f;
    if (col_ok && group >= 0)
#pragma unroll 8
      for (int k = group; k < nblk; k += 4) {
        const float cnt = (float)(min((k + 1) * BN_ROWS, a.B) - k * BN_ROWS), d = q.part[((long)k * q.N + j) * 2] - m;
        m2 += q.part[((long)k * q.N + j) * 2 + 1] + cnt * d * d;
      }
    if (group >= 0) s_p[group][col] = m2;
    __syncthreads();
    var = (((s_p[0][col] + s_p[1][col]) + s_p[2][col]) + s_p[3][col]) / (float)a.B;
    if (writer && col_ok) {
      q.running_mean[j] = (1.f - a.momentum) * q.running_mean[j] + a.momentum * m;
      q.running_var[j] = (1.f - a.momentum) * q.running_var[j] + a.momentum * var * ((float)a.B / fmaxf((float)a.B - 1.f, 1.f));
    }
  } else if (col_ok) {
    m = q.running_mean[j];
    var = q.running_var[j];
  }
  mean = m;
  rstd = rsqrtf(var + a.eps);
  if (writer && col_ok) { q.stats[2 * j] = mean; q.stats[2 * j + 1] = rstd; }
}

// FIN: the statistics are finalised here from the partials (all problems N <= 64; no bn_finalize_kernel launch before this one)
template <bool FIN>
__device__ __forceinline__ void bn_apply_body(const BnBatch& a, float (*s_p)[64]) {
  const BnProb& q = a.p[blockIdx.y];
  const int rows = (q.img_hi != nullptr && q.img_rows > a.B) ? q.img_rows : a.B;   // image rows beyond the batch are zero filled
  const int b0 = blockIdx.x * BN_ROWS, b1 = min(b0 + BN_ROWS, rows);
  const int NC = pow2_at_least(q.N), RS = 256 / NC;
  const int c = threadIdx.x % NC, slot = threadIdx.x / NC;
  float mean, rstd;
  if constexpr (FIN) {
    if (b0 >= rows) return;   // (workgroup-uniform: nobody is left waiting at the barriers below)
    // NC <= 64, so there are at least four row slots: slots 0..3 are the four partial groups, the others read the result
    bn_block_stats(a, q, c, c < q.N, slot < 4 ? slot : -1, c, blockIdx.x == 0 && slot == 0, s_p, mean, rstd);
    if (c >= q.N) return;
  } else {
    if (c >= q.N || b0 >= rows) return;
    mean = q.stats[2 * c]; rstd = q.stats[2 * c + 1];
  }
  const float sc = rstd * q.gamma[c], be = q.beta[c];
#pragma unroll 4
  for (int b = b0 + slot; b < b1; b += RS) {
    // (the load is unconditional -- rows of the image padding re-read the last batch row -- so that the unrolled iterations' loads
    // are all issued before the first use: see load8)
    float v = (q.X[(long)min(b, a.B - 1) * q.ldx + c] - mean) * sc + be;
    if (a.relu) v = fmaxf(v, 0.f);
    if (b < a.B) q.Y[(long)b * q.ldy + c] = v;
    else v = 0.f;
    if (q.img_hi != nullptr) {
      bf16_t hi, lo;
      split_bf16(v, hi, lo);
      q.img_hi[(long)b * q.ld_img + c] = hi;
      if (q.img_lo != nullptr) q.img_lo[(long)b * q.ld_img + c] = lo;
    }
  }
}
__global__ __launch_bounds__(256) void bn_apply_kernel(BnBatch a) {
  bn_apply_body<false>(a, nullptr);
}
__global__ __launch_bounds__(256) void bn_apply_fin_kernel(BnBatch a) {
  __shared__ float s_p[4][64];
  bn_apply_body<true>(a, s_p);
}

// (RELU / TRAINING are template parameters: a load in the arm of a run-time branch is waited for inside that arm, and the unrolled row
// loops below then pay one dependent L2 round trip per row -- see load8)
template <bool RELU>
__device__ __forceinline__ float bn_masked_dy(const BnProb& q, int b, int j) {
  const float g = q.dY[(long)b * q.lddy + j];
  if constexpr (RELU) return (q.Y[(long)b * q.ldy + j] > 0.f) ? g : 0.f;
  else return g;
}

template <bool RELU>
__device__ __forceinline__ void bn_bwd_reduce_body(const BnBatch& a) {
  const BnProb& q = a.p[blockIdx.y];
  __shared__ float s_g[256], s_gx[256];
  const int NC = pow2_at_least(q.N), RS = 256 / NC;
  const int c = threadIdx.x % NC, slot = threadIdx.x / NC;
  const int b0 = blockIdx.x * BN_ROWS, b1 = min(b0 + BN_ROWS, a.B);
  float sg = 0.f, sgx = 0.f;
  if (c < q.N) {
    const float mean = q.stats[2 * c], inv = q.stats[2 * c + 1];
#pragma unroll 4
    for (int b = b0 + slot; b < b1; b += RS) {
      const float g = bn_masked_dy<RELU>(q, b, c);
      sg += g;
      sgx += g * (q.X[(long)b * q.ldx + c] - mean) * inv;
    }
  }
  s_g[threadIdx.x] = sg; s_gx[threadIdx.x] = sgx;
  __syncthreads();
  if (threadIdx.x < q.N) {
    float t0 = 0.f, t1 = 0.f;
    for (int k = 0; k < RS; ++k) { t0 += s_g[k * NC + threadIdx.x]; t1 += s_gx[k * NC + threadIdx.x]; }
    q.part[((long)blockIdx.x * q.N + threadIdx.x) * 2] = t0;
    q.part[((long)blockIdx.x * q.N + threadIdx.x) * 2 + 1] = t1;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(BnBatch a) {
  if (a.relu) bn_bwd_reduce_body<true>(a); else bn_bwd_reduce_body<false>(a);
}

// d gamma = sum_b g * xhat, d beta = sum_b g: the per-block partials added in block order within four interleaved groups
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(BnBatch a) {
  const BnProb& q = a.p[blockIdx.y];
  __shared__ float s_g[4][64], s_gx[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + tx;
  const int nblk = (a.B + BN_ROWS - 1) / BN_ROWS;
  float sg = 0.f, sgx = 0.f;
  if (j < q.N)
#pragma unroll 8
    for (int k = ty; k < nblk; k += 4) { sg += q.part[((long)k * q.N + j) * 2]; sgx += q.part[((long)k * q.N + j) * 2 + 1]; }
  s_g[ty][tx] = sg; s_gx[ty][tx] = sgx;
  __syncthreads();
  if (ty == 0 && j < q.N) {
    q.dbeta[j] = ((s_g[0][tx] + s_g[1][tx]) + s_g[2][tx]) + s_g[3][tx];
    q.dgamma[j] = ((s_gx[0][tx] + s_gx[1][tx]) + s_gx[2][tx]) + s_gx[3][tx];
  }
}

// FIN: d gamma / d beta are summed here from the per-block partials (all problems N <= 64; no bn_bwd_finalize_kernel launch before this
// one): the first four row slots take blocks slot, slot + 4, ... in order, the four sums are added in order (bn_bwd_finalize_kernel's
// association); workgroup 0 of a problem stores the two parameter gradients.
template <bool RELU, bool TRAINING, bool FIN>
__device__ __forceinline__ void bn_bwd_apply_body(const BnBatch& a, float (*s_g)[64], float (*s_gx)[64]) {
  const BnProb& q = a.p[blockIdx.y];
  const int b0 = blockIdx.x * BN_ROWS, b1 = min(b0 + BN_ROWS, a.B);
  const float invB = 1.0f / (float)a.B;
  const int NC = pow2_at_least(q.N), RS = 256 / NC;
  const int c = threadIdx.x % NC, slot = threadIdx.x / NC;
  float db, dg;
  if constexpr (FIN) {
    if (b0 >= a.B) return;   // (workgroup-uniform)
    const int nblk = (a.B + BN_ROWS - 1) / BN_ROWS;
    float sg = 0.f, sgx = 0.f;
    if (c < q.N && slot < 4)
#pragma unroll 8
      for (int k = slot; k < nblk; k += 4) { sg += q.part[((long)k * q.N + c) * 2]; sgx += q.part[((long)k * q.N + c) * 2 + 1]; }
    if (slot < 4) { s_g[slot][c] = sg; s_gx[slot][c] = sgx; }
    __syncthreads();
    if (c >= q.N) return;
    db = ((s_g[0][c] + s_g[1][c]) + s_g[2][c]) + s_g[3][c];
    dg = ((s_gx[0][c] + s_gx[1][c]) + s_gx[2][c]) + s_gx[3][c];
    if (blockIdx.x == 0 && slot == 0) { q.dbeta[c] = db; q.dgamma[c] = dg; }
  } else {
    if (c >= q.N) return;
    db = q.dbeta[c]; dg = q.dgamma[c];  // written by bn_bwd_finalize_kernel
  }
  const float mean = q.stats[2 * c], inv = q.stats[2 * c + 1], gi = q.gamma[c] * inv;
  const float msg = db * invB, msgx = dg * invB;
#pragma unroll 4
  for (int b = b0 + slot; b < b1; b += RS) {
    const float g = bn_masked_dy<RELU>(q, b, c);
    float dx;
    if constexpr (TRAINING) {
      const float xhat = (q.X[(long)b * q.ldx + c] - mean) * inv;
      dx = gi * (g - msg - xhat * msgx);
    } else {
      dx = gi * g;
    }
    q.dX[(long)b * q.lddx + c] = dx;
  }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(BnBatch a) {
  if (a.relu) { if (a.training) bn_bwd_apply_body<true, true, false>(a, nullptr, nullptr); else bn_bwd_apply_body<true, false, false>(a, nullptr, nullptr); }
  else { if (a.training) bn_bwd_apply_body<false, true, false>(a, nullptr, nullptr); else bn_bwd_apply_body<false, false, false>(a, nullptr, nullptr); }
}
__global__ __launch_bounds__(256) void bn_bwd_apply_fin_kernel(BnBatch a) {
  __shared__ float s_g[4][64], s_gx[4][64];
  if (a.relu) { if (a.training) bn_bwd_apply_body<true, true, true>(a, s_g, s_gx); else bn_bwd_apply_body<true, false, true>(a, s_g, s_gx); }
  else { if (a.training) bn_bwd_apply_body<false, true, true>(a, s_g, s_gx); else bn_bwd_apply_body<false, false, true>(a, s_g, s_gx); }
}

// ---------------------------------------------------------------------------------------------
// Encoder head sampling (nn/networks.py:125-129) + KL to N(0,1) (module/spVIPESmodule.py:841-854), batched
//   post = BatchNorm output [B][2n] = (loc | logvar);  scale = exp(logvar/2);  log_z = loc + scale*eps;
//   theta = softmax(log_z);  kl_b = sum_d 0.5 (scale^2 + loc^2 - 1 - logvar)
// backward: d post from (g_loc, g_logvar, g_scale, g_logz, g_kl) (any may be null)
// ---------------------------------------------------------------------------------------------
typedef spv_sample_prob SampleProb;
typedef spv_sample_batch SampleBatch;

// one lane per latent dimension (n <= 32), 32 lanes per cell, 8 cells per 256-thread block
__device__ __forceinline__ float sum32(float v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float max32(float v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
  return v;
}

__global__ __launch_bounds__(256) void enc_sample_fwd_kernel(SampleBatch a) {
  const SampleProb& q = a.p[blockIdx.y];
  const int b = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  const int n = q.n;
  const bool ok = b < a.B && d < n;
  const long i = (long)b * n + d;
  float loc = 0.f, lv = 0.f, sc = 0.f, z = -INFINITY, klt = 0.f;
  if (ok) {
    loc = q.post[(long)b * 2 * n + d]; lv = q.post[(long)b * 2 * n + n + d];
    sc = expf(0.5f * lv);
    z = loc + sc * q.eps[i];
    klt = 0.5f * (sc * sc + loc * loc - 1.0f - lv);
  }
  const float mx = max32(z);
  const float e = ok ? expf(z - mx) : 0.f;
  const float sum = sum32(e), kl = sum32(klt);
  if (ok) { q.scale[i] = sc; q.logz[i] = z; q.theta[i] = e / sum; }
  if (b < a.B && d == 0) q.kl[b] = kl;
}

__global__ __launch_bounds__(256) void enc_sample_bwd_kernel(SampleBatch a) {
  const SampleProb& q = a.p[blockIdx.y];
  const int b = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  const int n = q.n;
  if (b >= a.B || d >= n) return;
  const long i = (long)b * n + d;
  const float gk = ld_or_zero(q.g_kl, b);
  const float loc = q.post[(long)b * 2 * n + d], sc = q.scale[i];
  const float gz = ld_or_zero(q.g_logz, i);
  const long ig = q.g_ld ? (long)b * q.g_ld + d : i;
  const float gl = ld_or_zero(q.g_loc, ig) + gz + gk * loc;
  const float gs = ld_or_zero(q.g_scale, i) + gz * q.eps[i] + gk * sc;
  const float gv = ld_or_zero(q.g_logvar, ig) + 0.5f * sc * gs - 0.5f * gk;
  q.d_post[(long)b * 2 * n + d] = gl;
  q.d_post[(long)b * 2 * n + n + d] = gv;
}

// ---- encoder heads: BatchNorm (statistics finalised here) + sampling + KL in ONE launch ---------------------------------------------
// bn.p[2e], bn.p[2e + 1] = the BatchNorm problems of encoder e's mu / logvar heads (N = n each), sb.p[e] its sampling problem whose
// `post` is where the two BatchNorm outputs land side by side ([B][2n]).  Workgroup = 32 rows of one encoder; lane = column of
// (loc | logvar), wave w = rows b0 + w, b0 + w + 4, ...  The three kernels this replaces (bn_finalize, bn_apply, enc_sample_fwd) evaluated
// the same expressions in the same order: the outputs are bit-identical to theirs.
__global__ __launch_bounds__(256) void enc_heads_bn_sample_fwd_kernel(BnBatch a, SampleBatch sbat) {
  __shared__ float s_p[4][64];
  const int e = blockIdx.y;
  const SampleProb& sq = sbat.p[e];
  const int n = sq.n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool col_ok = lane < 2 * n;
  const int half = (lane >= n && col_ok) ? 1 : 0, j = col_ok ? lane - half * n : 0;
  const BnProb& q = a.p[2 * e + half];
  const int b0 = blockIdx.x * BN_ROWS;
  if (b0 >= a.B) return;   // (workgroup-uniform)
  float mean, rstd;
  bn_block_stats(a, q, j, col_ok, wave, lane, blockIdx.x == 0 && wave == 0, s_p, mean, rstd);
  const float sc_bn = rstd * q.gamma[j], be = q.beta[j];
  const float* X = q.X + j;
  float* Y = q.Y + j;
  constexpr int RPW = BN_ROWS / 4;   // rows per wave
  float x[RPW], ep[RPW];
#pragma unroll
  for (int r = 0; r < RPW; ++r) {   // all loads first (clamped rows: unconditional)
    const int b = min(b0 + wave + 4 * r, a.B - 1);
    x[r] = X[(long)b * q.ldx];
    ep[r] = sq.eps[(long)b * n + min(lane, n - 1)];
  }
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int b = b0 + wave + 4 * r;
    const bool row_ok = b < a.B;   // (wave-uniform)
    const float y = (x[r] - mean) * sc_bn + be;       // bn_apply_kernel's expression
    if (row_ok && col_ok) Y[(long)b * q.ldy] = y;
    // sampling (enc_sample_fwd_kernel): lane d < n holds loc, lane n + d logvar
    const float lv = __shfl(y, min(lane + n, 63), 64);
    const bool ok = row_ok && lane < n;
    float sc = 0.f, z = -INFINITY, klt = 0.f;
    if (ok) {
      sc = expf(0.5f * lv);
      z = y + sc * ep[r];
      klt = 0.5f * (sc * sc + y * y - 1.0f - lv);
    }
    const float mx = max32(z);
    const float ex = ok ? expf(z - mx) : 0.f;
    const float sum = sum32(ex), kl = sum32(klt);
    if (ok) {
      const long i = (long)b * n + lane;
      sq.scale[i] = sc; sq.logz[i] = z; sq.theta[i] = ex / sum;
    }
    if (row_ok && lane == 0) sq.kl[b] = kl;
  }
}

// backward, first launch: d_post from the upstream gradients (enc_sample_bwd_kernel's expressions) and, from the same registers, the
// per-block partial sums of BatchNorm's backward (sum g, sum g xhat per column).  Same workgroup shape as the forward kernel; the
// four waves' partials are added in wave order.
__global__ __launch_bounds__(256) void enc_heads_bwd_reduce_kernel(BnBatch a, SampleBatch sbat) {
  __shared__ float s_g[4][64], s_gx[4][64];
  const int e = blockIdx.y;
  const SampleProb& sq = sbat.p[e];
  const int n = sq.n;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const bool col_ok = lane < 2 * n;
  const int half = (lane >= n && col_ok) ? 1 : 0, j = col_ok ? lane - half * n : 0;
  const BnProb& q = a.p[2 * e + half];
  const int b0 = blockIdx.x * BN_ROWS;
  if (b0 >= a.B) return;
  const float mean = q.stats[2 * j], inv = q.stats[2 * j + 1];
  constexpr int RPW = BN_ROWS / 4;
  float sg = 0.f, sgx = 0.f;
#pragma unroll
  for (int r = 0; r < RPW; ++r) {
    const int b = b0 + wave + 4 * r;
    const bool row_ok = b < a.B;
    const int bb = min(b, a.B - 1);
    const long i = (long)bb * n + j;
    const float gk = ld_or_zero(sq.g_kl, bb);
    const float loc = sq.post[(long)bb * 2 * n + j], sc = sq.scale[i];
    const float gz = ld_or_zero(sq.g_logz, i);
    const long ig = sq.g_ld ? (long)bb * sq.g_ld + j : i;
    const float gl = ld_or_zero(sq.g_loc, ig) + gz + gk * loc;
    const float gs = ld_or_zero(sq.g_scale, i) + gz * sq.eps[i] + gk * sc;
    const float gv = ld_or_zero(sq.g_logvar, ig) + 0.5f * sc * gs - 0.5f * gk;
    const float g = (row_ok && col_ok) ? (half ? gv : gl) : 0.f;
    const float xv = q.X[(long)bb * q.ldx + j];
    if (row_ok && col_ok) sq.d_post[(long)b * 2 * n + lane] = g;
    sg += g;
    sgx += g * (xv - mean) * inv;
  }
  s_g[wave][lane] = sg; s_gx[wave][lane] = sgx;
  __syncthreads();
  if (wave == 0 && col_ok) {
    q.part[((long)blockIdx.x * q.N + j) * 2] = ((s_g[0][lane] + s_g[1][lane]) + s_g[2][lane]) + s_g[3][lane];
    q.part[((long)blockIdx.x * q.N + j) * 2 + 1] = ((s_gx[0][lane] + s_gx[1][lane]) + s_gx[2][lane]) + s_gx[3][lane];
  }
}

// ---------------------------------------------------------------------------------------------
// Label-based Product of Experts (module/spVIPESmodule.py:583-718 + _poe2 :282-379), on device.
//
// Pairing: cell i of group g with label L and rank k among the same-label cells of its minibatch
// (batch order) is fused with the k-th cell of label L in the other minibatch (mode 0); if L occurs
// there but k >= its count the missing expert is the ones/zeros padding of _poe2 (mode 1); if L does
// not occur there it is the dummy expert loc = 0, logvar = 1 (mode 2).  The reference finds this with
// a Python loop and three .item() syncs per cell (:685-701); here one wave per group walks its
// minibatch in 64-cell chunks with ballots and a running per-label counter in LDS.
// ---------------------------------------------------------------------------------------------
constexpr int POE_LMAX = 1024;   // label codes must be integers in [0, POE_LMAX)
constexpr int POE_WAVES = 8;

// stage 1 (one workgroup of 8 waves per group): stable rank of every cell within its label, label counts and
// first slots, and the cells of each label listed in batch order.  Wave w owns a contiguous segment of the
// minibatch and counts into its own histogram row; an exclusive prefix over the waves makes the ranks global.
constexpr int PR_CH = 8;   // 64-cell chunks of a wave whose labels / ranks stay in registers (B <= 4096: all of them)
__global__ __launch_bounds__(512) void poe_rank_kernel(const float* lab0, const float* lab1, int B0, int B1, int* order0, int* order1,
                                                       int* rank0, int* rank1, int* tables /*[2][2][POE_LMAX]: cnt, start*/, int* err) {
  __shared__ int hist[POE_WAVES][POE_LMAX];
  __shared__ int s_scan[512], s_start[POE_LMAX];
  const int g = blockIdx.x;
  const float* lab = g ? lab1 : lab0;
  const int B = g ? B1 : B0;
  int* order = g ? order1 : order0;
  int* rank = g ? rank1 : rank0;
  int* cnt = tables + (g * 2 + 0) * POE_LMAX;
  int* start = tables + (g * 2 + 1) * POE_LMAX;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const int seg = ((B + POE_WAVES * 64 - 1) / (POE_WAVES * 64)) * 64;
  const int cbeg = w * seg, cend = min(cbeg + seg, B);
  // the first PR_CH chunks' labels are requested up front, all at once (one load per chunk in the loop below is one dependent
  // round trip per chunk), and their ranks stay in registers for the last phase
  int Lc[PR_CH], Rk[PR_CH];
#pragma unroll
  for (int c = 0; c < PR_CH; ++c) {
    const int i = cbeg + 64 * c + lane;
    Lc[c] = (int)lab[min(i, B - 1)];
    Rk[c] = 0;
  }
  for (int l = tid; l < POE_WAVES * POE_LMAX; l += 512) (&hist[0][0])[l] = 0;
  __syncthreads();
  auto rank_chunk = [&](int i, bool act, int L) {   // stable rank of the chunk's cells within their labels, wave-local running counts in hist[w]
    if (act && (L < 0 || L >= POE_LMAX)) { *err = 1; L = 0; }
    if (!act) L = -1;
    unsigned long long todo = __ballot(act);
    int my_rank = 0;
    while (todo) {
      const int src = __ffsll((long long)todo) - 1;
      const int cur = __shfl(L, src, 64);
      const unsigned long long same = __ballot(act && L == cur);
      const int base = hist[w][cur];
      if (act && L == cur) my_rank = base + __popcll(same & ((1ull << lane) - 1ull));
      if (lane == src) hist[w][cur] = base + __popcll(same);
      todo &= ~same;
    }
    return my_rank;
  };
#pragma unroll
  for (int c = 0; c < PR_CH; ++c) {
    const int i = cbeg + 64 * c + lane;
    if (cbeg + 64 * c < cend) Rk[c] = rank_chunk(i, i < cend, Lc[c]);   // (wave-uniform test)
  }
  for (int c0 = cbeg + 64 * PR_CH; c0 < cend; c0 += 64) {   // B > 4096: the remaining chunks through memory
    const int i = c0 + lane;
    const bool act = i < cend;
    const int r = rank_chunk(i, act, act ? (int)lab[i] : -1);
    if (act) rank[i] = r;
  }
  __syncthreads();
  // exclusive prefix over the waves (per label) and the label totals; two labels per thread
  int tot[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int l = 2 * tid + k;
    int run = 0;
    for (int ww = 0; ww < POE_WAVES; ++ww) { const int t = hist[ww][l]; hist[ww][l] = run; run += t; }
    tot[k] = run;
    cnt[l] = run;
  }
  // exclusive scan of the totals over the 1024 labels
  s_scan[tid] = tot[0] + tot[1];
  __syncthreads();
  for (int off = 1; off < 512; off <<= 1) {
    const int v = (tid >= off) ? s_scan[tid - off] : 0;
    __syncthreads();
    s_scan[tid] += v;
    __syncthreads();
  }
  const int excl = s_scan[tid] - (tot[0] + tot[1]);
  start[2 * tid] = excl;
  start[2 * tid + 1] = excl + tot[0];
  s_start[2 * tid] = excl;
  s_start[2 * tid + 1] = excl + tot[0];
  __syncthreads();
#pragma unroll
  for (int c = 0; c < PR_CH; ++c) {
    const int i = cbeg + 64 * c + lane;
    if (i < cend) {
      const int L = min(max(Lc[c], 0), POE_LMAX - 1);
      const int r = hist[w][L] + Rk[c];
      rank[i] = r;
      order[s_start[L] + r] = i;
    }
  }
  for (int c0 = cbeg + 64 * PR_CH; c0 < cend; c0 += 64) {
    const int i = c0 + lane;
    if (i < cend) {
      const int L = min(max((int)lab[i], 0), POE_LMAX - 1);
      const int r = hist[w][L] + rank[i];
      rank[i] = r;
      order[s_start[L] + r] = i;
    }
  }
}

// stage 2: partner / mode of every cell from the other group's tables
__global__ __launch_bounds__(256) void poe_lookup_kernel(const float* lab0, const float* lab1, int B0, int B1, const int* order0,
                                                         const int* order1, const int* rank0, const int* rank1, const int* tables,
                                                         int* partner0, int* mode0, int* partner1, int* mode1) {
  const int g = blockIdx.y, o = 1 - g;
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= (g ? B1 : B0)) return;
  const int L = min(max((int)(g ? lab1 : lab0)[i], 0), POE_LMAX - 1), k = (g ? rank1 : rank0)[i];
  const int co = tables[(o * 2 + 0) * POE_LMAX + L], so = tables[(o * 2 + 1) * POE_LMAX + L];
  const int m = (k < co) ? 0 : (co > 0 ? 1 : 2);
  (g ? mode1 : mode0)[i] = m;
  (g ? partner1 : partner0)[i] = (m == 0) ? (o ? order1 : order0)[so + k] : -1;
}

typedef spv_poe_args PoeArgs;

// One fusion kernel serves the three pairings of the reference:
//   label PoE   (:583-718): self expert = the cell's own encoder statistics, partner / mode from poe_lookup_kernel
//   paired PoE  (:511-571): same, every cell has a partner (mode 0) = arg max of its transport-plan row / column;
//                           clamp_scale: the Normal the draw and the KL use has scale.clamp(min = 1e-6)
//   cluster PoE (:184-280): self / partner experts = plan-weighted averages (expert[g], see plan_expert_*), the
//                           _poe2 padding for mode 1, and with lone_passthrough a component absent from the other
//                           minibatch (mode 2) keeps its own encoder statistics
__global__ __launch_bounds__(256) void poe_fuse_fwd_kernel(PoeArgs a) {
  const int g = blockIdx.y, o = 1 - g;
  const int b = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  const int n = a.n;
  const bool ok = b < a.B[g] && d < n;
  float z = -INFINITY, klt = 0.f, jl = 0.f, jv = 0.f, sc = 0.f;
  const long i = (long)b * n + d;
  if (ok) {
    int m, pr;
    if (a.lab[0] != nullptr) {   // (uniform) label PoE: partner / mode from the other group's tables, as poe_lookup_kernel derives them
      const int L = min(max((int)a.lab[g][b], 0), POE_LMAX - 1), k = a.rank[g][b];
      const int co = a.tables[(o * 2 + 0) * POE_LMAX + L], so = a.tables[(o * 2 + 1) * POE_LMAX + L];
      m = (k < co) ? 0 : (co > 0 ? 1 : 2);
      pr = (m == 0) ? a.order[o][so + k] : -1;
      if (d == 0) { const_cast<int*>(a.mode[g])[b] = m; const_cast<int*>(a.partner[g])[b] = pr; }   // kept for the backward pass
    } else {
      m = a.mode[g][b]; pr = a.partner[g][b];
    }
    const float* own = a.stats[g] + (long)b * a.ld[g];           // (loc | logvar) rows of the shared encoder
    const float* self = a.expert[g] ? a.expert[g] + (long)b * a.ld_expert[g] : own;
    const float* oth = a.expert[o] ? a.expert[o] + (long)(pr < 0 ? 0 : pr) * a.ld_expert[o] : a.stats[o] + (long)(pr < 0 ? 0 : pr) * a.ld[o];
    if (m == 2 && a.lone_passthrough) {
      jl = own[d]; jv = own[n + d]; sc = expf(0.5f * jv);
    } else {
      const float loc = self[d], inv = expf(-self[n + d]);
      float t, u;
      if (m == 0) { t = expf(-oth[n + d]); u = oth[d] * t; }
      else { t = (m == 1) ? 1.0f : 0.36787944117144233f; u = 0.f; }
      const float J = 1.0f / (1.0f + (inv + t));
      jl = (loc * inv + u) * J; jv = logf(J); sc = sqrtf(expf(jv));
    }
    const float sq = a.clamp_scale ? fmaxf(sc, 1e-6f) : sc;
    z = jl + sq * a.eps[g][i];
    klt = 0.5f * (sq * sq + jl * jl - 1.0f - logf(sq * sq));
  }
  const float mx = max32(z);
  const float e = ok ? expf(z - mx) : 0.f;
  const float sum = sum32(e), kl = sum32(klt);
  if (ok) { a.loc[g][i] = jl; a.logvar[g][i] = jv; a.scale[g][i] = sc; a.logz[g][i] = z; a.theta[g][i] = e / sum; }
  if (b < a.B[g] && d == 0) a.kl[g][b] = kl;
}

// d_stats[g] / d_expert[g] (same layouts as stats / expert, zeroed by spv_poe_fuse_bwd before the launch) receive the
// gradient of a cell's self expert and, through its partner, of the other group's expert.  Label PoE: at most two adds
// per element (order-independent); paired PoE: a cell can be the arg max of several rows, so its adds commute only up
// to fp32 rounding (the same holds for the index_add of the reference's autograd).
// AGG (paired / cluster PoE: partners are arg maxima and may coincide -- with a sparse plan most rows of a minibatch block are empty
// and their arg max is cell 0, so thousands of cells add into ONE row; 136 us at C5's shard shape with one atomic per cell and
// column): a workgroup takes 32 cells, the contributions of cells that share a partner are summed in the workgroup (cell order) and the
// first of them issues one atomic per column.  Label PoE pairs cells one to one: no aggregation, 8 cells per workgroup.
template <bool AGG>
__global__ __launch_bounds__(AGG ? 1024 : 256) void poe_fuse_bwd_kernel(PoeArgs a) {
  constexpr int CPB = AGG ? 32 : 8;
  __shared__ float s_c[AGG ? 32 : 1][64];
  __shared__ int s_pr[AGG ? 32 : 1];
  const int g = blockIdx.y, o = 1 - g;
  const int cell = threadIdx.x >> 5, d = threadIdx.x & 31;
  const int b = blockIdx.x * CPB + cell;
  const int n = a.n;
  const bool ok = b < a.B[g] && d < n;
  float c0 = 0.f, c1 = 0.f;
  int prt = -1;
  if (ok) {
    const int m = a.mode[g][b], pr = a.partner[g][b];
    const long prc = pr < 0 ? 0 : pr;
    const float* own = a.stats[g] + (long)b * a.ld[g];
    const float* self = a.expert[g] ? a.expert[g] + (long)b * a.ld_expert[g] : own;
    const float* oth = a.expert[o] ? a.expert[o] + prc * a.ld_expert[o] : a.stats[o] + prc * a.ld[o];
    float* d_own = a.d_stats[g] + (long)b * a.ld[g];
    float* d_self = a.expert[g] ? a.d_expert[g] + (long)b * a.ld_expert[g] : d_own;
    const float gk = ld_or_zero(a.g_kl[g], b);
    const long i = (long)b * n + d;
    const float jl = a.loc[g][i], sc = a.scale[g][i];
    const bool live = !a.clamp_scale || sc >= 1e-6f;   // below the clamp the draw and the KL do not depend on the scale
    const float sq = a.clamp_scale ? fmaxf(sc, 1e-6f) : sc;
    const float gz = ld_or_zero(a.g_logz[g], i);
    const float Gl = ld_or_zero(a.g_loc[g], i) + gz + gk * jl;
    const float Gs = ld_or_zero(a.g_scale[g], i) + (live ? gz * a.eps[g][i] + gk * (sq - 1.0f / sq) : 0.f);
    const float Gv = ld_or_zero(a.g_logvar[g], i) + 0.5f * sc * Gs;
    if (m == 2 && a.lone_passthrough) {  // loc* = loc, logvar* = logvar, scale* = exp(logvar / 2)
      atomicAdd(&d_own[d], Gl);
      atomicAdd(&d_own[n + d], Gv);
    } else {
      const float loc = self[d], inv = expf(-self[n + d]);
      float t = (m == 1) ? 1.0f : 0.36787944117144233f, lo = 0.f;
      if (m == 0) { lo = oth[d]; t = expf(-oth[n + d]); }
      const float J = 1.0f / (1.0f + (inv + t));
      const float dN = Gl * J, dP = -J * (Gl * jl + Gv);
      atomicAdd(&d_self[d], dN * inv);
      atomicAdd(&d_self[n + d], -inv * (dN * loc + dP));
      if (m == 0) { c0 = dN * t; c1 = -t * (dP + dN * lo); prt = pr; }
    }
  }
  if constexpr (AGG) {
    s_c[cell][d] = c0; s_c[cell][32 + d] = c1;
    if (d == 0) s_pr[cell] = prt;   // (lane 0 of a cell is in range whenever the cell is: n >= 1)
    __syncthreads();
    if (prt < 0) return;
    for (int c = 0; c < cell; ++c) if (s_pr[c] == prt) return;   // an earlier cell of the workgroup carries this partner's sum
    for (int c = cell + 1; c < CPB; ++c)
      if (s_pr[c] == prt) { c0 += s_c[c][d]; c1 += s_c[c][32 + d]; }
  } else {
    if (prt < 0) return;
  }
  float* d_oth = a.expert[o] ? a.d_expert[o] + (long)prt * a.ld_expert[o] : a.d_stats[o] + (long)prt * a.ld[o];
  atomicAdd(&d_oth[d], c0);
  atomicAdd(&d_oth[n + d], c1);
}

// ---------------------------------------------------------------------------------------------
// Transport plans kept sparse on device.  The reference stores the plan dense, gathers the
// [B0, B1] block of a minibatch pair (module/spVIPESmodule.py:474-482) and takes arg max over its rows and columns
// (:520-523) or masked row-normalised products with it (:213-229).  Here the plan is CSR (rows = cells of group 0)
// plus its transpose (CSR of the plan^T: rows = cells of group 1); a step touches only the stored entries of the
// minibatch's rows.  inv*[c] = position of dataset cell c in the current minibatch, -1 if absent.
// Plan values are transport masses (>= 0); positions that are not stored count as 0.
// ---------------------------------------------------------------------------------------------
__global__ void plan_invmap_kernel(const int* idx0, int B0, const int* idx1, int B1, int* inv0, int n0, int* inv1, int n1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  const int* idx = g ? idx1 : idx0;
  int* inv = g ? inv1 : inv0;
  const int B = g ? B1 : B0, nn = g ? n1 : n0;
  if (i < B) { const int c = idx[i]; if (c >= 0 && c < nn) inv[c] = i; }
}

// partner[i] = arg max_j block[i][j] with block[i][j] = plan[idx_self[i]][idx_other[j]]: first maximum, like torch.argmax
// (an all-zero row gives 0).  One thread per cell; a row holds ~10 stored entries.
__global__ void plan_argmax_kernel(const int* ptr0, const int* ind0, const float* val0, const int* ptr1, const int* ind1, const float* val1,
                                   const int* idx0, const int* idx1, const int* inv0, const int* inv1, int B0, int B1, int n0, int n1,
                                   int* partner0, int* partner1) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= (g ? B1 : B0)) return;
  const int* ptr = g ? ptr1 : ptr0; const int* ind = g ? ind1 : ind0; const float* val = g ? val1 : val0;
  const int* inv_o = g ? inv0 : inv1;
  const int r = (g ? idx1 : idx0)[i];
  float best = 0.f; int bj = 0; bool any = false;
  if (r >= 0 && r < (g ? n1 : n0)) {
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) {
      const int j = inv_o[ind[k]];
      if (j < 0) continue;
      const float v = val[k];
      if (v > 0.f && (!any || v > best || (v == best && j < bj))) { best = v; bj = j; any = true; }
    }
  }
  (g ? partner1 : partner0)[i] = any ? bj : 0;
}

// Cluster PoE experts (module/spVIPESmodule.py:213-229): for cell i of group g in component c,
//   E_g[i] = sum_j w'_ij * stats_g[j],  j over the other minibatch's positions in component c,  w'_ij = w_ij / max(sum_j w_ij, 1e-10)
// with w_ij the plan entry between the two cells -- the reference multiplies group g's OWN statistics by weights indexed with
// the other group's positions (both minibatches have B rows), and so does this.  32 lanes per cell (one per latent
// dimension, loc and logvar together), a serial walk over the row's stored entries.
typedef spv_plan_expert_args ExpertArgs;

__global__ __launch_bounds__(256) void plan_expert_fwd_kernel(ExpertArgs a) {
  const int g = blockIdx.y, o = 1 - g;
  const int i = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  if (i >= a.B) return;
  const int* ptr = g ? a.plan.ptr1 : a.plan.ptr0; const int* ind = g ? a.plan.ind1 : a.plan.ind0; const float* val = g ? a.plan.val1 : a.plan.val0;
  const int nrows = g ? a.plan.n1 : a.plan.n0;
  const int r = a.idx[g][i];
  const float ci = a.comp[g][i];
  const bool on = d < a.n;
  float al = 0.f, av = 0.f, rs = 0.f;
  if (r >= 0 && r < nrows) {
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) {
      const int j = a.inv[o][ind[k]];
      if (j < 0 || a.comp[o][j] != ci) continue;
      const float w = val[k];
      rs += w;
      if (on) { const float* sj = a.stats[g] + (long)j * a.ld[g]; al += w * sj[d]; av += w * sj[a.n + d]; }
    }
  }
  const float rc = fmaxf(rs, 1e-10f);
  if (on) { float* e = a.expert[g] + (long)i * a.ld_expert[g]; e[d] = al / rc; e[a.n + d] = av / rc; }
  if (d == 0) a.rowsum[g][i] = rc;
}

// d stats_g[t] += sum_i w'_it * dE_g[i]: a gather over the transposed structure (the stored entries of the OTHER group's
// dataset cell at position t), so every element is written by one lane and the sum order is fixed.
__global__ __launch_bounds__(256) void plan_expert_bwd_kernel(ExpertArgs a) {
  const int g = blockIdx.y, o = 1 - g;
  const int t = blockIdx.x * 8 + (threadIdx.x >> 5), d = threadIdx.x & 31;
  if (t >= a.B || d >= a.n) return;
  // rows of the transposed-for-g structure are the other group's cells: g = 0 walks plan^T, g = 1 walks the plan
  const int* ptr = g ? a.plan.ptr0 : a.plan.ptr1; const int* ind = g ? a.plan.ind0 : a.plan.ind1; const float* val = g ? a.plan.val0 : a.plan.val1;
  const int nrows = g ? a.plan.n0 : a.plan.n1;
  const int r = a.idx[o][t];
  const float ct = a.comp[o][t];
  float gl = 0.f, gv = 0.f;
  if (r >= 0 && r < nrows) {
    for (int k = ptr[r]; k < ptr[r + 1]; ++k) {
      const int i = a.inv[g][ind[k]];
      if (i < 0 || a.comp[g][i] != ct) continue;
      const float w = val[k] / a.rowsum[g][i];
      const float* de = a.d_expert[g] + (long)i * a.ld_expert[g];
      gl += w * de[d]; gv += w * de[a.n + d];
    }
  }
  float* ds = a.d_stats[g] + (long)t * a.ld[g];
  ds[d] += gl; ds[a.n + d] += gv;
}

// ---------------------------------------------------------------------------------------------
// Decoder preparation (module/spVIPESmodule.py:733-754 + the BatchNorm of the two factor regressors,
// nn/networks.py:314,318 with scvi FCLayers' BatchNorm1d(eps 1e-3, momentum 0.01)).
// ---------------------------------------------------------------------------------------------
typedef spv_zsplit_args ZsplitArgs;
constexpr int DEC_PACK_KP = 16, DEC_PACK_KPS = 48;   // = SPV_DEC_KP, SPV_DEC_KP + SPV_DEC_KS (layout of the regressor operand image)

// Z = cat(private_log_z, poe_log_z); z_private = Z[:, n_s:n_s+n_p], z_shared = Z[:, :n_s]  (the A6 quirk)
// zcat = [z_private | z_shared] (input of the mixing trunk, nn/networks.py:322)
__global__ __launch_bounds__(256) void zsplit_fwd_kernel(ZsplitArgs a) {
  const int g = blockIdx.y;
  const int nt = a.n_p + a.n_s;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)a.B * nt) return;
  const int b = (int)(i / nt), c = (int)(i % nt);  // column of zcat
  const int zc = (c < a.n_p) ? a.n_s + c : c - a.n_p;  // column of Z
  const float v = (zc < a.n_p) ? a.priv[g][(long)b * a.n_p + zc] : a.poe[g][(long)b * a.n_s + zc - a.n_p];
  a.zcat[g][i] = v;
}
// column c of zcat = [z_private | z_shared] for cell b, straight from the two latents (the A6 quirk above)
__device__ __forceinline__ float zcat_value(const ZsplitArgs& a, int g, int b, int c) {
  const int zc = (c < a.n_p) ? a.n_s + c : c - a.n_p;   // column of Z = cat(private_log_z, poe_log_z)
  const float pv = a.priv[g][(long)b * a.n_p + min(zc, a.n_p - 1)], qv = a.poe[g][(long)b * a.n_s + max(zc - a.n_p, 0)];   // (both loads unconditional)
  return (zc < a.n_p) ? pv : qv;
}
// zcat AND the decoder's packed bf16 operand images of the same latents (see spv_zsplit_args) in one pass: one thread per image
// element; the threads of the mixture-operand tail also store zcat (its columns are the first n_p + n_s of that tail)
__global__ __launch_bounds__(256) void zsplit_pack_kernel(ZsplitArgs a) {
  const int g = blockIdx.y;
  const int nt = a.n_p + a.n_s;
  const int wcols = DEC_PACK_KPS + a.am_cols;              // [ aps: 48 | am tail: am_cols ]
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)a.Bp * wcols) return;
  const int b = (int)(i / wcols), c = (int)(i % wcols);
  const int bb = min(b, a.B - 1);
  float v = 0.f;
  bf16_t* dh; bf16_t* dl; long o;
  if (c < DEC_PACK_KPS) {
    const int cs = (c < DEC_PACK_KP) ? c : c - DEC_PACK_KP, n = (c < DEC_PACK_KP) ? a.n_p : a.n_s, zoff = (c < DEC_PACK_KP) ? 0 : a.n_p;
    const float z = zcat_value(a, g, bb, zoff + min(cs, n - 1));
    if (b < a.B) v = (cs < n) ? z : (cs == n ? 1.f : 0.f);
    dh = a.aps_hi[g]; dl = a.aps_lo[g]; o = (long)b * DEC_PACK_KPS + c;
  } else {
    const int cm = c - DEC_PACK_KPS;
    const float z = zcat_value(a, g, bb, min(cm, nt - 1));
    if (b < a.B) {
      v = (cm < nt) ? z : (cm == nt ? 1.f : 0.f);
      if (cm < nt) a.zcat[g][(long)b * nt + cm] = z;
    }
    dh = a.am_hi[g]; dl = a.am_lo[g]; o = (long)b * a.ld_am + a.am_col + cm;
  }
  if (dh == nullptr) return;
  bf16_t hi, lo;
  split_bf16(v, hi, lo);
  dh[o] = hi;
  if (dl != nullptr) dl[o] = lo;
}
__global__ __launch_bounds__(256) void zsplit_bwd_kernel(ZsplitArgs a) {
  const int g = blockIdx.y;
  const int nt = a.n_p + a.n_s;
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)a.B * nt) return;
  const int b = (int)(i / nt), zc = (int)(i % nt);  // column of Z
  const int c = (zc < a.n_s) ? a.n_p + zc : zc - a.n_s;  // column of zcat holding it (Z is n_p + n_s wide: each column used once)
  const float v = a.d_zcat[g][(long)b * nt + c];
  if (zc < a.n_p) a.d_priv[g][(long)b * a.n_p + zc] = v;
  else a.d_poe[g][(long)b * a.n_s + zc - a.n_p] = v;
}

typedef spv_fold_prob FoldProb;
typedef spv_fold_batch FoldBatch;
constexpr int FOLD_KMAX = 32;

// per gene: BatchNorm over the batch of the linear map z -> w_g . z folded into W'_g = inv * w_g and
// c_g = beta_g - inv * mean_g, with mean_g = w_g . zbar, var_g = w_g^T C w_g (biased), inv = gamma_g / sqrt(var_g + eps);
// written straight into the packed bf16 hi/lo operand image.  zc[0..K) = column sums of z, zc[K..) = z^T z.
// Four lanes per gene (FOLD_FWD_GENES genes per workgroup): lane `sub` takes rows sub, sub + 4, ... of C and the 8-element chunks sub, sub + 4, ...
// of the image row; mean and variance are summed over the four lanes.  (One lane per gene was 40 workgroups per problem on 256 CUs, each
// lane a K x K chain.)
constexpr int FOLD_FWD_GENES = 64;
__global__ __launch_bounds__(256) void bn_fold_fwd_kernel(FoldBatch a) {
  const FoldProb& q = a.p[blockIdx.y];
  __shared__ float s_zbar[FOLD_KMAX], s_C[FOLD_KMAX * FOLD_KMAX];
  const int K = q.K;
  if (blockIdx.x * FOLD_FWD_GENES >= q.Gp) return;   // (the grid is sized for the largest problem)
  if (a.training) {
    const float invB = 1.0f / (float)a.B;
    for (int i = threadIdx.x; i < K; i += 256) s_zbar[i] = q.zsum[i] * invB;
    for (int i = threadIdx.x; i < K * K; i += 256) s_C[i] = q.zz[i] * invB - (q.zsum[i / K] * invB) * (q.zsum[i % K] * invB);
    __syncthreads();
  }
  const int sub = threadIdx.x & 3;
  const int g = blockIdx.x * FOLD_FWD_GENES + (threadIdx.x >> 2);
  const bool in_img = g < q.Gp, real = g < q.G;
  const int gi = min(g, q.G - 1);                    // padding rows compute on the last gene and store zeros: no lane leaves before the shuffles
  const float* w = q.W + (long)gi * K;               // K <= 32 floats, L1-resident: re-read instead of a runtime-indexed register array
  float mean, var;
  if (a.training) {
    float pm = 0.f, pv = 0.f;
    for (int k = sub; k < K; k += 4) pm += w[k] * s_zbar[k];
    for (int k0 = sub; k0 < K; k0 += 16) {   // this lane's rows k0, k0 + 4, k0 + 8, k0 + 12: independent chains, one read of w[l] for the four
      float v[4] = {0.f, 0.f, 0.f, 0.f};
      int row[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) row[u] = min(k0 + 4 * u, K - 1) * K;
      for (int l = 0; l < K; ++l) {
        const float wl = w[l];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] += s_C[row[u] + l] * wl;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) pv += (k0 + 4 * u < K) ? w[k0 + 4 * u] * v[u] : 0.f;
    }
    pm += __shfl_xor(pm, 1); pm += __shfl_xor(pm, 2);   // (the same bits on the four lanes: the additions commute)
    pv += __shfl_xor(pv, 1); pv += __shfl_xor(pv, 2);
    mean = pm;
    var = fmaxf(pv, 0.f);
    if (real && sub == 0) {
      q.running_mean[g] = (1.f - a.momentum) * q.running_mean[g] + a.momentum * mean;
      q.running_var[g] = (1.f - a.momentum) * q.running_var[g] + a.momentum * var * ((float)a.B / fmaxf((float)a.B - 1.f, 1.f));
    }
  } else {
    mean = q.running_mean[gi]; var = q.running_var[gi];
  }
  const float inv = q.gamma[gi] * rsqrtf(var + a.eps);
  const float cg = q.beta[gi] - mean * inv;
  if (real && sub == 0) { q.stat[2 * g] = mean; q.stat[2 * g + 1] = var; }
  if (!in_img) return;
  bf16_t* hi = q.img_hi + (long)g * q.ld_img + q.col_off;
  bf16_t* lo = q.img_lo + (long)g * q.ld_img + q.col_off;
  for (int k8 = 8 * sub; k8 < q.slot; k8 += 32) {   // slot, ld_img and col_off are multiples of 8 elements: one 16-byte store per image
    unsigned hw[4], lw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      bf16_t h0, l0, h1, l1;
      const int ka = k8 + 2 * j, kb = ka + 1;
      const float va = (ka < K) ? w[min(ka, K - 1)] * inv : (ka == K ? cg : 0.f), vb = (kb < K) ? w[min(kb, K - 1)] * inv : (kb == K ? cg : 0.f);
      split_bf16(real ? va : 0.f, h0, l0);
      split_bf16(real ? vb : 0.f, h1, l1);
      hw[j] = (unsigned)h0 | ((unsigned)h1 << 16);
      lw[j] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
    *reinterpret_cast<u4v*>(hi + k8) = u4v{hw[0], hw[1], hw[2], hw[3]};
    *reinterpret_cast<u4v*>(lo + k8) = u4v{lw[0], lw[1], lw[2], lw[3]};
  }
}

// backward of the fold: from dW' [G][ld_dw] (columns 0..K-1 = d W'_g, column K = d c_g) to dW, d gamma, d beta and
// per-block partial sums of d zbar [K] and d C [K][K]
// 256 genes per workgroup (the unit of the block partials), four lanes per gene as in the forward kernel: 1024 threads
constexpr int FOLD_BWD_THREADS = 1024;
__global__ __launch_bounds__(FOLD_BWD_THREADS) void bn_fold_bwd_kernel(FoldBatch a) {
  const FoldProb& q = a.p[blockIdx.y];
  __shared__ float s_zbar[FOLD_KMAX], s_C[FOLD_KMAX * FOLD_KMAX], s_w[256 * FOLD_KMAX], s_dm[256], s_dv[256], s_acc[3][17][64];
  const int K = q.K;
  const int nred = K + K * K;
  if (blockIdx.x * 256 >= q.G) return;  // (the grid is sized for the largest problem)
  const int gl = threadIdx.x >> 2, sub = threadIdx.x & 3;   // gene of the block, lane of the gene
  const int g = blockIdx.x * 256 + gl;
  const bool real = g < q.G;
  const int gi = min(g, q.G - 1);   // lanes past the last gene compute on it and store nothing: no lane leaves before the shuffles
  float* w = s_w + gl * FOLD_KMAX;  // this gene's weight row lives in LDS
  for (int k = sub; k < K; k += 4) w[k] = real ? q.W[(long)g * K + k] : 0.f;
  if (a.training) {
    const float invB = 1.0f / (float)a.B;
    for (int i = threadIdx.x; i < K; i += FOLD_BWD_THREADS) s_zbar[i] = q.zsum[i] * invB;
    for (int i = threadIdx.x; i < K * K; i += FOLD_BWD_THREADS) s_C[i] = q.zz[i] * invB - (q.zsum[i / K] * invB) * (q.zsum[i % K] * invB);
  }
  __syncthreads();
  float dmean = 0.f, dvar = 0.f;
  {
    const float mean = q.stat[2 * gi], var = q.stat[2 * gi + 1];
    const float rs = rsqrtf(var + a.eps), gam = q.gamma[gi], inv = gam * rs;
    const float* gW = q.dWeff + (long)gi * q.ld_dw;
    const float gc = gW[K];
    float dinv = 0.f;
    for (int k = sub; k < K; k += 4) dinv += gW[k] * w[k];
    dinv += __shfl_xor(dinv, 1); dinv += __shfl_xor(dinv, 2);   // (the same bits on the four lanes: the additions commute)
    dinv -= gc * mean;
    if (real && sub == 0) {
      q.dbeta[g] = gc;
      q.dgamma[g] = dinv * rs;
    }
    if (a.training) {
      if (real) {
        dmean = -gc * inv;
        dvar = dinv * gam * (-0.5f) * rs * rs * rs;
      }
      for (int k0 = sub; k0 < K; k0 += 16) {   // this lane's rows k0, k0 + 4, k0 + 8, k0 + 12 of C (independent chains)
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        int row[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) row[u] = min(k0 + 4 * u, K - 1) * K;
        for (int l = 0; l < K; ++l) {
          const float wl = w[l];
#pragma unroll
          for (int u = 0; u < 4; ++u) v[u] += s_C[row[u] + l] * wl;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int k = k0 + 4 * u;
          if (real && k < K) q.dW[(long)g * K + k] = gW[k] * inv + dmean * s_zbar[k] + 2.0f * dvar * v[u];
        }
      }
    } else if (real) {
      for (int k = sub; k < K; k += 4) q.dW[(long)g * K + k] = gW[k] * inv;
    }
  }
  if (!a.training) return;
  // block partials of d zbar[k] = sum_g dmean_g w_gk and d C[k][l] = sum_g dvar_g w_gk w_gl, summed in gene order
  if (sub == 0) { s_dm[gl] = dmean; s_dv[gl] = dvar; }
  __syncthreads();
  // One exact-fp32 MFMA tile per wave (v_mfma_f32_32x32x2_f32, two genes per step): dC = (dv . W)^T W is a
  // [K x 256] x [256 x K] product and d zbar rides along as row 0 of a second tile; the four waves' tiles are added in
  // wave order.  (The scalar form -- every thread a 256-term chain of LDS reads per output -- made this the longest
  // kernel of the small-kernel backward chain.)
  {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31;
    f16v accC;
#pragma unroll
    for (int i = 0; i < 16; ++i) accC[i] = 0.f;
    f16v accM = accC;
    if (wave < 4)   // (waves 4..15 only served the per-gene phase)
      for (int t = 64 * wave; t < 64 * wave + 64; t += 2) {
        const float wv = (r < K) ? s_w[(t + h) * FOLD_KMAX + r] : 0.f;
        accC = mfma_f32(s_dv[t + h] * wv, wv, accC);
        accM = mfma_f32((r == 0) ? s_dm[t + h] : 0.f, wv, accM);
      }
    if (wave > 0 && wave < 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) s_acc[wave - 1][i][lane] = accC[i];
      s_acc[wave - 1][16][lane] = accM[0];
    }
    __syncthreads();
    if (wave > 0) return;
    float* out = q.red_part + (long)blockIdx.x * nred;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = crow(i, h);
      const float v = ((accC[i] + s_acc[0][i][lane]) + s_acc[1][i][lane]) + s_acc[2][i][lane];
      if (row < K && r < K) out[K + row * K + r] = v;
    }
    if (h == 0 && r < K) out[r] = ((accM[0] + s_acc[0][16][lane]) + s_acc[1][16][lane]) + s_acc[2][16][lane];
  }
}

// red = sum over gene blocks of red_part, in block order within four interleaved groups; stored in row nblk of red_part
__global__ __launch_bounds__(256) void fold_red_finalize_kernel(FoldBatch a) {
  const FoldProb& q = a.p[blockIdx.y];
  __shared__ float s_p[4][64];
  const int K = q.K, nred = K + K * K;
  const int nblk = (q.G + 255) / 256;
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const int i = blockIdx.x * 64 + tx;
  float s = 0.f;
  if (i < nred)
#pragma unroll 8
    for (int k = ty; k < nblk; k += 4) s += q.red_part[(long)k * nred + i];
  s_p[ty][tx] = s;
  __syncthreads();
  if (ty == 0 && i < nred) q.red_part[(long)nblk * nred + i] = ((s_p[0][tx] + s_p[1][tx]) + s_p[2][tx]) + s_p[3][tx];
}

// d z[b][k] (+)= d zbar[k] / B + (1/B) sum_l (dC + dC^T)[k][l] (z[b][l] - zbar[l]), with (d zbar | dC) = the sum over the gene blocks of
// bn_fold_bwd_kernel's partials, taken HERE by every workgroup (block order within four interleaved groups, groups added in order:
// fold_red_finalize_kernel's association -- that launch is no longer needed in front of this one; workgroup 0 still leaves the sum in
// row nblk of red_part).  With q.out_priv set the finished d zcat columns of this problem go straight to d private_log_z / d poe_log_z
// (the latent-slicing backward, zsplit_bwd_kernel) instead of back into q.dz.
__global__ __launch_bounds__(256) void zstats_bwd_kernel(FoldBatch a) {
  const FoldProb& q = a.p[blockIdx.y];
  __shared__ float s_zbar[FOLD_KMAX], s_dz[FOLD_KMAX], s_S[FOLD_KMAX * FOLD_KMAX], s_red[FOLD_KMAX + FOLD_KMAX * FOLD_KMAX], s_z[64 * (FOLD_KMAX + 1)];
  const int K = q.K, nred = K + K * K;
  const int nblk = (q.G + 255) / 256;
  const float invB = 1.0f / (float)a.B;
  for (int i = threadIdx.x; i < nred; i += 256) {
    float g4[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = 0; k0 < nblk; k0 += 4) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { const float v = q.red_part[(long)min(k0 + u, nblk - 1) * nred + i]; g4[u] += (k0 + u < nblk) ? v : 0.f; }
    }
    const float r = ((g4[0] + g4[1]) + g4[2]) + g4[3];
    s_red[i] = r;
    if (blockIdx.x == 0) q.red_part[(long)nblk * nred + i] = r;
  }
  for (int i = threadIdx.x; i < K; i += 256) s_zbar[i] = q.zsum[i] * invB;
  __syncthreads();
  for (int i = threadIdx.x; i < K; i += 256) s_dz[i] = s_red[i] * invB;
  for (int i = threadIdx.x; i < K * K; i += 256) s_S[i] = (s_red[K + i] + s_red[K + (i % K) * K + i / K]) * invB;
  // Workgroup = 64 cells: their z rows are staged (centred) in LDS; thread (kg = tid / 64, cell = tid % 64) forms the outputs k = kg, kg + 4, ...
  // -- the 64 lanes of a wave share their k set, so every S[k][l] read is a broadcast and the eight sums are independent chains -- and the
  // finished values go back through LDS for coalesced accesses to d z and the outputs.  (Round 4; before: four threads per cell, each a
  // K x K chain of LDS reads against L1 reads of z: 19 us at C2 on the critical chain of the backward pass.  Same summation order per output.)
  const int b0 = blockIdx.x * 64;
  for (int i = threadIdx.x; i < 64 * K; i += 256) {
    const int c = i / K, l = i - c * K, b = min(b0 + c, a.B - 1);
    s_z[c * (FOLD_KMAX + 1) + l] = q.z[(long)b * q.ldz + l] - s_zbar[l];
  }
  __syncthreads();
  const int c = threadIdx.x & 63, kg = threadIdx.x >> 6;
  constexpr int KK = FOLD_KMAX / 4;
  float acc[KK];
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) acc[kk] = s_dz[min(kg + 4 * kk, K - 1)];
  for (int l = 0; l < K; ++l) {
    const float zl = s_z[c * (FOLD_KMAX + 1) + l];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) acc[kk] += s_S[min(kg + 4 * kk, K - 1) * K + l] * zl;
  }
  __syncthreads();   // every thread has read its z row: the tile is reused for the results
#pragma unroll
  for (int kk = 0; kk < KK; ++kk) {
    const int k = kg + 4 * kk;
    if (k < K) s_z[c * (FOLD_KMAX + 1) + k] = acc[kk];
  }
  __syncthreads();
  for (int i = threadIdx.x; i < 64 * K; i += 256) {
    const int cc = i / K, k = i - cc * K, b = b0 + cc;
    if (b >= a.B) continue;
    const float out = q.dz[(long)b * q.lddz + k] + s_z[cc * (FOLD_KMAX + 1) + k];
    if (q.out_priv != nullptr) {
      const int col = q.zcol + k;                                     // column of zcat = [z_private | z_shared]
      const int zc = (col >= q.n_p) ? col - q.n_p : col + q.n_s;      // the column of Z = cat(private_log_z, poe_log_z) it came from
      if (zc < q.n_p) q.out_priv[(long)b * q.n_p + zc] = out;
      else q.out_poe[(long)b * q.n_s + zc - q.n_p] = out;
    } else {
      q.dz[(long)b * q.lddz + k] = out;
    }
  }
}


// ---------------------------------------------------------------------------------------------
// Decoder mixing trunk m = relu(BN(cat(z) W_a^T + b_a)) with the training-mode BatchNorm folded (spv_trunk_prob in the header;
// nn/networks.py:322-323).  mean0_j = w_j . zbar (the Linear's bias cancels in the batch mean), var_j = w_j^T C w_j,
// inv_j = gamma_j / sqrt(var_j + eps):  W'_j = inv_j w_j,  c'_j = beta_j - inv_j mean0_j.   stat = (mean0, var).
// Eval mode: mean0_j = running_mean_j - b_j, var_j = running_var_j.
// ---------------------------------------------------------------------------------------------
typedef spv_trunk_prob TrunkProb;
typedef spv_trunk_batch TrunkBatch;
constexpr int TRUNK_KMAX = SPV_TRUNK_KMAX;

__device__ __forceinline__ void trunk_stats_to_lds(const TrunkProb& q, int B, float* s_zbar, float* s_C, int nthreads) {
  const int K = q.K;
  const float invB = 1.0f / (float)B;
  for (int i = threadIdx.x; i < K; i += nthreads) s_zbar[i] = q.zsum[i] * invB;
  for (int i = threadIdx.x; i < K * K; i += nthreads) s_C[i] = q.zz[i] * invB - (q.zsum[i / K] * invB) * (q.zsum[i % K] * invB);
}

// One WAVE per row of the layer (lane l = column l of the row, K <= 48 < 64), 16 rows per workgroup: (C w)_l = sum_k w_k C[k][l] is a K-step
// loop with w_k broadcast from lane k and C[k][.] read coalesced from LDS -- the first version gave a row to 4 lanes, each walking its share
// of the K x K products as one dependent chain (14 us for 2 x 256 rows on 8 CUs).
constexpr int TRUNK_FWD_ROWS = 16;
// (C w)_l for the row whose weights sit in LDS at wrow[0..K): w_k is a broadcast read, C[k][.] a coalesced one; independent of lane l's own
// chain only through the accumulate (a __shfl of w_k from lane k costs a ds_bpermute per step: 35 of them in a row made this loop the
// kernels' longest stretch)
__device__ __forceinline__ float trunk_Cw(const float* s_C, const float* wrow, int K, int l) {
  float t = 0.f;
  const int lc = min(l, K - 1);
#pragma unroll 4
  for (int k = 0; k < K; ++k) t += wrow[k] * s_C[k * K + lc];
  return (l < K) ? t : 0.f;
}
__global__ __launch_bounds__(1024) void trunk_fold_fwd_kernel(TrunkBatch a) {
  const TrunkProb& q = a.p[blockIdx.y];
  __shared__ float s_zbar[TRUNK_KMAX], s_C[TRUNK_KMAX * TRUNK_KMAX], s_wr[TRUNK_FWD_ROWS][64];
  const int K = q.K;
  if (blockIdx.x * TRUNK_FWD_ROWS >= q.N) return;   // (the grid is sized for the largest problem)
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6, j = blockIdx.x * TRUNK_FWD_ROWS + wv;
  const int ji = min(j, q.N - 1);
  const float wl = (l < K) ? q.W[(long)ji * K + l] : 0.f;   // (requested before the statistics: one round trip for both)
  const float bj = q.bias ? q.bias[ji] : 0.f;
  s_wr[wv][l] = wl;
  if (a.training) trunk_stats_to_lds(q, a.B, s_zbar, s_C, 1024);
  __syncthreads();
  if (j >= q.N) return;   // (wave-uniform)
  float mean0, var;
  if (a.training) {
    const float t = trunk_Cw(s_C, s_wr[wv], K, l);
    mean0 = wave_sum_all(wl * ((l < K) ? s_zbar[l] : 0.f));
    var = fmaxf(wave_sum_all(wl * t), 0.f);
    if (l == 0) {   // torch: running statistics take the batch mean (bias included) and the UNBIASED variance
      q.running_mean[j] = (1.f - a.momentum) * q.running_mean[j] + a.momentum * (mean0 + bj);
      q.running_var[j] = (1.f - a.momentum) * q.running_var[j] + a.momentum * var * ((float)a.B / fmaxf((float)a.B - 1.f, 1.f));
    }
  } else {
    mean0 = q.running_mean[j] - bj;
    var = q.running_var[j];
  }
  const float inv = q.gamma[j] * rsqrtf(var + a.eps);
  if (l < K) q.Wf[(long)j * K + l] = wl * inv;
  if (l == 0) { q.cf[j] = q.beta[j] - inv * mean0; q.stat[2 * j] = mean0; q.stat[2 * j + 1] = var; }
}

// backward of the fold: 64 rows per workgroup (one wave per row, four passes of 16), then the block's partial of
// (d zbar | d C) = (sum_j dmean0_j w_j | sum_j dvar_j w_j w_j^T) on the exact-fp32 matrix pipe: wave t < 4 owns the 32 x 32 tile (t >> 1, t & 1)
// of d C (K <= 48: 2 x 2 tiles) and, in the first tile row, d zbar as row 0 of a second accumulator.  One partial [K + K*K] per workgroup in
// q.dred (summed in block order by trunk_zstats_kernel): a fixed summation order, no atomics.
constexpr int TRUNK_BWD_ROWS = 64;
__global__ __launch_bounds__(1024) void trunk_fold_bwd_kernel(TrunkBatch a) {
  const TrunkProb& q = a.p[blockIdx.y];
  __shared__ float s_zbar[TRUNK_KMAX], s_C[TRUNK_KMAX * TRUNK_KMAX], s_w[TRUNK_BWD_ROWS * TRUNK_KMAX], s_dm[TRUNK_BWD_ROWS], s_dv[TRUNK_BWD_ROWS];
  const int K = q.K;
  const int j0 = blockIdx.x * TRUNK_BWD_ROWS;
  if (j0 >= q.N) return;
  if (a.training) trunk_stats_to_lds(q, a.B, s_zbar, s_C, 1024);
  __syncthreads();
  const int l = threadIdx.x & 63, wv = threadIdx.x >> 6;
  for (int pass = 0; pass < TRUNK_BWD_ROWS / 16; ++pass) {
    const int jl = 16 * pass + wv, j = j0 + jl;
    const bool real = j < q.N;   // (wave-uniform)
    const int ji = min(j, q.N - 1);
    const float wl = (real && l < K) ? q.W[(long)ji * K + l] : 0.f;
    const float gWl = (real && l < K) ? q.dWf[(long)ji * K + l] : 0.f;
    const float mean0 = q.stat[2 * ji], var = q.stat[2 * ji + 1];
    const float rs = rsqrtf(var + a.eps), gam = q.gamma[ji], inv = gam * rs;
    const float gc = real ? q.dcf[ji] : 0.f;
    const float dinv = wave_sum_all(gWl * wl) - gc * mean0;
    float dmean = 0.f, dvar = 0.f;
    if (real && l == 0) {
      q.dbeta[j] = gc;
      q.dgamma[j] = dinv * rs;
      if (q.dbias) q.dbias[j] = a.training ? 0.f : gc * inv;   // training: the bias cancels in the batch mean
    }
    if (a.training) {
      if (l < TRUNK_KMAX) s_w[jl * TRUNK_KMAX + l] = wl;   // (zero beyond K and for rows past N; row jl belongs to this wave alone)
      const float v = trunk_Cw(s_C, s_w + jl * TRUNK_KMAX, K, l);
      if (real) { dmean = -gc * inv; dvar = dinv * gam * (-0.5f) * rs * rs * rs; }
      if (real && l < K) q.dW[(long)j * K + l] = gWl * inv + dmean * s_zbar[l] + 2.0f * dvar * v;
      if (l == 0) { s_dm[jl] = dmean; s_dv[jl] = dvar; }
    } else if (real && l < K) {
      q.dW[(long)j * K + l] = gWl * inv;
    }
  }
  if (!a.training) return;
  __syncthreads();
  const int lane = l, h = lane >> 5, r = lane & 31;
  if (wv >= 4) return;
  const int tr = wv >> 1, tc = wv & 1;
  if (32 * tr >= K || 32 * tc >= K) return;   // (wave-uniform: a tile outside the K x K block)
  f16v accC, accM;
#pragma unroll
  for (int i = 0; i < 16; ++i) { accC[i] = 0.f; accM[i] = 0.f; }
  const int ca = 32 * tr + r, cb = 32 * tc + r;
  for (int t = 0; t < TRUNK_BWD_ROWS; t += 2) {
    const float wa = (ca < TRUNK_KMAX) ? s_w[(t + h) * TRUNK_KMAX + min(ca, TRUNK_KMAX - 1)] : 0.f;
    const float wb = (cb < TRUNK_KMAX) ? s_w[(t + h) * TRUNK_KMAX + min(cb, TRUNK_KMAX - 1)] : 0.f;
    accC = mfma_f32(s_dv[t + h] * wa, wb, accC);
    if (tr == 0) accM = mfma_f32((r == 0) ? s_dm[t + h] : 0.f, wb, accM);
  }
  float* out = q.dred + (long)blockIdx.x * (K + K * K);
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = 32 * tr + crow(i, h), col = cb;
    if (row < K && col < K) out[K + row * K + col] = accC[i];
  }
  if (tr == 0 && h == 0 && cb < K) out[cb] = accM[0];   // register 0 of lane half 0 = row 0 of the tile
}

// d z[b][k] += d zbar[k] / B + (1 / B) sum_l (d C + d C^T)[k][l] (z[b][l] - zbar[l]),  (d zbar | d C) = the sum of trunk_fold_bwd_kernel's block
// partials in block order.  Workgroup = 64 cells: their z rows are staged (centred) in LDS, thread (kg = tid / 64, cell = tid % 64) forms the
// outputs k = kg, kg + 4, ... -- the 64 lanes of a wave share their k set, so every S[k][l] read is a broadcast and the twelve sums are
// independent chains; the results go back through LDS for coalesced read-modify-writes of d z.
constexpr int TRUNK_ZS_KK = (TRUNK_KMAX + 3) / 4;
__global__ __launch_bounds__(256) void trunk_zstats_kernel(TrunkBatch a, int nblk) {
  const TrunkProb& q = a.p[blockIdx.y];
  __shared__ float s_zbar[TRUNK_KMAX], s_dz[TRUNK_KMAX], s_S[TRUNK_KMAX * TRUNK_KMAX], s_z[64 * (TRUNK_KMAX + 1)];
  const int K = q.K, nred = K + K * K;
  const float invB = 1.0f / (float)a.B;
  const int nb = (q.N + TRUNK_BWD_ROWS - 1) / TRUNK_BWD_ROWS;   // (<= 4: N <= 256)
  // the block partials are requested all at once (clamped index + select: a loop over a run-time count is one dependent round trip per partial)
  auto part4 = [&](int i) {
    float v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = q.dred[(long)min(u, nb - 1) * nred + i];
    float d = v[0];
#pragma unroll
    for (int u = 1; u < 4; ++u) d += (u < nb) ? v[u] : 0.f;
    return d;
  };
  for (int i = threadIdx.x; i < K; i += 256) { s_zbar[i] = q.zsum[i] * invB; s_dz[i] = part4(i) * invB; }
  for (int i = threadIdx.x; i < K * K; i += 256) {
    const int k = i / K, l = i - k * K;
    s_S[i] = (part4(K + i) + part4(K + l * K + k)) * invB;
  }
  const int b0 = blockIdx.x * 64;
  __syncthreads();   // (s_zbar is read by the staging below)
  for (int i = threadIdx.x; i < 64 * K; i += 256) {
    const int c = i / K, l = i % K, b = min(b0 + c, a.B - 1);
    s_z[c * (TRUNK_KMAX + 1) + l] = q.z[(long)b * q.ldz + l] - s_zbar[l];
  }
  __syncthreads();
  const int c = threadIdx.x & 63, kg = threadIdx.x >> 6;
  float acc[TRUNK_ZS_KK];
#pragma unroll
  for (int kk = 0; kk < TRUNK_ZS_KK; ++kk) acc[kk] = s_dz[min(kg + 4 * kk, K - 1)];
  for (int l = 0; l < K; ++l) {
    const float zl = s_z[c * (TRUNK_KMAX + 1) + l];
#pragma unroll
    for (int kk = 0; kk < TRUNK_ZS_KK; ++kk) acc[kk] += s_S[min(kg + 4 * kk, K - 1) * K + l] * zl;
  }
  __syncthreads();   // every thread has read its z row: the tile is reused for the outputs
#pragma unroll
  for (int kk = 0; kk < TRUNK_ZS_KK; ++kk) {
    const int k = kg + 4 * kk;
    if (k < K) s_z[c * (TRUNK_KMAX + 1) + k] = acc[kk];
  }
  __syncthreads();
  (void)nblk;
  for (int i = threadIdx.x; i < 64 * K; i += 256) {
    const int cc = i / K, k = i % K, b = b0 + cc;
    if (b < a.B) q.dz[(long)b * q.lddz + k] += s_z[cc * (TRUNK_KMAX + 1) + k];
  }
}


// ---- deterministic slab reduction ------------------------------------------------------------------
// few slabs (<= 8): one thread per element adds them in order.  Many slabs: block = 64 elements x 4 slab groups;
// group y adds slabs y, y+4, ... in order and the four group sums are then added in order.
__device__ __forceinline__ void reduce_finish(const spv_reduce_prob& q, int r, int c, float v, float alpha) {
  v *= alpha;
  if (q.exp_scale) v *= expf(q.exp_scale[c]);
  float* d = q.dst + (long)r * q.ld_dst + c;
  *d = q.accumulate ? *d + v : v;
}
// NS slabs, NS known at compile time: NS independent loads per element, then an ordered sum (with a run-time count every load sits
// behind its own `k < nslabs` branch and is waited for there: NS dependent round trips per element instead of one).  V = 4 elements
// of a row per thread (16-byte loads and stores) where the problem's columns, offsets, pitches and pointers allow (reduce_vec4).
constexpr int RED_FEW_MAX = 8;   // (more slabs in registers cost the kernel its occupancy: 16 x 4 elements = 88 VGPRs, and the 3-slab d W_m sum went 22.8 -> 28.9 us)
template <int V> struct RedVec { float x[V]; };
template <int V> __device__ __forceinline__ RedVec<V> red_ld(const float* p);
template <> __device__ __forceinline__ RedVec<1> red_ld<1>(const float* p) { RedVec<1> r; r.x[0] = *p; return r; }
template <> __device__ __forceinline__ RedVec<4> red_ld<4>(const float* p) {
  const float4 t = *reinterpret_cast<const float4*>(p);
  RedVec<4> r; r.x[0] = t.x; r.x[1] = t.y; r.x[2] = t.z; r.x[3] = t.w; return r;
}
__device__ __forceinline__ void red_st(float* p, const RedVec<1>& v) { *p = v.x[0]; }
__device__ __forceinline__ void red_st(float* p, const RedVec<4>& v) { *reinterpret_cast<float4*>(p) = make_float4(v.x[0], v.x[1], v.x[2], v.x[3]); }
__host__ __device__ inline bool reduce_vec4(const spv_reduce_prob& q) {
  return !((q.cols | q.col_off) & 3) && !((q.ld_src | q.slab_stride | q.ld_dst) & 3) &&
         !(((unsigned long long)q.src | (unsigned long long)q.dst | (unsigned long long)q.exp_scale) & 15ull);
}
// the sum in slab order
template <int NS>
__device__ __forceinline__ float reduce_ordered_sum(const float (&v)[NS]) {
  float acc = v[0];
#pragma unroll
  for (int k = 1; k < NS; ++k) acc += v[k];
  return acc;
}
// EXTRA: the problem has a column scale and / or accumulates
template <int NS, int V, bool EXTRA>
__device__ __forceinline__ void reduce_few_slabs(const spv_reduce_prob& q, float alpha) {
  const unsigned cols_v = (unsigned)q.cols / V, total = (unsigned)q.rows * cols_v;   // (rows * cols < 2^31: spv_reduce_slabs)
  for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
    const unsigned r = i / cols_v, c = (i - r * cols_v) * V;
    const float* s = q.src + (long)r * q.ld_src + q.col_off + c;
    float* d = q.dst + (long)r * q.ld_dst + c;
    RedVec<V> v[NS];
#pragma unroll
    for (int k = 0; k < NS; ++k) v[k] = red_ld<V>(s + (long)k * q.slab_stride);
    // the optional scale and the accumulation target through selected addresses: in flight together with the slabs
    RedVec<V> e, prev, out;
    if constexpr (EXTRA) {
      e = red_ld<V>(q.exp_scale ? q.exp_scale + c : g_spv_zero4);
      prev = red_ld<V>(q.accumulate ? d : g_spv_zero4);
    }
#pragma unroll
    for (int j = 0; j < V; ++j) {
      float t[NS];
#pragma unroll
      for (int k = 0; k < NS; ++k) t[k] = v[k].x[j];
      float a = reduce_ordered_sum<NS>(t) * alpha;
      if constexpr (EXTRA) {
        if (q.exp_scale) a *= expf(e.x[j]);
        a = q.accumulate ? prev.x[j] + a : a;
      }
      out.x[j] = a;
    }
    red_st(d, out);
  }
}
template <int V, bool EXTRA>
__device__ __forceinline__ void reduce_few_dispatch(const spv_reduce_prob& q, float alpha) {
  switch (q.nslabs) {   // (block-uniform)
    case 1: reduce_few_slabs<1, V, EXTRA>(q, alpha); break;
    case 2: reduce_few_slabs<2, V, EXTRA>(q, alpha); break;
    case 3: reduce_few_slabs<3, V, EXTRA>(q, alpha); break;
    case 4: reduce_few_slabs<4, V, EXTRA>(q, alpha); break;
    case 5: reduce_few_slabs<5, V, EXTRA>(q, alpha); break;
    case 6: reduce_few_slabs<6, V, EXTRA>(q, alpha); break;
    case 7: reduce_few_slabs<7, V, EXTRA>(q, alpha); break;
    default: reduce_few_slabs<8, V, EXTRA>(q, alpha); break;
  }
}
// workgroups problem q needs (one pass): 256 threads x 1 or 4 elements on the few-slab path, 64 elements on the many-slab path
__host__ __device__ inline long reduce_blocks_needed(const spv_reduce_prob& q) {
  const long total = (long)q.rows * q.cols;
  if (q.nslabs > RED_FEW_MAX) return (total + 63) / 64;
  return reduce_vec4(q) ? (total / 4 + 255) / 256 : (total + 255) / 256;
}
__global__ __launch_bounds__(256) void reduce_slabs_kernel(spv_reduce_batch b) {
  __shared__ float s_part[4][64];
  const spv_reduce_prob& q = b.p[blockIdx.y];
  const long total = (long)q.rows * q.cols;
  const float alpha = *(q.alpha ? q.alpha : &g_spv_one);
  if (q.nslabs <= RED_FEW_MAX) {
    const bool extra = q.exp_scale || q.accumulate;
    if (reduce_vec4(q)) { if (extra) reduce_few_dispatch<4, true>(q, alpha); else reduce_few_dispatch<4, false>(q, alpha); }
    else { if (extra) reduce_few_dispatch<1, true>(q, alpha); else reduce_few_dispatch<1, false>(q, alpha); }
    return;
  }
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  for (long base = (long)blockIdx.x * 64; base < total; base += (long)gridDim.x * 64) {  // block-uniform trip count
    const long i = base + tx;
    const bool ok = i < total;
    int r = 0, c = 0;
    float acc = 0.f;
    if (ok) {
      r = (int)(i / q.cols); c = (int)(i - (long)r * q.cols);
      const float* s = q.src + (long)r * q.ld_src + q.col_off + c;
#pragma unroll 8
      for (int k = ty; k < q.nslabs; k += 4) acc += s[(long)k * q.slab_stride];
    }
    s_part[ty][tx] = acc;
    __syncthreads();
    if (ty == 0 && ok) reduce_finish(q, r, c, ((s_part[0][tx] + s_part[1][tx]) + s_part[2][tx]) + s_part[3][tx], alpha);
    __syncthreads();
  }
}

// ---- loss assembly: one workgroup, fixed-order tree --------------------------------------------------------
__global__ __launch_bounds__(1024) void loss_assemble_kernel(const float* rec0, const float* rec1, const float* w, const float* kl0,
                                                             const float* kl1, const float* kl2, const float* kl3, int B,
                                                             const float* kl_weight, float* loss, float* rec_sum, float* gkl) {
  __shared__ float s_r[1024], s_k[1024];
  const int t = threadIdx.x;
  float ar = 0.f, ak = 0.f;
  const float klw = kl_weight ? *kl_weight : 1.f;
  const float gk = klw / (float)B;
  for (int i = t; i < B; i += 1024) {
    const float wi = w[i];   // (optional inputs through ld_or_zero: seven loads in flight at once instead of seven dependent round trips)
    ar += wi * rec0[i];
    ar += wi * ld_or_zero(rec1, i);
    float k = 0.f;
    k += ld_or_zero(kl0, i);
    k += ld_or_zero(kl1, i);
    k += ld_or_zero(kl2, i);
    k += ld_or_zero(kl3, i);
    ak += k;
    if (gkl) gkl[i] = gk;
  }
  s_r[t] = ar; s_k[t] = ak;
  __syncthreads();
  for (int o = 512; o > 0; o >>= 1) {
    if (t < o) { s_r[t] += s_r[t + o]; s_k[t] += s_k[t + o]; }
    __syncthreads();
  }
  if (t == 0) {
    if (rec_sum) *rec_sum = s_r[0];
    *loss = s_r[0] + gk * s_k[0];
  }
}

}  // namespace spv
