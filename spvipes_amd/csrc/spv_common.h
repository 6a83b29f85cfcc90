// Common device helpers for the spVIPES hot-path kernels (gfx950 / CDNA4 only).
//
// Conventions
//   * bf16 values travel as raw 16-bit words (bf16_t = unsigned short).
//   * "split" operands: a float is carried as hi = bf16(x), lo = bf16(x - hi); the product
//     a*b is then accumulated as a_hi*b_lo + a_lo*b_hi + a_hi*b_hi (three MFMAs, ~2^-17
//     relative error per product) -- the "fp32" precision mode.  NSPLIT == 1 is plain bf16.
//   * MFMA shape: v_mfma_f32_32x32x16_bf16.  Lane l = (r = l & 31, h = l >> 5):
//       A fragment: A[row r][k = 8h + j], j = 0..7        (8 bf16 = 16 bytes)
//       B fragment: B[k = 8h + j][col r]
//       C/D       : reg q -> row (q & 3) + 8 (q >> 2) + 4 h, col r
//     (verified with exact integer data by tools/probes/mfma_tr_probe.hip)
//   * ds_read_b64_tr_b16 ("tr read"): per 16-lane group a 4(k) x 16(col) block of a k-major
//     LDS image is delivered column-major; lane 4q+p supplies the address of row q, cols 4p..4p+3.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef unsigned short bf16_t;
typedef __attribute__((ext_vector_type(4))) short s4v;
typedef __attribute__((ext_vector_type(8))) short s8v;
typedef __attribute__((ext_vector_type(4))) float f4v;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) unsigned int u4v;
typedef __attribute__((ext_vector_type(2))) unsigned int u2v;
typedef __attribute__((address_space(3))) s4v lds_s4v;

#define SPV_EPS_NB 1e-8f

// status codes returned through the C ABI
#define SPV_OK 0
#define SPV_ERR_ARG (-1)
#define SPV_ERR_LAUNCH (-2)
#define SPV_ERR_UNSUPPORTED (-3)

namespace spv {

__device__ __forceinline__ bf16_t f2bf(float f) {
  // plain cast -> v_cvt_pk_bf16_f32 (RNE, NaN stays NaN)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
__device__ __forceinline__ float bf2f(bf16_t b) { return __uint_as_float(((unsigned)b) << 16); }
// two values -> one word (a in the low half): ONE v_cvt_pk_bf16_f32.  Written as a vector conversion because the library
// is built without SLP vectorisation (build.py), which would otherwise be what pairs two scalar casts.
__device__ __forceinline__ unsigned pack2bf(float a, float b) {
  typedef float f2cv __attribute__((ext_vector_type(2)));
  typedef __bf16 b2cv __attribute__((ext_vector_type(2)));
  const b2cv r = __builtin_convertvector(f2cv{a, b}, b2cv);
  return __builtin_bit_cast(unsigned, r);
}
// hi / lo words of two values (x = hi + lo to ~2^-17 relative)
__device__ __forceinline__ void split2bf(float a, float b, unsigned& hi, unsigned& lo) {
  hi = pack2bf(a, b);
  lo = pack2bf(a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u));
}

__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
  hi = f2bf(x);
  lo = f2bf(x - bf2f(hi));
}

// raw v_log_f32 / v_exp_f32 (base 2, ~1 ulp, no denormal rescue code around them: every operand on
// the hot path is a normal number or may flush to zero)
__device__ __forceinline__ float fast_log(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
__device__ __forceinline__ float fast_exp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }

// log1p of a raw count as the reference computes it: torch.log(1 + x) in fp32
// (module/spVIPESmodule.py:433, :821)
__device__ __forceinline__ float log1p_count(float c) { return fast_log(1.0f + c); }

template <typename CT>
__device__ __forceinline__ float count_to_float(CT c);
template <>
__device__ __forceinline__ float count_to_float<float>(float c) { return c; }
template <>
__device__ __forceinline__ float count_to_float<unsigned short>(unsigned short c) { return (float)c; }

// ---- MFMA wrappers ---------------------------------------------------------
__device__ __forceinline__ f16v mfma32(const s8v& a, const s8v& b, const f16v& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
}
template <int NSPLIT>
__device__ __forceinline__ f16v mfma32_split(const s8v& a_hi, const s8v& a_lo, const s8v& b_hi, const s8v& b_lo, f16v c) {
  if constexpr (NSPLIT == 3) {
    c = mfma32(a_hi, b_lo, c);
    c = mfma32(a_lo, b_hi, c);
  }
  return mfma32(a_hi, b_hi, c);
}

// ---- IEEE half operands for the encoder's first layer (NSPLIT == 1) -------------------------------------------------------
// The fc1 contraction runs over 10 000 .. 30 000 genes; with both operands rounded to bf16 (8 significant bits) the latent means
// came out 5-6e-3 off the fp32 reference, outside the 1e-3 the north star asks for.  log1p(count) <= 11.1 and |W| << 1 fit
// f16's range, f16 keeps 11 significant bits and v_mfma_f32_32x32x16_f16 runs at the bf16 rate, so in the "bf16" precision
// mode the fc1 operand images (resident log1p image, weight image, dh image of the weight gradient) are f16 words.  The
// weight image holds W * FC1_W_SCALE (a power of two: exact) so that small weights stay clear of f16's subnormal range; the
// forward epilogue multiplies the accumulated sums by 1 / FC1_W_SCALE.  The dh image is scaled per step by a power of two
// derived from max |dh| (fc1_bwd_* kernels) and the weight-gradient kernels multiply by its inverse on the way out.
#define SPV_FC1_W_SCALE 256.0f
typedef _Float16 h8v __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f16v mfma32h(const s8v& a, const s8v& b, const f16v& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8v, a), __builtin_bit_cast(h8v, b), c, 0, 0, 0);
}
template <bool HALF>
__device__ __forceinline__ f16v mfma32x(const s8v& a, const s8v& b, const f16v& c) {
  if constexpr (HALF) return mfma32h(a, b, c);
  else return mfma32(a, b, c);
}
// fp32 -> f16 word, round to nearest even, saturating (an overflow must not become an infinity inside a GEMM operand)
__device__ __forceinline__ bf16_t f2h(float f) {
  const float c = fminf(fmaxf(f, -65504.f), 65504.f);   // (fminf / fmaxf drop a NaN operand: a NaN must stay a NaN)
  const _Float16 h = (_Float16)(f == f ? c : f);
  return __builtin_bit_cast(bf16_t, h);
}
__device__ __forceinline__ float h2f(bf16_t w) { return (float)__builtin_bit_cast(_Float16, w); }
__device__ __forceinline__ unsigned pack2h(float a, float b) { return (unsigned)f2h(a) | ((unsigned)f2h(b) << 16); }
// power-of-two scale that brings a tensor whose largest magnitude is `amax` to [4096, 8192): (scale, 1 / scale), both exact
__device__ __forceinline__ void pow2_scale_for(float amax, float& scale, float& inv) {
  scale = 1.f; inv = 1.f;
  if (amax > 0.f && amax < 3.0e38f) {   // (zero, infinity, NaN: leave the values alone)
    int e = (int)((__float_as_uint(amax) >> 23) & 0xFF) - 126;   // amax = m * 2^e, m in [0.5, 1) (subnormal amax: e = -126, close enough)
    int k = 13 - e;
    k = k > 100 ? 100 : (k < -100 ? -100 : k);
    scale = __uint_as_float((unsigned)(127 + k) << 23);
    inv = __uint_as_float((unsigned)(127 - k) << 23);
  }
}

// C/D row owned by accumulator register q of lane-half h
__device__ __forceinline__ int crow(int q, int h) { return (q & 3) + 8 * (q >> 2) + 4 * h; }

// p ? p[i] : 0 as an UNCONDITIONAL load (the address is selected, not the load skipped): a load behind a null test is waited for inside
// that branch, and a kernel that reads five optional gradients pays five dependent round trips instead of one
__device__ const float g_spv_zero = 0.f;
__device__ __forceinline__ float ld_or_zero(const float* p, long i) { return *(p ? p + i : &g_spv_zero); }
__device__ const float g_spv_one = 1.f;
__device__ const float g_spv_zero4[4] __attribute__((aligned(16))) = {0.f, 0.f, 0.f, 0.f};

// ---- fragment reads from LDS -----------------------------------------------
// natural image: [rows][pitch] bf16, k contiguous.  Fragment of the 32-row tile starting at
// row0, k-step kbase (multiple of 16): lane reads 16 B at (row0 + r, kbase + 8h).
__device__ __forceinline__ s8v frag_natural(const bf16_t* img, int pitch, int row0, int kbase, int lane) {
  return *reinterpret_cast<const s8v*>(img + (row0 + (lane & 31)) * pitch + kbase + 8 * (lane >> 5));
}
// k-major image: [k][pitch] bf16, tile columns contiguous.  Fragment of the 32-column tile
// starting at col0 for k-step kbase: two transposed reads (k = 8h + 0..3 and 8h + 4..7).
__device__ __forceinline__ s8v frag_kmajor(const bf16_t* img, int pitch, int col0, int kbase, int lane) {
  const int gi = lane >> 4, q = (lane & 15) >> 2, p = lane & 3, h = lane >> 5;
  const bf16_t* base = img + (kbase + 8 * h + q) * pitch + col0 + 16 * (gi & 1) + 4 * p;
  s4v v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(base));
  s4v v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4v*)(base + 4 * pitch));
  s8v out;
  out[0] = v0[0]; out[1] = v0[1]; out[2] = v0[2]; out[3] = v0[3];
  out[4] = v1[0]; out[5] = v1[1]; out[6] = v1[2]; out[7] = v1[3];
  return out;
}

// Workgroup barrier that orders LDS traffic only: wait for this wave's LDS operations, then s_barrier.  __syncthreads() is a workgroup
// FENCE plus the barrier, and the fence also drains every global load / store the wave has in flight (s_waitcnt vmcnt(0)): a kernel
// that keeps tiles of prefetch in flight across the barrier loses them at every barrier and runs at memory LATENCY.  Use this one where
// the only data handed between the waves goes through LDS.  (The "memory" clobber keeps the compiler from moving memory operations
// across it.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// ---- cross-lane helpers ------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}
// sum over all 64 lanes, result in every lane
__device__ __forceinline__ float wave_sum_all(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
// sum over the 32 lanes of each wave half (result valid in every lane of the half)
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
  return v;
}
__device__ __forceinline__ float other_half(float v) { return __shfl_xor(v, 32, 64); }

// ---- special functions (fp32, a > 0) -----------------------------------------
// lgamma(a) and digamma(a) by an 8-step upward shift (while a < 8) + Stirling / asymptotic
// series at z >= 8.  Absolute error ~1e-6 for 0.01 < a < 1e4.  Both share log z, 1/z, 1/P.
struct LgammaDigamma { float lg, dg; };
__device__ __forceinline__ LgammaDigamma lgamma_digamma(float a) {
  float z = a, P = 1.0f, dP = 0.0f;  // P = prod (a+i), dP = dP/da
  if (a < 8.0f) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      float t = a + (float)i;
      dP = dP * t + P;
      P = P * t;
    }
    z = a + 8.0f;
  }
  const float lz = fast_log(z), iz = fast_rcp(z), iz2 = iz * iz;
  LgammaDigamma o;
  // lgamma(z) = (z-1/2) ln z - z + ln(2pi)/2 + 1/(12z) - 1/(360z^3) + 1/(1260z^5)
  o.lg = (z - 0.5f) * lz - z + 0.9189385332046727f +
         iz * (0.08333333333333333f + iz2 * (-0.002777777777777778f + iz2 * 0.0007936507936507937f)) - fast_log(P);
  // digamma(z) = ln z - 1/(2z) - 1/(12z^2) + 1/(120z^4) - 1/(252z^6)
  o.dg = lz - 0.5f * iz - iz2 * (0.08333333333333333f - iz2 * (0.008333333333333333f - iz2 * 0.003968253968253968f)) -
         dP * fast_rcp(P);
  return o;
}

}  // namespace spv
