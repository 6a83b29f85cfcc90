// Probe (dev tool): how many VALU wave-instructions per cycle does ONE SIMD of gfx950 sustain as a function of the
// number of waves resident on it (1..8), for independent v_fma_f32, for v_exp_f32, and for mixes of the two in the
// ratio the NB-mixture likelihood kernel has (16 transcendental : 70 plain per element)?  VERDICT r01 item 4.
//
//   hipcc --offload-arch=gfx950 -O3 -o valu_issue valu_issue.hip && ./valu_issue
//
// Every wave runs ITER iterations of a fully unrolled body of independent instructions (8 accumulators per kind, so
// that no instruction waits for its predecessor's result).  Waves per SIMD are set by the workgroup size (256 * w
// threads = w waves on each of the CU's 4 SIMDs) with one workgroup per CU (LDS request 100 KiB).  Cycles come from
// s_memtime around the loop of wave 0 of every workgroup (median over workgroups); the clock from s_memrealtime.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum { K_FMA = 0, K_EXP = 1, K_MIX = 2, K_LOG = 3, K_RCP = 4, K_MIX_MFMA = 5, K_PKFMA = 6, K_MIX_DEP = 7, K_FMA_SERIAL = 8, K_TRANS_SERIAL = 9, K_FMA_2CH = 10 };

typedef __attribute__((ext_vector_type(8))) short s8v;
typedef __attribute__((ext_vector_type(4))) float f4v;

#define FMA(a) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a) : "v"(m), "v"(c))
#define EXP(a) asm volatile("v_exp_f32 %0, %0" : "+v"(a))
#define LOG(a) asm volatile("v_log_f32 %0, %0" : "+v"(a))
#define RCP(a) asm volatile("v_rcp_f32 %0, %0" : "+v"(a))

template <int KIND>
__global__ __launch_bounds__(1024) void issue_kernel(float* out, unsigned long long* cyc, unsigned long long* rt, int iters) {
  extern __shared__ char smem[];
  float m = 0.999f + 1e-6f * threadIdx.x, c = 1e-3f;
  float a0 = 1.f, a1 = 1.1f, a2 = 1.2f, a3 = 1.3f, a4 = 1.4f, a5 = 1.5f, a6 = 1.6f, a7 = 1.7f;
  float t0 = 0.5f, t1 = 0.6f, t2 = 0.7f, t3 = 0.8f;
  s8v fa, fb;
  for (int i = 0; i < 8; ++i) { fa[i] = (short)(0x3f80 + threadIdx.x % 7); fb[i] = (short)(0x3f00 + i); }
  f4v acc = {0.f, 0.f, 0.f, 0.f};
  typedef float f2v __attribute__((ext_vector_type(2)));
  f2v p0 = {1.f, 1.1f}, p1 = {1.2f, 1.3f}, p2 = {1.4f, 1.5f}, p3 = {1.6f, 1.7f}, pm = {m, m}, pc = {c, c};
  __syncthreads();
  unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    if constexpr (KIND == K_FMA) {   // 32 plain
#pragma unroll
      for (int r = 0; r < 4; ++r) { FMA(a0); FMA(a1); FMA(a2); FMA(a3); FMA(a4); FMA(a5); FMA(a6); FMA(a7); }
    } else if constexpr (KIND == K_EXP) {   // 32 transcendental (x -> 2^x stays bounded with the fma-free chain: exp2 of a value in (0,2) -> (1,4) -> ...; use log/exp pairs)
#pragma unroll
      for (int r = 0; r < 4; ++r) { EXP(a0); EXP(a1); EXP(a2); EXP(a3); LOG(a0); LOG(a1); LOG(a2); LOG(a3); }
    } else if constexpr (KIND == K_LOG) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { LOG(a0); LOG(a1); LOG(a2); LOG(a3); EXP(a0); EXP(a1); EXP(a2); EXP(a3); }
    } else if constexpr (KIND == K_RCP) {
#pragma unroll
      for (int r = 0; r < 4; ++r) { RCP(a0); RCP(a1); RCP(a2); RCP(a3); RCP(a4); RCP(a5); RCP(a6); RCP(a7); }
    } else if constexpr (KIND == K_MIX || KIND == K_MIX_MFMA) {
      // 8 transcendental + 35 plain = the likelihood kernel's ratio (16 : 70), interleaved
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        FMA(a0); FMA(a1); FMA(a2); FMA(a3); EXP(t0); FMA(a4); FMA(a5); FMA(a6); FMA(a7); LOG(t0);
        FMA(a0); FMA(a1); FMA(a2); FMA(a3); EXP(t1); FMA(a4); FMA(a5); FMA(a6); FMA(a7); LOG(t1);
      }
      FMA(a0); FMA(a1); FMA(a2);
      if constexpr (KIND == K_MIX_MFMA) {  // + the kernel's MFMA share: 6 x 16x16x32 per 4 elements x 86 instr = 1 MFMA per ~57 VALU
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc, 0, 0, 0);
      }
    } else if constexpr (KIND == K_MIX_DEP) {
      // the same 8 : 35 mix as ONE dependent chain per 4 "elements" (4-way ILP, like 4 genes per lane)
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        FMA(a0); FMA(a1); FMA(a2); FMA(a3); EXP(a0); EXP(a1); FMA(a2); FMA(a3); LOG(a0); LOG(a1); FMA(a2); FMA(a3);
        FMA(a0); FMA(a1); FMA(a2); FMA(a3); FMA(a0); FMA(a1); FMA(a2); FMA(a3);
      }
      FMA(a0); FMA(a1); FMA(a2);
    } else if constexpr (KIND == K_FMA_SERIAL) {   // ONE dependent chain: every instruction needs its predecessor's result
#pragma unroll
      for (int r = 0; r < 32; ++r) { FMA(a0); }
    } else if constexpr (KIND == K_FMA_2CH) {      // two dependent chains, interleaved
#pragma unroll
      for (int r = 0; r < 16; ++r) { FMA(a0); FMA(a1); }
    } else if constexpr (KIND == K_TRANS_SERIAL) { // one dependent chain of transcendentals
#pragma unroll
      for (int r = 0; r < 16; ++r) { EXP(a0); LOG(a0); }
    } else if constexpr (KIND == K_PKFMA) {  // 16 packed fma = 32 fp32 fma
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(pm), "v"(pc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p1) : "v"(pm), "v"(pc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2) : "v"(pm), "v"(pc));
        asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p3) : "v"(pm), "v"(pc));
      }
    }
  }
  unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + t0 + t1 + t2 + t3 + acc[0] + p0[0] + p1[1] + p2[0] + p3[1];
  if (threadIdx.x == 0) { cyc[blockIdx.x] = c1 - c0; rt[blockIdx.x] = r1 - r0; }
  (void)smem;
}

template <int KIND>
static void run(const char* name, int instr_per_iter, float* out, unsigned long long* cyc, unsigned long long* rt) {
  const int iters = 20000;
  printf("%-34s", name);
  for (int w = 1; w <= 4; ++w) {
    const int threads = 256 * w;
    hipFuncSetAttribute(reinterpret_cast<const void*>(issue_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((issue_kernel<KIND>), dim3(256), dim3(threads), 100 * 1024, 0, out, cyc, rt, iters);
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> hc(256), hr(256);
    CK(hipMemcpy(hc.data(), cyc, 256 * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hr.data(), rt, 256 * 8, hipMemcpyDeviceToHost));
    std::sort(hc.begin(), hc.end()); std::sort(hr.begin(), hr.end());
    const double cycles = (double)hc[128], ghz = (double)hc[128] / ((double)hr[128] * 10.0);  // s_memrealtime ticks at 100 MHz
    const double per_simd = cycles / ((double)iters * instr_per_iter * w);   // cycles per wave-instruction per SIMD
    printf("  w=%d: %5.2f cyc/instr (%.2f GHz)", w, per_simd, ghz);
  }
  printf("\n");
}

int main() {
  float* out; unsigned long long *cyc, *rt;
  CK(hipMalloc(&out, 256 * 1024 * 4)); CK(hipMalloc(&cyc, 256 * 8)); CK(hipMalloc(&rt, 256 * 8));
  printf("cycles per wave-instruction per SIMD vs waves per SIMD (one 256*w-thread workgroup per CU, 256 CUs)\n");
  run<K_FMA>("v_fma_f32 x32 independent", 32, out, cyc, rt);
  run<K_PKFMA>("v_pk_fma_f32 x16 (=32 fma)", 16, out, cyc, rt);
  run<K_EXP>("v_exp/v_log x32", 32, out, cyc, rt);
  run<K_RCP>("v_rcp_f32 x32", 32, out, cyc, rt);
  run<K_MIX>("mix 8 trans : 35 fma (indep)", 43, out, cyc, rt);
  run<K_MIX_DEP>("mix 8 : 35, 4 dependent chains", 43, out, cyc, rt);
  run<K_MIX_MFMA>("mix 8 : 35 + 1 mfma16x16x32", 43, out, cyc, rt);
  run<K_FMA_SERIAL>("v_fma_f32 x32, ONE dependent chain", 32, out, cyc, rt);
  run<K_FMA_2CH>("v_fma_f32 x32, two chains", 32, out, cyc, rt);
  run<K_TRANS_SERIAL>("v_exp/v_log x32, ONE dep. chain", 32, out, cyc, rt);
  return 0;
}
