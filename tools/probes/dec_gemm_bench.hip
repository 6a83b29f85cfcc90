// Probe (dev tool): the LDS-DMA 320-column decoder GEMMs (spv_dec_gemm.h) alone: correctness against a naive fp32 kernel on the
// same bf16 operands, and time per launch.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o dec_gemm_bench dec_gemm_bench.hip && ./dec_gemm_bench [B G]
#include "../../spvipes_amd/csrc/spv_dec_gemm.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>

using namespace spv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static unsigned short f2bf_host(float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7FFF + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
template <typename T> static T* upload(const std::vector<T>& v) { T* p = dalloc<T>(v.size()); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }

__device__ __host__ inline size_t tiled_index(int cell, int gene, int T) {
  const int ct = cell >> 5, gt = gene >> 5, c = cell & 31, g = gene & 31;
  const int qq = g >> 3, hh = (g >> 2) & 1, j = g & 3;
  return ((size_t)ct * T + gt) * 1024 + qq * 256 + (c + 32 * hh) * 4 + j;
}
// dA[b][n] = sum_g dL[b][g] W[g][n]
__global__ void naive_dA(const unsigned short* dL, int T, const unsigned short* W, int B, int G, float* out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, b = blockIdx.y;
  if (n >= 320) return;
  float s = 0.f;
  for (int g = 0; g < G; ++g) s += bf2f(dL[tiled_index(b, g, T)]) * bf2f(W[(size_t)g * 320 + n]);
  out[(size_t)b * 320 + n] = s;
}
// dW[g][n] = sum_b dL[b][g] A[b][n]
__global__ void naive_dW(const unsigned short* dL, int T, const unsigned short* A, int B, int G, float* out) {
  const int n = blockIdx.x * blockDim.x + threadIdx.x, g = blockIdx.y;
  if (n >= 320) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += bf2f(dL[tiled_index(b, g, T)]) * bf2f(A[(size_t)b * 320 + n]);
  out[(size_t)g * 320 + n] = s;
}
__global__ void sum_slabs(const float* slabs, int splits, size_t n, float* out) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float s = 0.f;
  for (int k = 0; k < splits; ++k) s += slabs[(size_t)k * n + i];
  out[i] = s;
}

template <bool KM, bool FOUR>
static void run(const char* name, GemmParams p, int M, int K, const float* dref) {
  const int mtiles = (M + DG_BM - 1) / DG_BM, ktiles = (K + DG_BK - 1) / DG_BK;
  const int splits = std::max(1, std::min(256 / std::max(mtiles, 1), std::max(ktiles / 4, 1)));
  p.c_split_row = splits;
  p.k_per_split = (ktiles + splits - 1) / splits * DG_BK;
  p.slab_stride = (long)M * 320;
  float* slabs = dalloc<float>((size_t)splits * M * 320);
  float* out = dalloc<float>((size_t)M * 320);
  p.C = slabs;
  void (*kfn)(GemmParams) = FOUR ? dec_gemm320_dma4_kernel<KM> : dec_gemm320_dma_kernel<KM>;
  const int DG_LDS = FOUR ? D4_LDS_BYTES : DG_LDS_BYTES;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, DG_LDS));
  CK(hipMemset(slabs, 0xFF, (size_t)splits * M * 320 * 4));
  hipLaunchKernelGGL(kfn, dim3(mtiles * splits), dim3(512), DG_LDS, 0, p);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL(sum_slabs, dim3((unsigned)(((size_t)M * 320 + 255) / 256)), dim3(256), 0, 0, slabs, splits, (size_t)M * 320, out);
  CK(hipDeviceSynchronize());
  std::vector<float> h((size_t)M * 320), ref((size_t)M * 320);
  CK(hipMemcpy(h.data(), out, h.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0; size_t bad = 0;
  for (size_t i = 0; i < h.size(); ++i) { const double e = fabs((double)h[i] - ref[i]); maxerr = std::max(maxerr, e); maxref = std::max(maxref, (double)fabs(ref[i])); if (!(e <= 2e-3 + 1e-3 * fabs(ref[i]))) ++bad; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 30;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kfn, dim3(mtiles * splits), dim3(512), DG_LDS, 0, p);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kfn, dim3(mtiles * splits), dim3(512), DG_LDS, 0, p);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, fl = 2.0 * M * (double)K * 320;
  printf("  %s M=%d K=%d splits=%d grid=%d: max |err| %.3e (max |ref| %.3e), %zu outside tolerance;  %.1f us (%.0f TFLOP/s = %.1f %% of 2500)\n", name, M, K, splits,
         mtiles * splits, maxerr, maxref, bad, us, fl / us / 1e6, fl / us / 1e6 / 25.0);
  CK(hipFree(slabs)); CK(hipFree(out));
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, G = argc > 2 ? atoi(argv[2]) : 10000;
  const int Bp = (B + 127) / 128 * 128, Gp = (G + 255) / 256 * 256, T = Gp / 32;
  std::mt19937 rng(7);
  std::uniform_real_distribution<float> uni(0.f, 1.f);
  std::normal_distribution<float> nrm(0.f, 1.f);
  std::vector<unsigned short> dL((size_t)Bp * Gp, 0), W((size_t)Gp * 320, 0), A((size_t)Bp * 320, 0);
  for (int b = 0; b < B; ++b)
    for (int g = 0; g < G; ++g) dL[tiled_index(b, g, T)] = f2bf_host(0.05f * nrm(rng));
  for (int g = 0; g < G; ++g)
    for (int n = 0; n < 292; ++n) W[(size_t)g * 320 + n] = f2bf_host(0.1f * nrm(rng));
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < 292; ++n) A[(size_t)b * 320 + n] = f2bf_host(nrm(rng));
  unsigned short* ddL = upload(dL); unsigned short* dW = upload(W); unsigned short* dA = upload(A);
  float* refA = dalloc<float>((size_t)B * 320); float* refW = dalloc<float>((size_t)G * 320);
  hipLaunchKernelGGL(naive_dA, dim3(2, B), dim3(160), 0, 0, ddL, T, dW, B, G, refA);
  hipLaunchKernelGGL(naive_dW, dim3(2, G), dim3(160), 0, 0, ddL, T, dA, B, G, refW);
  CK(hipDeviceSynchronize());
  printf("dec_gemm320_dma B=%d G=%d (Bp %d, Gp %d)\n", B, G, Bp, Gp);
  GemmParams p{};
  p.A = ddL; p.tiles_inner = T; p.ldb = 320; p.ldc = 320; p.N = 320;
  { GemmParams q = p; q.B = dW; q.M = B; q.K = G; run<false, false>("d A_m 2 stages x 64", q, B, G, refA); run<false, true>("d A_m 4 stages x 32", q, B, G, refA); }
  { GemmParams q = p; q.B = dA; q.M = G; q.K = Bp; run<true, false>("d W_m 2 stages x 64", q, G, Bp, refW); run<true, true>("d W_m 4 stages x 32", q, G, Bp, refW); }
  return 0;
}
