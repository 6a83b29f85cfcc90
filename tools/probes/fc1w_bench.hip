// Probe (dev tool): the LDS-DMA encoder fc1 weight-gradient kernel (spv_fc1.h: fc1_wgrad_dma_kernel<64|128>) alone:
// correctness against a naive fp32 kernel on the same bf16 operands and time per launch, alone and with two launches on two
// streams (the two groups of a training step run side by side).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -o fc1w_bench fc1w_bench.hip && ./fc1w_bench [B G]
#include "../../spvipes_amd/csrc/spv_fc1.h"
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

// dW[n][g] = sum_b dh[b][n] * A[rows[b]][g]
__global__ void naive_kernel(const unsigned short* dh, const unsigned short* A, long lda, const int* rows, int B, int G, float* out) {
  const int g = blockIdx.x * blockDim.x + threadIdx.x, n = blockIdx.y;
  if (g >= G) return;
  float s = 0.f;
  for (int b = 0; b < B; ++b) s += bf2f(dh[(long)b * 256 + n]) * bf2f(A[(long)rows[b] * lda + g]);
  out[(long)n * G + g] = s;
}

template <int BN>
static void run(const GemmParams& p, int G, int Kpad, const std::vector<float>& ref, float* dW, hipStream_t s2, const GemmParams& p2) {
  const int lds = fw_lds_bytes(BN, Kpad);
  if (lds > 160 * 1024) { printf("  WG_BN=%d: needs %d B of LDS, skipped\n", BN, lds); return; }
  void (*kfn)(GemmParams) = fc1_wgrad_dma_kernel<BN>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const int grid = (G + BN - 1) / BN;
  CK(hipMemset(dW, 0xFF, (size_t)256 * G * 4));
  hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, 0, p);
  CK(hipDeviceSynchronize());
  std::vector<float> h((size_t)256 * G);
  CK(hipMemcpy(h.data(), dW, h.size() * 4, hipMemcpyDeviceToHost));
  double maxerr = 0, maxref = 0; size_t bad = 0;
  for (size_t i = 0; i < h.size(); ++i) { const double e = fabs((double)h[i] - ref[i]); maxerr = std::max(maxerr, e); maxref = std::max(maxref, (double)fabs(ref[i])); if (!(e <= 2e-3 + 1e-3 * fabs(ref[i]))) ++bad; }
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 30;
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, 0, p);
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, 0, p);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, fl = 2.0 * Kpad * (double)G * 256;
  // two launches side by side (second stream, second output)
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipStream_t s1; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
  CK(hipEventRecord(e0, s1));
  for (int i = 0; i < reps; ++i) {
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, s1, p);
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), lds, s2, p2);
  }
  CK(hipStreamSynchronize(s2));
  CK(hipEventRecord(e1, s1)); CK(hipEventSynchronize(e1));
  float ms2; CK(hipEventElapsedTime(&ms2, e0, e1));
  printf("  WG_BN=%3d grid=%3d lds=%6d: max |err| %.3e (max |ref| %.3e), %zu outside tolerance;  %.1f us alone (%.0f TFLOP/s = %.1f %%), %.1f us per PAIR on two streams\n", BN, grid,
         lds, maxerr, maxref, bad, us, fl / us / 1e6, fl / us / 1e6 / 25.0, ms2 * 1e3 / reps);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, G = argc > 2 ? atoi(argv[2]) : 10000;
  const int NCELLS = 30000;
  const long lda = (G + 127) / 128 * 128;   // (the 96-gene tile's last workgroup reads past this: the kernel clamps)
  const int Kpad = (B + 63) / 64 * 64;
  std::mt19937 rng(5);
  std::uniform_real_distribution<float> uni(0.f, 1.f);
  std::normal_distribution<float> nrm(0.f, 1.f);
  std::vector<unsigned short> A((size_t)NCELLS * lda, 0), dh((size_t)Kpad * 256, 0);
  for (int c = 0; c < NCELLS; ++c)
    for (int g = 0; g < G; ++g) A[(size_t)c * lda + g] = uni(rng) < 0.8f ? 0 : f2bf_host(logf(1.f + (float)(1 + (int)(-3.f * logf(uni(rng) + 1e-6f)))));
  for (int b = 0; b < B; ++b)
    for (int n = 0; n < 256; ++n) dh[(size_t)b * 256 + n] = uni(rng) < 0.5f ? 0 : f2bf_host(0.01f * nrm(rng));
  std::vector<int> rows(B);
  for (auto& r : rows) r = (int)(uni(rng) * (NCELLS - 1));
  unsigned short* dA = upload(A); unsigned short* ddh = upload(dh); int* drows = upload(rows);
  unsigned short* dA2 = upload(A);   // the second group's image: the pair must not share cache lines
  float* dW = dalloc<float>((size_t)256 * G); float* dW_b = dalloc<float>((size_t)256 * G); float* dref = dalloc<float>((size_t)256 * G);
  hipLaunchKernelGGL(naive_kernel, dim3((G + 255) / 256, 256), dim3(256), 0, 0, ddh, dA, lda, drows, B, G, dref);
  CK(hipDeviceSynchronize());
  std::vector<float> ref((size_t)256 * G);
  CK(hipMemcpy(ref.data(), dref, ref.size() * 4, hipMemcpyDeviceToHost));
  GemmParams p{};
  p.A = ddh; p.lda = 256; p.B = dA; p.ldb = lda; p.rows = drows; p.n_cells = B; p.N = G; p.M = 256; p.K = B;
  p.C = dW; p.C2 = dW + (size_t)128 * G; p.c_split_row = 128; p.ldc = G;
  GemmParams p2 = p; p2.C = dW_b; p2.C2 = dW_b + (size_t)128 * G; p2.B = dA2;
  hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  printf("fc1_wgrad_dma B=%d G=%d (Kpad %d)\n", B, G, Kpad);
  run<128>(p, G, Kpad, ref, dW, s2, p2);
  run<96>(p, G, Kpad, ref, dW, s2, p2);
  run<64>(p, G, Kpad, ref, dW, s2, p2);
  return 0;
}
