// Probe (dev tool): how fast can a workgroup stream a row-gathered bf16 operand [rows][K] out of HBM, as a function of
// the bytes taken per row per step (the A-operand access pattern of the fc1 GEMM) and of the loads kept in flight?
//   hipcc --offload-arch=gfx950 -O3 -o gather_bw gather_bw.hip && ./gather_bw
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <numeric>
#include <random>

typedef __attribute__((ext_vector_type(4))) unsigned int u4v;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

// workgroup: ROWS rows x K-range [k0, k0 + kper); each step takes BK elements (BK * 2 bytes) per row; PF steps in flight
template <int ROWS, int BK, int PF>
__global__ __launch_bounds__(256) void gather_kernel(const unsigned short* X, long ld, const int* rows, int kper, unsigned* out) {
  constexpr int CH = BK / 8;                 // 16-byte chunks per row per step
  constexpr int PER = ROWS * CH / 256;       // chunks per thread per step
  static_assert(ROWS * CH % 256 == 0, "");
  const int tid = threadIdx.x;
  const int r0 = blockIdx.x * ROWS, k0 = blockIdx.y * kper;
  const unsigned short* src[PER];
#pragma unroll
  for (int i = 0; i < PER; ++i) {
    const int c = tid + 256 * i, r = c / CH, ch = c % CH;
    src[i] = X + (long)rows[r0 + r] * ld + k0 + 8 * ch;
  }
  u4v acc = {0, 0, 0, 0};
  u4v ring[PF][PER];
  const int nt = kper / BK;
#pragma unroll
  for (int p = 0; p < PF; ++p)
#pragma unroll
    for (int i = 0; i < PER; ++i) ring[p][i] = *reinterpret_cast<const u4v*>(src[i] + (long)min(p, nt - 1) * BK);
  for (int t = 0; t < nt; t += PF) {
#pragma unroll
    for (int p = 0; p < PF; ++p) {
#pragma unroll
      for (int i = 0; i < PER; ++i) {
        acc ^= ring[p][i];
        ring[p][i] = *reinterpret_cast<const u4v*>(src[i] + (long)min(t + PF + p, nt - 1) * BK);
      }
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

template <int ROWS, int BK, int PF>
static void run(const unsigned short* X, long ld, const int* rows, int B, int K, int ksplit, unsigned* out, const char* note) {
  const int kper = K / ksplit / (BK * PF) * (BK * PF);
  dim3 grid(B / ROWS, ksplit);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL((gather_kernel<ROWS, BK, PF>), grid, dim3(256), 0, 0, X, ld, rows, kper, out);
  CK(hipEventRecord(e0));
  const int reps = 24;
  for (int w = 0; w < reps; ++w)   // a different 4096-row set every launch (12 sets = 0.98 GB: nothing survives in the 256 MB MALL)
    hipLaunchKernelGGL((gather_kernel<ROWS, BK, PF>), grid, dim3(256), 0, 0, X, ld, rows + (w % 12) * B, kper, out);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double bytes = (double)B * kper * ksplit * 2;
  printf("rows/WG %3d  bytes/row/step %4d  in-flight steps %d  ksplit %2d  WGs %4d : %7.1f us  %6.2f TB/s  %s\n", ROWS, BK * 2, PF, ksplit,
         grid.x * grid.y, ms / reps * 1e3, bytes / (ms / reps * 1e-3) / 1e12, note);
}

int main() {
  const long n_cells = 50000, G = 10000, ld = 10016;
  const int B = 4096;
  unsigned short* X; int* rows; unsigned* out;
  CK(hipMalloc(&X, n_cells * ld * 2)); CK(hipMemset(X, 1, n_cells * ld * 2));
  CK(hipMalloc(&rows, 12 * B * 4)); CK(hipMalloc(&out, 64));
  std::vector<int> perm(n_cells); std::iota(perm.begin(), perm.end(), 0);
  std::mt19937 rng(1); std::shuffle(perm.begin(), perm.end(), rng);
  CK(hipMemcpy(rows, perm.data(), 12 * B * 4, hipMemcpyHostToDevice));
  const int K = 9984;
  run<128, 32, 4>(X, ld, rows, B, K, 8, out, "(current fc1 fwd pattern)");
  run<128, 32, 8>(X, ld, rows, B, K, 8, out, "");
  run<128, 64, 2>(X, ld, rows, B, K, 8, out, "");
  run<128, 64, 4>(X, ld, rows, B, K, 8, out, "");
  run<128, 128, 2>(X, ld, rows, B, K, 8, out, "");
  run<128, 128, 4>(X, ld, rows, B, K, 8, out, "");
  run<128, 256, 2>(X, ld, rows, B, K, 8, out, "");
  run<64, 256, 2>(X, ld, rows, B, K, 4, out, "");
  run<64, 512, 1>(X, ld, rows, B, K, 4, out, "");
  run<128, 32, 4>(X, ld, rows, B, K, 16, out, "");
  run<128, 64, 4>(X, ld, rows, B, K, 16, out, "");
  run<128, 128, 2>(X, ld, rows, B, K, 16, out, "");
  run<256, 64, 2>(X, ld, rows, B, K, 16, out, "");
  run<256, 64, 4>(X, ld, rows, B, K, 16, out, "");
  run<256, 128, 2>(X, ld, rows, B, K, 16, out, "");
  run<32, 256, 4>(X, ld, rows, B, K, 4, out, "");
  run<32, 512, 2>(X, ld, rows, B, K, 2, out, "");
  return 0;
}
