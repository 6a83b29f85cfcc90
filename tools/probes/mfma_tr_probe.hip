// Probe (dev tool, not product): verifies on gfx950
//  (1) ds_read_b64_tr_b16 lane map, (2) mfma_f32_32x32x16_bf16 A/B/C lane maps with
//  natural ([m][k]) and transposed-read ([k][m]) LDS images.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
typedef __attribute__((ext_vector_type(4))) short s4;
typedef __attribute__((ext_vector_type(8))) short s8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((address_space(3))) s4 lds_s4;

__device__ inline unsigned short f2bf(float f) { unsigned u = __float_as_uint(f); return (unsigned short)((u + 0x7FFF + ((u >> 16) & 1)) >> 16); }

// A_km: [16][32] bf16 (k-major), B_kn: [16][32], A_mk: [32][16] natural. C1 = A_km^T*B_kn via tr reads; C2 = A_mk * B_kn (A natural)
__global__ void k(const unsigned short* A_km, const unsigned short* B_kn, const unsigned short* A_mk, float* C1, float* C2) {
  __shared__ __attribute__((aligned(16))) unsigned short sa[16*32], sb[16*32], sn[32*16];
  for (int i = threadIdx.x; i < 512; i += 64) { sa[i] = A_km[i]; sb[i] = B_kn[i]; sn[i] = A_mk[i]; }
  __syncthreads();
  int l = threadIdx.x, gi = l >> 4, q = (l & 15) >> 2, p = l & 3;
  int h = l >> 5;
  int n0 = 16 * (gi & 1);
  s8 af, bf, an;
  for (int t = 0; t < 2; ++t) {
    int k0 = 8 * h + 4 * t;
    s4 va = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)&sa[(k0 + q) * 32 + n0 + 4 * p]);
    s4 vb = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s4*)&sb[(k0 + q) * 32 + n0 + 4 * p]);
    for (int j = 0; j < 4; ++j) { af[4*t+j] = va[j]; bf[4*t+j] = vb[j]; }
  }
  // natural: lane (r=l&31,h) holds A[r][8h..8h+7]
  an = *(s8*)&sn[(l & 31) * 16 + 8 * h];
  f16v c1 = {0}, c2 = {0};
  c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, bf, c1, 0, 0, 0);
  c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(an, bf, c2, 0, 0, 0);
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * h, col = l & 31;
    C1[row * 32 + col] = c1[r]; C2[row * 32 + col] = c2[r];
  }
}
static unsigned short hbf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }
int main() {
  unsigned short A_km[512], B_kn[512], A_mk[512]; float Af[32][16], Bf[16][32];
  srand(1);
  for (int m = 0; m < 32; ++m) for (int kk = 0; kk < 16; ++kk) { Af[m][kk] = (float)(rand() % 17 - 8); A_mk[m*16+kk] = hbf(Af[m][kk]); A_km[kk*32+m] = hbf(Af[m][kk]); }
  for (int kk = 0; kk < 16; ++kk) for (int n = 0; n < 32; ++n) { Bf[kk][n] = (float)(rand() % 13 - 6 + (n > kk)); B_kn[kk*32+n] = hbf(Bf[kk][n]); }
  unsigned short *dA, *dB, *dN; float *dC1, *dC2;
  (void)hipMalloc(&dA, 1024); (void)hipMalloc(&dB, 1024); (void)hipMalloc(&dN, 1024); (void)hipMalloc(&dC1, 4096); (void)hipMalloc(&dC2, 4096);
  (void)hipMemcpy(dA, A_km, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dB, B_kn, 1024, hipMemcpyHostToDevice); (void)hipMemcpy(dN, A_mk, 1024, hipMemcpyHostToDevice);
  k<<<1, 64>>>(dA, dB, dN, dC1, dC2);
  float C1[1024], C2[1024]; (void)hipMemcpy(C1, dC1, 4096, hipMemcpyDeviceToHost); (void)hipMemcpy(C2, dC2, 4096, hipMemcpyDeviceToHost);
  int bad1 = 0, bad2 = 0;
  for (int m = 0; m < 32; ++m) for (int n = 0; n < 32; ++n) { float s = 0; for (int kk = 0; kk < 16; ++kk) s += Af[m][kk] * Bf[kk][n]; bad1 += (C1[m*32+n] != s); bad2 += (C2[m*32+n] != s); }
  printf("tr-read A^T.B mismatches: %d / 1024\nnatural A.B (B via tr) mismatches: %d / 1024\n", bad1, bad2);
  return (bad1 || bad2) ? 1 : 0;
}
