// Encoder fc1 forward for the resident bf16 log1p image, N1 = 2H = 256 output columns (the reference's default H = 128):
//
//   slab[split][cell][col] = sum_{k in split} log1p(X)[rows[cell]][k] * W[col][k]      (module/spVIPESmodule.py:428-435,
//                                                                                       nn/networks.py:119, both encoders)
//
// The shape is tall and skinny (M = cells 4096, N = 256, K = genes 10 000 .. 30 000): the whole N extent is one
// workgroup tile, K is split over workgroups, and the kernel is co-bound by the HBM stream of the gathered A rows
// (B x G x 2 bytes) and the MFMA rate.  What this kernel does differently from the generic gemm_kernel (spv_gemm.h), whose
// register-staged loop spent more than half of every K tile outside the MFMAs:
//   * both operands are staged by LDS-DMA (global_load_lds_dwordx4: no VGPR round trip, no ds_write), three 48 KiB tile
//     buffers, two tiles in flight behind a counted s_waitcnt vmcnt and ONE raw s_barrier per K tile;
//   * the A rows are gathered by the DMA itself (its source address is per lane): 8 lanes fetch one row's 128 bytes;
//   * the LDS image is the DMA's lane-linear one (row-major, 128-byte rows, no padding); bank conflicts of the 16-byte
//     fragment reads are removed by XOR-swizzling the 16-byte chunk index with (row >> 1) & 7 on the SOURCE address and
//     again on the read (a row's eight chunks still come from one 128-byte line);
//   * 8 waves (2 per SIMD) as 2 (M) x 4 (N), wave tile 64 x 64 = 2 x 2 MFMA 32x32x16 tiles;
//   * the fp32 partial slabs are written in accumulator-tile order (16-byte stores, 1 KiB per wave instruction) and summed,
//     biased and rectified by fc1_epilogue_tiled_kernel.
#pragma once
#include "spv_common.h"
#include "spv_gemm.h"

namespace spv {

constexpr int F1_BM = 128, F1_BN = 256, F1_BK = 64, F1_NBUF = 3;
constexpr int F1_A_BYTES = F1_BM * F1_BK * 2, F1_B_BYTES = F1_BN * F1_BK * 2, F1_STAGE = F1_A_BYTES + F1_B_BYTES;
constexpr int F1_LDS_BYTES = F1_STAGE * F1_NBUF;   // 147 456 B: one workgroup per CU
constexpr int F1_PIECES_PER_WAVE = (F1_A_BYTES + F1_B_BYTES) / 1024 / 8;   // 6 LDS-DMA pieces (1 KiB each) per wave and tile

typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(1))) const unsigned char glb_byte;

// p.A = bf16 image [n_cells_total][lda] (zero padded to a multiple of 64 genes), p.rows = minibatch row index (nullable),
// p.B = W bf16 [256][ldb] (zero padded likewise), p.C = slabs in tile order, p.M = cells in the minibatch, p.K = genes,
// p.k_per_split multiple of 64.  grid = (ceil(M / 128), 1, splits), 512 threads.
__global__ __launch_bounds__(512) void fc1_fwd_dma_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char f1_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the LDS-DMA destination must be
  const int wm = wave >> 2, wn = wave & 3;
  const int m0 = blockIdx.x * F1_BM;
  const int split = blockIdx.z;
  const int kbeg = split * p.k_per_split;
  const int Kpad = (p.K + F1_BK - 1) / F1_BK * F1_BK;
  int kend = kbeg + p.k_per_split;
  if (kend > Kpad) kend = Kpad;
  const int ntiles = (kend - kbeg) / F1_BK;   // may be <= 0 for a trailing split: its slab is zeros

  // ---- this lane's six DMA source rows (fixed over the K loop) ----------------------------------------------------------
  const int lr = lane >> 3, pos = lane & 7;
  const glb_byte* src[F1_PIECES_PER_WAVE];
  int dst_off[F1_PIECES_PER_WAVE];   // wave-uniform byte offset of the piece inside a stage
#pragma unroll
  for (int i = 0; i < 2; ++i) {   // A pieces 2 wave + i: rows 8 (2 wave + i) + lr of the tile
    const int piece = 2 * wave + i, row = 8 * piece + lr;
    int cell = m0 + row;
    if (cell > p.M - 1) cell = p.M - 1;   // rows beyond the minibatch re-read its last row; their outputs are never used
    const long ridx = p.rows ? (long)p.rows[cell] : (long)cell;
    const int c = pos ^ ((row >> 1) & 7);
    src[i] = (glb_byte*)(p.A) + (ridx * p.lda + kbeg + 8 * c) * 2;
    dst_off[i] = piece * 1024;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // B pieces 4 wave + i: rows (output columns) 8 (4 wave + i) + lr of W
    const int piece = 4 * wave + i, n = 8 * piece + lr;
    const int c = pos ^ ((n >> 1) & 7);
    src[2 + i] = (glb_byte*)(p.B) + ((long)n * p.ldb + kbeg + 8 * c) * 2;
    dst_off[2 + i] = F1_A_BYTES + piece * 1024;
  }
  lds_byte* const lds = (lds_byte*)(f1_smem);
  auto issue = [&](int t) {   // tile t of this split -> buffer t % 3
    const int stage = (t % F1_NBUF) * F1_STAGE;
#pragma unroll
    for (int i = 0; i < F1_PIECES_PER_WAVE; ++i)
      __builtin_amdgcn_global_load_lds(src[i] + (long)t * (F1_BK * 2), lds + stage + dst_off[i], 16, 0, 0);
  };

  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);
  const int r = lane & 31, h = lane >> 5, sw = (r >> 1) & 7;
  const int a_row_off = (wm * 64 + r) * 128, b_row_off = F1_A_BYTES + (wn * 64 + r) * 128;
  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed once all but this wave's youngest six DMAs (tile t + 1) are done; the barrier extends that to the
    // other waves' pieces and also says everybody is done reading buffer (t + 2) % 3 (= tile t - 1's)
    if (t + 1 < ntiles) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (t + 2 < ntiles) issue(t + 2);
    const unsigned char* st = f1_smem + (t % F1_NBUF) * F1_STAGE;
#pragma unroll
    for (int ks = 0; ks < F1_BK / 16; ++ks) {
      const int cpos = ((2 * ks + h) ^ sw) * 16;
      s8v a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const s8v*>(st + a_row_off + i * 32 * 128 + cpos);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const s8v*>(st + b_row_off + j * 32 * 128 + cpos);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
  }

  // ---- partial slab in accumulator-tile order: S[split][row / 32][col / 32][qq][lane][4] ------------------------------------
  const long mtiles = gridDim.x * (F1_BM / 32);
  float* slab = p.C + (long)split * mtiles * (F1_BN / 32) * 1024;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long tm = (long)blockIdx.x * (F1_BM / 32) + wm * 2 + i, tn = wn * 2 + j;
      float* o = slab + (tm * (F1_BN / 32) + tn) * 1024 + lane * 4;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        *reinterpret_cast<f4v*>(o + qq * 256) = f4v{acc[i][j][4 * qq], acc[i][j][4 * qq + 1], acc[i][j][4 * qq + 2], acc[i][j][4 * qq + 3]};
    }
}

// h1[cell][col] = relu(bias[col] + sum_splits S[split][...]), library from the data set's table (spv_prepare_log1p).
// One thread per (tile, qq, lane): 16 bytes of every split's slab in, four floats (rows jj + 8 qq + 4 h of the tile) out.
__global__ __launch_bounds__(256) void fc1_epilogue_tiled_kernel(const float* slabs, int splits, long slab_elems, int M, const float* bias,
                                                                 const float* bias2, int n_first, float* h1, float* library,
                                                                 const float* library_all, const int* rows) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index inside one slab
  if (idx * 4 < slab_elems) {
    const long tile = idx >> 8;
    const int qq = (int)(idx >> 6) & 3, lane = (int)idx & 63, r = lane & 31, h = lane >> 5;
    const long tm = tile / (F1_BN / 32);
    const int tn = (int)(tile % (F1_BN / 32));
    f4v s = *reinterpret_cast<const f4v*>(slabs + idx * 4);
    for (int k = 1; k < splits; ++k) {
      const f4v v = *reinterpret_cast<const f4v*>(slabs + (long)k * slab_elems + idx * 4);
      s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    const int col = tn * 32 + r;
    const float bv = (bias2 != nullptr && col >= n_first) ? bias2[col - n_first] : bias[col];
    const long row0 = tm * 32 + 8 * qq + 4 * h;
#pragma unroll
    for (int jj = 0; jj < 4; ++jj)
      if (row0 + jj < M) h1[(row0 + jj) * F1_BN + col] = fmaxf(s[jj] + bv, 0.f);   // relu(fc1(x)), nn/networks.py:119
  }
  if (idx < M) library[idx] = library_all[rows ? rows[idx] : (int)idx];   // log(sum_g log1p(x)), module/spVIPESmodule.py:435
}

}  // namespace spv
