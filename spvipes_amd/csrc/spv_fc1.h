// Encoder fc1 forward for the resident 16-bit log1p image, N1 = 2H = 256 output columns (the reference's default H = 128).
// Operand words are IEEE f16 (spv_common.h: 11 significant bits instead of bf16's 8 at the same MFMA rate; the weight image
// carries W * SPV_FC1_W_SCALE, undone by the epilogue's acc_scale; the dh image of the weight gradient a per-step power of two,
// undone through GemmParams::out_scale):
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

#ifndef F1_SPREAD
#define F1_SPREAD 1
#endif
#ifndef F1_PIPE
#define F1_PIPE 1   // measured at C2 (tools/probes/fc1_bench.hip): 28.1 -> 26.7 us; G = 20 000: 50.0 -> 48.2 us
#endif
constexpr int F1_BM = 128, F1_BN = 256, F1_BK = 64, F1_NBUF = 3;
constexpr int F1_A_BYTES = F1_BM * F1_BK * 2, F1_B_BYTES = F1_BN * F1_BK * 2, F1_STAGE = F1_A_BYTES + F1_B_BYTES;
constexpr int F1_LDS_BYTES = F1_STAGE * F1_NBUF;   // 147 456 B: one workgroup per CU
constexpr int F1_PIECES_PER_WAVE = (F1_A_BYTES + F1_B_BYTES) / 1024 / 8;   // 6 LDS-DMA pieces (1 KiB each) per wave and tile

typedef __attribute__((address_space(3))) unsigned char lds_byte;
typedef __attribute__((address_space(1))) const unsigned char glb_byte;

// Target builtins behind plain (non-template) device functions: a kernel TEMPLATE that names them directly cannot be instantiated in
// the host pass of the compilation (the builtins do not exist there), and its launch stub silently goes missing.
__device__ __forceinline__ void dma16(const glb_byte* src, lds_byte* dst) { __builtin_amdgcn_global_load_lds(src, dst, 16, 0, 0); }
__device__ __forceinline__ int uniform_wave_id() { return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)); }
__device__ __forceinline__ void raw_barrier() { __builtin_amdgcn_s_barrier(); }
// Transposed LDS reads as inline asm.  Through the builtin (__builtin_amdgcn_ds_read_tr16_b64_v4i16) hipcc (ROCm 7.2) puts an
// s_waitcnt vmcnt(0) in front of the first transposed read of every K tile -- it cannot tell the read from the LDS-DMA
// writes still in flight -- which drains the two-tile DMA pipeline once per tile.  The asm form is invisible to that pass:
// the reads of one k-step are issued, then ONE wait statement that names every destination (so no consumer is scheduled
// above it; cdna_hip_programming.md 5.7 form (ii)).  Byte addresses inside LDS, 8-byte aligned.
__device__ __forceinline__ unsigned lds_addr_of(const unsigned char* p) { return (unsigned)(size_t)((lds_byte*)p); }
__device__ __forceinline__ void tr_issue(s4v& d, unsigned addr) { asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(d) : "v"(addr)); }
__device__ __forceinline__ s8v join8(const s4v& v0, const s4v& v1) { return s8v{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]}; }

// p.A = bf16 image [n_cells_total][lda] (zero padded to a multiple of 64 genes), p.rows = minibatch row index (nullable),
// p.B = W bf16 [256][ldb] (zero padded likewise), p.C = slabs in tile order, p.M = cells in the minibatch, p.K = genes,
// p.k_per_split multiple of 64, p.c_split_row = number of K splits.  grid = ceil(M / 128) * splits (1-D), 512 threads.
// SPLIT ("fp32" mode): both operands come as bf16 hi / lo planes (p.A: the resident image of spv_prepare_log1p_split, planes interleaved
// in blocks of 32 k values; p.B / p.B_lo: the two weight images).  A K tile is then 32 deep and an LDS row holds [32 hi words | 32 lo words] of its 32 k values -- the
// DMA lanes of the upper four 16-byte chunks simply fetch from the lo plane -- so the stage image, the piece count, the swizzle and
// the counted waits are those of the one-plane kernel; each of the two k-steps of a tile reads hi and lo fragments (chunks 2 ks + h
// and 4 + 2 ks + h) and issues three bf16 MFMAs per fragment pair (hi*lo, lo*hi, hi*hi).
template <bool SPLIT>
__device__ __forceinline__ void fc1_fwd_dma_body(const GemmParams& p, const int blk, const int ntile, unsigned char* f1_smem) {
  constexpr int TK = SPLIT ? F1_BK / 2 : F1_BK;   // k values per tile
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // provably wave-uniform: the LDS-DMA destination must be
  const int wm = wave >> 2, wn = wave & 3;
  // 1-D grid, split fastest: workgroups are dealt round-robin over the 8 XCDs (MI355X_MICROARCH.md), so with 8 splits every
  // XCD works on ONE K range and its slice of W (256 x k_per_split bf16 = 640 KB at C2) stays resident in that XCD's L2
  const int split = blk % p.c_split_row, mtile = blk / p.c_split_row;   // (c_split_row carries the split count)
  const int m0 = mtile * F1_BM;
  const int kbeg = split * p.k_per_split;
  const int Kpad = (p.K + F1_BK - 1) / F1_BK * F1_BK;
  int kend = kbeg + p.k_per_split;
  if (kend > Kpad) kend = Kpad;
  const int ntiles = (kend - kbeg) / TK;   // may be <= 0 for a trailing split: its slab is zeros

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
    // (SPLIT: the image interleaves the planes in blocks of 32 k values, [32 hi words | 32 lo words]: a tile's row is ONE 128-byte line
    //  here too -- with the planes in separate halves of the row the same kernel ran at 0.58 instead of 0.4x ms at C5: 64-byte requests)
    src[i] = (glb_byte*)(p.A) + (ridx * p.lda + (SPLIT ? 2 : 1) * kbeg + 8 * c) * 2;
    dst_off[i] = piece * 1024;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {   // B pieces 4 wave + i: rows (output columns) 8 (4 wave + i) + lr of W
    const int piece = 4 * wave + i, n = 8 * piece + lr + ntile * F1_BN;   // (ntile: which 256 of the N1 = 256 or 512 output columns)
    const int c = pos ^ ((n >> 1) & 7);
    if constexpr (SPLIT) src[2 + i] = (glb_byte*)((c >> 2) ? p.B_lo : p.B) + ((long)n * p.ldb + kbeg + 8 * (c & 3)) * 2;
    else src[2 + i] = (glb_byte*)(p.B) + ((long)n * p.ldb + kbeg + 8 * c) * 2;
    dst_off[2 + i] = F1_A_BYTES + piece * 1024;
  }
  lds_byte* const lds = (lds_byte*)(f1_smem);
  auto issue2 = [&](int t, int i0) {   // pieces i0, i0 + 1 of tile t of this split -> buffer t % 3
    const int stage = (t % F1_NBUF) * F1_STAGE;
#pragma unroll
    for (int i = i0; i < i0 + 2; ++i)
      __builtin_amdgcn_global_load_lds(src[i] + (long)t * ((SPLIT && i < 2) ? F1_BK * 2 : TK * 2), lds + stage + dst_off[i], 16, 0, 0);
  };
  auto issue = [&](int t) { issue2(t, 0); issue2(t, 2); issue2(t, 4); };

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
    const bool more = t + 2 < ntiles;
    const unsigned char* st = f1_smem + (t % F1_NBUF) * F1_STAGE;
    if constexpr (SPLIT) {
      // every fragment of the tile is requested BEFORE tile t + 2's DMA goes out: hipcc treats an LDS read behind an LDS-DMA of the same
      // block as a possible alias and drains the DMA queue in front of it (s_waitcnt vmcnt(0): no prefetch left at all -- first version)
      s8v a_hi[2][2], a_lo[2][2], b_hi[2][2], b_lo[2][2];
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        const int chi = ((2 * ks + h) ^ sw) * 16, clo = ((4 + 2 * ks + h) ^ sw) * 16;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          a_hi[ks][i] = *reinterpret_cast<const s8v*>(st + a_row_off + i * 32 * 128 + chi);
          a_lo[ks][i] = *reinterpret_cast<const s8v*>(st + a_row_off + i * 32 * 128 + clo);
          b_hi[ks][i] = *reinterpret_cast<const s8v*>(st + b_row_off + i * 32 * 128 + chi);
          b_lo[ks][i] = *reinterpret_cast<const s8v*>(st + b_row_off + i * 32 * 128 + clo);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more) issue(t + 2);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            acc[i][j] = mfma32(a_hi[ks][i], b_lo[ks][j], acc[i][j]);
            acc[i][j] = mfma32(a_lo[ks][i], b_hi[ks][j], acc[i][j]);
            acc[i][j] = mfma32(a_hi[ks][i], b_hi[ks][j], acc[i][j]);
          }
      continue;
    }
#if F1_PIPE
    // fragments of k-step ks + 1 are requested before the MFMAs of k-step ks; tile t + 2's six DMA pieces go out two at a
    // time behind the MFMAs of the first three k-steps.  Order pinned with sched_group_barrier (DS read 0x100, MFMA 0x8,
    // VMEM 0x10): left alone the scheduler hoists all six DMAs to the top of the tile, where both waves of every SIMD
    // issue them in lockstep right after the barrier while the matrix pipe idles.
    s8v fa[2][2], fb[2][2];
    auto frags = [&](int ks, s8v (&a)[2], s8v (&b)[2]) {
      const int cpos = ((2 * ks + h) ^ sw) * 16;
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const s8v*>(st + a_row_off + i * 32 * 128 + cpos);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const s8v*>(st + b_row_off + j * 32 * 128 + cpos);
    };
    frags(0, fa[0], fb[0]);
#pragma unroll
    for (int ks = 0; ks < F1_BK / 16; ++ks) {
      if (ks + 1 < F1_BK / 16) frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32h(fa[ks & 1][i], fb[ks & 1][j], acc[i][j]);
      if (more && ks < 3) issue2(t + 2, 2 * ks);
    }
    __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
#pragma unroll
    for (int ks = 0; ks < F1_BK / 16; ++ks) {
      if (ks + 1 < F1_BK / 16) __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      if (ks < 3) __builtin_amdgcn_sched_group_barrier(0x010, 2, 0);
    }
#else
#pragma unroll
    for (int ks = 0; ks < F1_BK / 16; ++ks) {
      // tile t + 2's six DMA pieces go out two at a time BEHIND the first three k-steps' MFMAs (F1_SPREAD), not in one
      // burst right after the barrier while both waves of every SIMD leave the matrix pipe idle
      if (F1_SPREAD ? (ks > 0) : (ks == 0)) {
        if (more) { if (F1_SPREAD) issue2(t + 2, 2 * (ks - 1)); else issue(t + 2); }
      }
      const int cpos = ((2 * ks + h) ^ sw) * 16;
      s8v a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const s8v*>(st + a_row_off + i * 32 * 128 + cpos);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const s8v*>(st + b_row_off + j * 32 * 128 + cpos);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32h(a[i], b[j], acc[i][j]);
      if (F1_SPREAD) __builtin_amdgcn_sched_barrier(0);   // keep the DMA pieces where they are written, between the k-steps
    }
#endif
  }

  // ---- partial slab in accumulator-tile order: S[split][row / 32][col / 32][qq][lane][4] ------------------------------------
  const long mtiles = (long)((p.M + F1_BM - 1) / F1_BM) * (F1_BM / 32);
  const int NT = p.N / 32;   // column tiles of a slab row
  float* slab = p.C + (long)split * mtiles * NT * 1024;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long tm = (long)mtile * (F1_BM / 32) + wm * 2 + i, tn = ntile * (F1_BN / 32) + wn * 2 + j;
      float* o = slab + (tm * NT + tn) * 1024 + lane * 4;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq)
        *reinterpret_cast<f4v*>(o + qq * 256) = f4v{acc[i][j][4 * qq], acc[i][j][4 * qq + 1], acc[i][j][4 * qq + 2], acc[i][j][4 * qq + 3]};
    }
}

__global__ __launch_bounds__(512) void fc1_fwd_dma_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char f1_smem_[];
  fc1_fwd_dma_body<false>(p, blockIdx.x, blockIdx.y, f1_smem_);
}
__global__ __launch_bounds__(512) void fc1_fwd_dma_split_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char f1_smem_[];
  fc1_fwd_dma_body<true>(p, blockIdx.x, blockIdx.y, f1_smem_);
}
__global__ __launch_bounds__(512) void fc1_fwd_dma_split_pair_kernel(GemmParams p0, GemmParams p1, int n0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char f1_smem_[];
  const bool first = (int)blockIdx.x < n0;
  const GemmParams p = first ? p0 : p1;
  fc1_fwd_dma_body<true>(p, first ? (int)blockIdx.x : (int)blockIdx.x - n0, blockIdx.y, f1_smem_);
}
// both groups of a step in ONE grid (workgroups 0 .. n0 - 1: the first group's): one launch instead of two launches on two streams
// whose 2 x 256 one-per-CU workgroups can only run one after the other anyway -- without the fork / join of the graph branches
__global__ __launch_bounds__(512) void fc1_fwd_dma_pair_kernel(GemmParams p0, GemmParams p1, int n0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char f1_smem_[];
  const bool first = (int)blockIdx.x < n0;   // (workgroup-uniform)
  const GemmParams p = first ? p0 : p1;
  fc1_fwd_dma_body<false>(p, first ? (int)blockIdx.x : (int)blockIdx.x - n0, blockIdx.y, f1_smem_);
}

// h1[cell][col] = relu(bias[col] + sum_splits S[split][...]), library from the data set's table (spv_prepare_log1p).
// One thread per (tile, qq, lane): 16 bytes of every split's slab in, four floats (rows jj + 8 qq + 4 h of the tile) out.
struct Fc1EpiArgs { const float* slabs; int splits; long slab_elems; int M; int N1; const float* bias; const float* bias2; int n_first; float* h1; float* library;
                    const float* library_all; const int* rows; float acc_scale; const float* cov; const int* cov_idx; };
__device__ __forceinline__ void fc1_epilogue_tiled_body(const float* slabs, int splits, long slab_elems, int M, const int N1, const float* bias,
                                                        const float* bias2, int n_first, float* h1, float* library,
                                                        const float* library_all, const int* rows, const float acc_scale,
                                                        const float* cov, const int* cov_idx) {
  const long idx = (long)blockIdx.x * blockDim.x + threadIdx.x;   // float4 index inside one slab
  if (idx * 4 < slab_elems) {
    const long tile = idx >> 8;
    const int qq = (int)(idx >> 6) & 3, lane = (int)idx & 63, r = lane & 31, h = lane >> 5;
    const long tm = tile / (N1 / 32);
    const int tn = (int)(tile % (N1 / 32));
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
      if (row0 + jj < M) {
        // one-hot batch covariates appended to the layer's input (nn/networks.py:105-119) = one more per-cell bias row: cov[batch][col]
        const float cv = cov ? cov[(long)cov_idx[row0 + jj] * N1 + col] : 0.f;
        h1[(row0 + jj) * N1 + col] = fmaxf(s[jj] * acc_scale + bv + cv, 0.f);   // relu(fc1(x)), nn/networks.py:119
      }
  }
  if (idx < M) library[idx] = library_all[rows ? rows[idx] : (int)idx];   // log(sum_g log1p(x)), module/spVIPESmodule.py:435
}
__global__ __launch_bounds__(256) void fc1_epilogue_tiled_kernel(const float* slabs, int splits, long slab_elems, int M, int N1, const float* bias,
                                                                 const float* bias2, int n_first, float* h1, float* library,
                                                                 const float* library_all, const int* rows, float acc_scale,
                                                                 const float* cov, const int* cov_idx) {
  fc1_epilogue_tiled_body(slabs, splits, slab_elems, M, N1, bias, bias2, n_first, h1, library, library_all, rows, acc_scale, cov, cov_idx);
}
__global__ __launch_bounds__(256) void fc1_epilogue_tiled_pair_kernel(Fc1EpiArgs a0, Fc1EpiArgs a1) {   // blockIdx.y = group
  const Fc1EpiArgs a = blockIdx.y ? a1 : a0;
  fc1_epilogue_tiled_body(a.slabs, a.splits, a.slab_elems, a.M, a.N1, a.bias, a.bias2, a.n_first, a.h1, a.library, a.library_all, a.rows, a.acc_scale, a.cov, a.cov_idx);
}

// ---- fc1 weight gradient:  dW[n1][gene] = sum_cell dh[cell][n1] * log1p(X)[rows[cell]][gene]   (backward of nn/networks.py:119) -------
// M = 256 output units (both encoders), N = genes, K = cells: both operands are k-major in memory (dh image [cell][256],
// gathered image rows [cell][gene]), so the MFMA fragments come out of LDS by transposed reads (ds_read_b64_tr_b16).  Same
// skeleton as the forward kernel: LDS-DMA staging into three tile buffers (one K tile = 64 cells), counted vmcnt, one
// barrier per tile, 8 waves as 4 (M) x 2 (N).  One workgroup owns ALL 256 rows x WG_BN genes over the whole K range: no
// split-K, the fp32 result goes straight to the two encoders' gradient arrays.  The gather's row indices (they change with
// every K tile) are staged once into an LDS table, so that the loop issues no register-destination global load (hipcc would
// drain the DMA queue with vmcnt(0) at its first use).
//   LDS images are lane-linear copies of the global rows (512-byte dh rows, 2 WG_BN-byte gene rows); the four k rows a
//   transposed read touches would fall on the same banks, so the 64-byte granule index inside a row is XOR-ed with
//   (row & 3) (128-byte rows: (row >> 1) & 1) on the DMA's source address and again on the read.  192-byte rows (96 genes)
//   need no swizzle: four consecutive rows start 0 / 192 / 128 / 64 bytes into the 256-byte bank period.
//   WG_BN = 96 exists for the chip's round structure: at G = 10 000 it makes 105 workgroups per group, so the two groups' launches
//   (two streams, one workgroup per CU) are resident TOGETHER in one round of 256 CUs, where 2 x 157 workgroups of 64 genes take two.
constexpr int FW_BM = 256, FW_BK = 64, FW_NBUF = 3;
template <int WG_BN>
struct FwCfg {
  static constexpr int A_BYTES = FW_BK * FW_BM * 2, B_ROW = WG_BN * 2, B_BYTES = FW_BK * B_ROW, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_PIECES = A_BYTES / 1024 / 8;                 // per wave
  static constexpr int B_PIECES_ALL = B_BYTES / 1024;                 // 8 (64 genes), 12 (96), 16 (128) per stage
  static constexpr int B_PIECES = (B_PIECES_ALL + 7) / 8;             // per wave; 96 genes: waves 0..3 issue two, waves 4..7 one
  static constexpr int PIECES = A_PIECES + B_PIECES;
  // 8 waves as 4 (M) x 2 (N); the 96-gene tile (three 32-gene MFMA tiles) as 8 (M) x 1 (N)
  static constexpr int WAVES_N = (WG_BN == 96) ? 1 : 2, WAVES_M = 8 / WAVES_N;
  static constexpr int TM = FW_BM / WAVES_M / 32, TN = WG_BN / WAVES_N / 32;
  static_assert(WG_BN == 64 || WG_BN == 96 || WG_BN == 128, "gene tile");
};
__host__ __device__ constexpr int fw_lds_bytes(int wg_bn, int kpad) { return (FW_BK * FW_BM * 2 + FW_BK * wg_bn * 2) * FW_NBUF + kpad * 4; }

// p.A = dh image bf16 [Kpad][256] (rows >= n_cells zero), p.B = image [cells_total][ldb], p.rows, p.n_cells = K = cells of the
// minibatch, p.N = genes, p.C / p.C2 = dW of the first / second 128 rows, p.ldc.  grid = ceil(N / WG_BN), 512 threads,
// dynamic LDS = fw_lds_bytes(WG_BN, Kpad).
template <int WG_BN>
__device__ __forceinline__ void fc1_wgrad_dma_body(const GemmParams& p, const int blk, const int mtile, unsigned char* fw_smem) {
  typedef FwCfg<WG_BN> Cfg;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_wave_id();
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int n0 = blk * WG_BN;
  const int Kpad = (p.n_cells + FW_BK - 1) / FW_BK * FW_BK, ntiles = Kpad / FW_BK;
  int* rowtab = reinterpret_cast<int*>(fw_smem + Cfg::STAGE * FW_NBUF);
  for (int k = tid; k < Kpad; k += 512) {
    const int c = k < p.n_cells ? k : p.n_cells - 1;   // padding cells re-read the last row: their dh rows are zero
    rowtab[k] = p.rows ? p.rows[c] : c;
  }
  __syncthreads();

  lds_byte* const lds = (lds_byte*)(fw_smem);
  // A pieces (2 dh rows of 512 B each): wave w owns pieces 4w .. 4w + 3 = rows 8w .. 8w + 7 of the K tile
  const glb_byte* srcA[Cfg::A_PIECES];
#pragma unroll
  for (int i = 0; i < Cfg::A_PIECES; ++i) {
    const int piece = Cfg::A_PIECES * wave + i, row = 2 * piece + (lane >> 5), ch = lane & 31;
    const int chs = (((ch >> 2) ^ (row & 3)) << 2) | (ch & 3);
    srcA[i] = (glb_byte*)(p.A) + ((long)row * p.lda + mtile * FW_BM) * 2 + chs * 16;   // (mtile: which 256 of the N1 = 256 or 512 units; lda = N1)
  }
  // B pieces: rows of 2 WG_BN bytes; wave w owns rows 8w .. 8w + 7 as well
  int browB[Cfg::B_PIECES], bcolB[Cfg::B_PIECES];
#pragma unroll
  for (int i = 0; i < Cfg::B_PIECES; ++i) {
    const int piece = Cfg::B_PIECES * wave + i;
    if constexpr (WG_BN == 96) {   // piece = wave + 8 i (the second one only exists for waves 0..3); pieces cross rows
      const int o = (wave + 8 * i) * 1024 + lane * 16;
      browB[i] = (o / Cfg::B_ROW) & (FW_BK - 1);
      int cb = o % Cfg::B_ROW;
      if ((long)n0 + cb / 2 + 8 > p.ldb) cb = 0;   // the last workgroup's columns beyond the image row: any in-bounds bytes do (never stored)
      bcolB[i] = cb;
    } else if constexpr (WG_BN == 128) {
      const int row = 4 * piece + (lane >> 4), ch = lane & 15;
      browB[i] = row; bcolB[i] = ((((ch >> 2) ^ (row & 3)) << 2) | (ch & 3)) * 16;
    } else {
      const int row = 8 * piece + (lane >> 3), ch = lane & 7;
      browB[i] = row; bcolB[i] = ((((ch >> 2) ^ ((row >> 1) & 1)) << 2) | (ch & 3)) * 16;
    }
  }
  const glb_byte* const Bbase = (glb_byte*)(p.B) + (long)n0 * 2;
  auto issueA = [&](int t, int i) {   // A piece i of tile t
    dma16(srcA[i] + (long)t * (FW_BK * p.lda * 2), lds + (t % FW_NBUF) * Cfg::STAGE + (Cfg::A_PIECES * wave + i) * 1024);
  };
  auto issueB = [&](int t) {          // all B pieces of tile t (their row indices come out of the LDS table)
#pragma unroll
    for (int i = 0; i < Cfg::B_PIECES; ++i) {
      if (WG_BN == 96 && i == 1 && wave >= 4) break;   // (wave-uniform)
      const long ridx = rowtab[t * FW_BK + browB[i]];
      const int piece = (WG_BN == 96) ? wave + 8 * i : Cfg::B_PIECES * wave + i;
      dma16(Bbase + ridx * p.ldb * 2 + bcolB[i], lds + (t % FW_NBUF) * Cfg::STAGE + Cfg::A_BYTES + piece * 1024);
    }
  };
  auto issue = [&](int t) {
#pragma unroll
    for (int i = 0; i < Cfg::A_PIECES; ++i) issueA(t, i);
    issueB(t);
  };

  f16v acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, h = lane >> 5;
  const int in_gran = 32 * (gi & 1) + 8 * p4;   // byte offset inside the 64-byte granule (32 columns of one k row)
  const unsigned lds0 = lds_addr_of(fw_smem);
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) {   // all but this wave's youngest tile (t + 1) have landed
      if constexpr (WG_BN == 96) {
        if (wave < 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(Cfg::PIECES) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    const bool more = t + 2 < ntiles;
    if (more) issueB(t + 2);   // (first: its LDS-table reads are compiler-visible and must not sit between the asm reads below and their waits)
    const unsigned stA = lds0 + (t % FW_NBUF) * Cfg::STAGE, stB = stA + Cfg::A_BYTES;
    // fragments of k-step ks + 1 are requested before the MFMAs of k-step ks (two register sets); one A piece of tile
    // t + 2 goes out behind each k-step's MFMAs
    s4v ra[2][Cfg::TM][2], rb[2][Cfg::TN][2];
    auto reads = [&](int ks, int set) {
      const int row = 16 * ks + 8 * h + q4;   // (row & 3) == q4
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        const unsigned ad = stA + row * 512 + (((wm * Cfg::TM + i) ^ q4) * 64) + in_gran;
        tr_issue(ra[set][i][0], ad);
        tr_issue(ra[set][i][1], ad + 4 * 512);
      }
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const int f = (WG_BN == 128) ? q4 : (WG_BN == 96 ? 0 : ((row >> 1) & 1));
        const unsigned ad = stB + row * Cfg::B_ROW + (((wn * Cfg::TN + j) ^ f) * 64) + in_gran;
        tr_issue(rb[set][j][0], ad);
        tr_issue(rb[set][j][1], ad + 4 * Cfg::B_ROW);
      }
    };
    reads(0, 0);
#pragma unroll
    for (int ks = 0; ks < FW_BK / 16; ++ks) {
      const int set = ks & 1;
      if (ks + 1 < FW_BK / 16) {
        reads(ks + 1, set ^ 1);
        // all but the newest 2 TM + 2 TN reads (k-step ks + 1's) are back
        if constexpr (Cfg::TN == 3)
          asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ra[set][0][0]), "+v"(ra[set][0][1]), "+v"(rb[set][0][0]), "+v"(rb[set][0][1]), "+v"(rb[set][1][0]), "+v"(rb[set][1][1]), "+v"(rb[set][2][0]), "+v"(rb[set][2][1]));
        else if constexpr (Cfg::TN == 2)
          asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(ra[set][0][0]), "+v"(ra[set][0][1]), "+v"(ra[set][1][0]), "+v"(ra[set][1][1]), "+v"(rb[set][0][0]), "+v"(rb[set][0][1]), "+v"(rb[set][1][0]), "+v"(rb[set][1][1]));
        else
          asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(ra[set][0][0]), "+v"(ra[set][0][1]), "+v"(ra[set][1][0]), "+v"(ra[set][1][1]), "+v"(rb[set][0][0]), "+v"(rb[set][0][1]));
      } else {
        if constexpr (Cfg::TN == 3)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[set][0][0]), "+v"(ra[set][0][1]), "+v"(rb[set][0][0]), "+v"(rb[set][0][1]), "+v"(rb[set][1][0]), "+v"(rb[set][1][1]), "+v"(rb[set][2][0]), "+v"(rb[set][2][1]));
        else if constexpr (Cfg::TN == 2)
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[set][0][0]), "+v"(ra[set][0][1]), "+v"(ra[set][1][0]), "+v"(ra[set][1][1]), "+v"(rb[set][0][0]), "+v"(rb[set][0][1]), "+v"(rb[set][1][0]), "+v"(rb[set][1][1]));
        else
          asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[set][0][0]), "+v"(ra[set][0][1]), "+v"(ra[set][1][0]), "+v"(ra[set][1][1]), "+v"(rb[set][0][0]), "+v"(rb[set][0][1]));
      }
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
        for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = mfma32h(join8(ra[set][i][0], ra[set][i][1]), join8(rb[set][j][0], rb[set][j][1]), acc[i][j]);
      if (more) issueA(t + 2, ks);   // A_PIECES == 4 == k-steps per tile
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  // ---- dW rows 0 .. 127 -> p.C, rows 128 .. 255 -> p.C2 (the two encoders' weight gradients), fp32 [128][ldc] ----------------------
  const int r = lane & 31;
  const float oscale = *(p.out_scale ? p.out_scale : &g_spv_one);   // 1 / (power-of-two scale of the f16 dh image)
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = n0 + wn * (WG_BN / Cfg::WAVES_N) + 32 * j + r;
      if (col >= p.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = mtile * FW_BM + wm * (FW_BM / Cfg::WAVES_M) + 32 * i + crow(q, h);
        float* dst = (row >= p.c_split_row) ? p.C2 + (long)(row - p.c_split_row) * p.ldc + col : p.C + (long)row * p.ldc + col;
        *dst = acc[i][j][q] * oscale;
      }
    }
}

// ---- the same weight gradient on WIDE gene tiles (192 / 256 genes), K tiles of 32 cells, four stages -------------------------------
// What bounds fc1_wgrad_dma_body is the bytes its workgroups pull through the L2 -> LDS path: every workgroup re-reads the whole dh image
// (2 MB at B 4096) for its 96 genes -- 420 MB per step at C2 beside the 164 MB of the gathered image, at the 6 - 7 TB/s all CUs together
// get through that path.  A 192- / 256-gene tile halves / thirds the dh re-reads per gene (C2: 106 workgroups x 3.6 MB), and with fewer
// workgroups than CUs the limit becomes the single CU's fill rate (~28 B/clk).  A stage is 32 cells deep so that FOUR fit (16 KB of dh +
// 12 / 16 KB of image rows each, three tiles in flight behind counted waits); LDS rows are the global rows (512-byte dh rows, 384- /
// 512-byte gene rows) with the 64-byte granule index XOR-ed by (row & 3) (512-byte rows) / its low bit by (row >> 1) & 1 (384-byte rows: four
// consecutive rows then start 0 / 2 / 1 / 3 granules into the 256-byte bank period).  8 waves as 4 (M) x 2 (N): wave tile 64 x 96 / 128.
constexpr int FWW_BK = 32, FWW_NBUF = 4;
template <int WG_BN>
struct FwwCfg {
  static constexpr int A_BYTES = FWW_BK * FW_BM * 2, B_ROW = WG_BN * 2, B_BYTES = FWW_BK * B_ROW, STAGE = A_BYTES + B_BYTES;
  static constexpr int A_PIECES = A_BYTES / 1024 / 8;                 // 2 per wave
  static constexpr int B_PIECES_ALL = B_BYTES / 1024;                 // 12 (192 genes), 16 (256)
  static constexpr int B_PIECES = (B_PIECES_ALL + 7) / 8;             // per wave; 192 genes: waves 0..3 issue two, waves 4..7 one
  static constexpr int WAVES_N = 2, WAVES_M = 4, TM = FW_BM / WAVES_M / 32, TN = WG_BN / WAVES_N / 32;
  static_assert(WG_BN == 192 || WG_BN == 256, "wide gene tile");
};
__host__ __device__ constexpr int fww_lds_bytes(int wg_bn, int kpad) { return (FWW_BK * FW_BM * 2 + FWW_BK * wg_bn * 2) * FWW_NBUF + kpad * 4; }

template <int WG_BN>
__device__ __forceinline__ void fc1_wgrad_dma_wide_body(const GemmParams& p, const int blk, const int mtile, unsigned char* fw_smem) {
  typedef FwwCfg<WG_BN> Cfg;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_wave_id();
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int n0 = blk * WG_BN;
  const int Kpad = (p.n_cells + FW_BK - 1) / FW_BK * FW_BK, ntiles = Kpad / FWW_BK;   // (the dh image is padded to 64 cells: an even tile count)
  int* rowtab = reinterpret_cast<int*>(fw_smem + Cfg::STAGE * FWW_NBUF);
  for (int k = tid; k < Kpad; k += 512) {
    const int c = k < p.n_cells ? k : p.n_cells - 1;   // padding cells re-read the last row: their dh rows are zero
    rowtab[k] = p.rows ? p.rows[c] : c;
  }
  __syncthreads();

  lds_byte* const lds = (lds_byte*)(fw_smem);
  // A pieces (2 dh rows of 512 B each): wave w owns pieces 2w, 2w + 1 = rows 4w .. 4w + 3 of the K tile
  const glb_byte* srcA[Cfg::A_PIECES];
#pragma unroll
  for (int i = 0; i < Cfg::A_PIECES; ++i) {
    const int piece = Cfg::A_PIECES * wave + i, row = 2 * piece + (lane >> 5), ch = lane & 31;
    const int chs = (((ch >> 2) ^ (row & 3)) << 2) | (ch & 3);
    srcA[i] = (glb_byte*)(p.A) + ((long)row * p.lda + mtile * FW_BM) * 2 + chs * 16;
  }
  int browB[Cfg::B_PIECES], bcolB[Cfg::B_PIECES];
#pragma unroll
  for (int i = 0; i < Cfg::B_PIECES; ++i) {
    int row, cb;
    if constexpr (WG_BN == 256) {   // as the dh rows
      const int piece = Cfg::B_PIECES * wave + i, ch = lane & 31;
      row = 2 * piece + (lane >> 5);
      cb = ((((ch >> 2) ^ (row & 3)) << 2) | (ch & 3)) * 16;
    } else {                        // piece = wave + 8 i (the second one only exists for waves 0..3); pieces cross the 384-byte rows
      const int o = (wave + 8 * i) * 1024 + lane * 16;
      row = (o / Cfg::B_ROW) & (FWW_BK - 1);
      const int ch = (o % Cfg::B_ROW) >> 4;
      cb = ((((ch >> 2) ^ ((row >> 1) & 1)) << 2) | (ch & 3)) * 16;
    }
    if ((long)n0 + cb / 2 + 8 > p.ldb) cb = 0;   // the last workgroup's columns beyond the image row: any in-bounds bytes do (never stored)
    browB[i] = row; bcolB[i] = cb;
  }
  const glb_byte* const Bbase = (glb_byte*)(p.B) + (long)n0 * 2;
  auto issueA = [&](int t, int i) {
    dma16(srcA[i] + (long)t * (FWW_BK * p.lda * 2), lds + (t % FWW_NBUF) * Cfg::STAGE + (Cfg::A_PIECES * wave + i) * 1024);
  };
  auto issueB = [&](int t) {          // all B pieces of tile t (their row indices come out of the LDS table)
#pragma unroll
    for (int i = 0; i < Cfg::B_PIECES; ++i) {
      if (WG_BN == 192 && i == 1 && wave >= 4) break;   // (wave-uniform)
      const long ridx = rowtab[t * FWW_BK + browB[i]];
      const int piece = (WG_BN == 192) ? wave + 8 * i : Cfg::B_PIECES * wave + i;
      dma16(Bbase + ridx * p.ldb * 2 + bcolB[i], lds + (t % FWW_NBUF) * Cfg::STAGE + Cfg::A_BYTES + piece * 1024);
    }
  };
  auto issue = [&](int t) {
    issueB(t);
#pragma unroll
    for (int i = 0; i < Cfg::A_PIECES; ++i) issueA(t, i);
  };

  f16v acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);
  if (ntiles > 2) issue(2);
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, h = lane >> 5;
  const int in_gran = 32 * (gi & 1) + 8 * p4;   // byte offset inside the 64-byte granule (32 columns of one k row)
  const unsigned lds0 = lds_addr_of(fw_smem);
  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed once all but this wave's pieces of the (up to two) younger tiles are done: 4 per tile (192 genes, waves 4..7: 3)
    const int younger = ntiles - 1 - t;
    if (WG_BN == 256 || wave < 4) {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      if (younger >= 2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
      else if (younger == 1) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    raw_barrier();
    const bool more = t + 3 < ntiles;
    if (more) issueB(t + 3);   // (first: its LDS-table reads are compiler-visible and must not sit between the asm reads below and their waits)
    const unsigned stA = lds0 + (t % FWW_NBUF) * Cfg::STAGE, stB = stA + Cfg::A_BYTES;
    s4v ra[2][Cfg::TM][2], rb[2][Cfg::TN][2];
    auto reads_a = [&](int ks, int set) {
      const int row = 16 * ks + 8 * h + q4;   // (row & 3) == q4
#pragma unroll
      for (int i = 0; i < Cfg::TM; ++i) {
        const unsigned ad = stA + row * 512 + (((wm * Cfg::TM + i) ^ q4) * 64) + in_gran;
        tr_issue(ra[set][i][0], ad);
        tr_issue(ra[set][i][1], ad + 4 * 512);
      }
    };
    auto reads_b = [&](int ks, int set) {
      const int row = 16 * ks + 8 * h + q4;
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const int f = (WG_BN == 256) ? q4 : ((q4 >> 1) & 1);
        const unsigned ad = stB + row * Cfg::B_ROW + (((wn * Cfg::TN + j) ^ f) * 64) + in_gran;
        tr_issue(rb[set][j][0], ad);
        tr_issue(rb[set][j][1], ad + 4 * Cfg::B_ROW);
      }
    };
    // (at most 16 LDS reads outstanding: the first k-step's 2 TM + 2 TN and the second one's 2 TM; its 2 TN follow the first wait)
    reads_a(0, 0); reads_b(0, 0); reads_a(1, 1);
    if constexpr (Cfg::TN == 4)
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ra[0][0][0]), "+v"(ra[0][0][1]), "+v"(ra[0][1][0]), "+v"(ra[0][1][1]), "+v"(rb[0][0][0]), "+v"(rb[0][0][1]), "+v"(rb[0][1][0]), "+v"(rb[0][1][1]),
                                            "+v"(rb[0][2][0]), "+v"(rb[0][2][1]), "+v"(rb[0][3][0]), "+v"(rb[0][3][1]));
    else
      asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(ra[0][0][0]), "+v"(ra[0][0][1]), "+v"(ra[0][1][0]), "+v"(ra[0][1][1]), "+v"(rb[0][0][0]), "+v"(rb[0][0][1]), "+v"(rb[0][1][0]), "+v"(rb[0][1][1]),
                                            "+v"(rb[0][2][0]), "+v"(rb[0][2][1]));
    reads_b(1, 1);
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = mfma32h(join8(ra[0][i][0], ra[0][i][1]), join8(rb[0][j][0], rb[0][j][1]), acc[i][j]);
    if (more) issueA(t + 3, 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (Cfg::TN == 4)
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[1][0][0]), "+v"(ra[1][0][1]), "+v"(ra[1][1][0]), "+v"(ra[1][1][1]), "+v"(rb[1][0][0]), "+v"(rb[1][0][1]), "+v"(rb[1][1][0]), "+v"(rb[1][1][1]),
                                            "+v"(rb[1][2][0]), "+v"(rb[1][2][1]), "+v"(rb[1][3][0]), "+v"(rb[1][3][1]));
    else
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[1][0][0]), "+v"(ra[1][0][1]), "+v"(ra[1][1][0]), "+v"(ra[1][1][1]), "+v"(rb[1][0][0]), "+v"(rb[1][0][1]), "+v"(rb[1][1][0]), "+v"(rb[1][1][1]),
                                            "+v"(rb[1][2][0]), "+v"(rb[1][2][1]));
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) acc[i][j] = mfma32h(join8(ra[1][i][0], ra[1][i][1]), join8(rb[1][j][0], rb[1][j][1]), acc[i][j]);
    if (more) issueA(t + 3, 1);
    __builtin_amdgcn_sched_barrier(0);
  }
  // ---- dW rows 0 .. 127 -> p.C, rows 128 .. 255 -> p.C2 (the two encoders' weight gradients), fp32 [128][ldc] ----------------------
  const int r = lane & 31;
  const float oscale = *(p.out_scale ? p.out_scale : &g_spv_one);   // 1 / (power-of-two scale of the f16 dh image)
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = n0 + wn * (WG_BN / Cfg::WAVES_N) + 32 * j + r;
      if (col >= p.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = mtile * FW_BM + wm * (FW_BM / Cfg::WAVES_M) + 32 * i + crow(q, h);
        float* dst = (row >= p.c_split_row) ? p.C2 + (long)(row - p.c_split_row) * p.ldc + col : p.C + (long)row * p.ldc + col;
        *dst = acc[i][j][q] * oscale;
      }
    }
}
// both groups in one grid (workgroups 0 .. n0 - 1: the first group's gene tiles); dynamic LDS = fww_lds_bytes of the larger group
template <int WG_BN>
__global__ __launch_bounds__(512) void fc1_wgrad_dma_wide_pair_kernel(GemmParams p0, GemmParams p1, int n0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fw_smem_[];
  const bool first = (int)blockIdx.x < n0;
  const GemmParams p = first ? p0 : p1;
  fc1_wgrad_dma_wide_body<WG_BN>(p, first ? (int)blockIdx.x : (int)blockIdx.x - n0, blockIdx.y, fw_smem_);
}

// ---- the weight gradient on split-bf16 operands ("fp32" mode), 96- or 128-gene tiles -----------------------------------------------
// p.A / p.A_lo = the hi / lo dh images [Kpad][lda], p.B = the resident image of spv_prepare_log1p_split (hi / lo interleaved per 32-gene
// block).  A stage is 32 cells deep: LDS rows 0..31 hold the hi rows of its cells, rows 32..63 the lo rows, for both operands -- the
// stage image, the piece counts and the counted waits are those of fc1_wgrad_dma_body<WG_BN>; each of the two k-steps reads hi and lo
// fragments (rows r and r + 32) and issues three MFMAs per fragment pair.  Results straight into dW / dW2.  (128-gene tiles exist for
// the round structure: at G = 30 000 they make 235 workgroups = ONE round of the 256 CUs per group where 96-gene tiles make 313 = two,
// the second a quarter full: 373 -> 2xx us per group.)
template <int WG_BN>
__device__ __forceinline__ void fc1_wgrad_dma_split_body(const GemmParams& p, const int blk, const int mtile, unsigned char* fw_smem) {
  typedef FwCfg<WG_BN> Cfg;
  static_assert(WG_BN == 96 || WG_BN == 128, "gene tile");
  constexpr int SK = FW_BK / 2;   // cells per stage
  constexpr int TM = Cfg::TM, TN = Cfg::TN;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_wave_id();
  const int wm = wave / Cfg::WAVES_N, wn = wave % Cfg::WAVES_N;
  const int n0 = blk * WG_BN;
  const int Kpad = (p.n_cells + FW_BK - 1) / FW_BK * FW_BK, ntiles = Kpad / SK;
  int* rowtab = reinterpret_cast<int*>(fw_smem + Cfg::STAGE * FW_NBUF);
  for (int k = tid; k < Kpad; k += 512) {
    const int c = k < p.n_cells ? k : p.n_cells - 1;   // padding cells re-read the last row: their dh rows are zero
    rowtab[k] = p.rows ? p.rows[c] : c;
  }
  __syncthreads();

  lds_byte* const lds = (lds_byte*)(fw_smem);
  // A pieces: LDS rows 2 piece + (lane >> 5), piece = 4 wave + i: plane = row >> 5, cell of the stage = row & 31
  const glb_byte* srcA[Cfg::A_PIECES];
#pragma unroll
  for (int i = 0; i < Cfg::A_PIECES; ++i) {
    const int piece = Cfg::A_PIECES * wave + i, row = 2 * piece + (lane >> 5), ch = lane & 31;
    const int chs = (((ch >> 2) ^ (row & 3)) << 2) | (ch & 3);
    srcA[i] = (glb_byte*)((row >> 5) ? p.A_lo : p.A) + ((long)(row & 31) * p.lda + mtile * FW_BM) * 2 + chs * 16;
  }
  // B pieces.  96 genes: piece = wave + 8 i (the second only for waves 0..3), pieces cross the 192-byte LDS rows, no swizzle;
  // 128 genes: piece = 2 wave + i = LDS rows 4 piece + (lane >> 4), 64-byte granules XOR-ed with row & 3 (fc1_wgrad_dma_body)
  int bcell[Cfg::B_PIECES], boff[Cfg::B_PIECES], bdst[Cfg::B_PIECES];
#pragma unroll
  for (int i = 0; i < Cfg::B_PIECES; ++i) {
    int row, gl;   // LDS row, first of the chunk's 8 genes inside the tile
    if constexpr (WG_BN == 96) {
      const int o = (wave + 8 * i) * 1024 + lane * 16;
      row = (o / Cfg::B_ROW) & (FW_BK - 1); gl = (o % Cfg::B_ROW) / 2;
      bdst[i] = (wave + 8 * i) * 1024;
    } else {
      const int piece = Cfg::B_PIECES * wave + i, ch = lane & 15;
      row = 4 * piece + (lane >> 4);
      gl = ((((ch >> 2) ^ (row & 3)) << 2) | (ch & 3)) * 8;
      bdst[i] = piece * 1024;
    }
    bcell[i] = row & 31;
    int e = ((n0 + gl) >> 5) * 64 + (gl & 31) + 32 * (row >> 5);   // word offset inside the interleaved image row (n0 is a multiple of 32)
    if ((long)e + 8 > p.ldb) e = 0;   // the last workgroup's columns beyond the image row: any in-bounds bytes do (never stored)
    boff[i] = e * 2;
  }
  const glb_byte* const Bbase = (glb_byte*)(p.B);
  auto issueA = [&](int t, int i) {
    dma16(srcA[i] + (long)t * (SK * p.lda * 2), lds + (t % FW_NBUF) * Cfg::STAGE + (Cfg::A_PIECES * wave + i) * 1024);
  };
  auto issueB = [&](int t) {
#pragma unroll
    for (int i = 0; i < Cfg::B_PIECES; ++i) {
      if (WG_BN == 96 && i == 1 && wave >= 4) break;   // (wave-uniform)
      const long ridx = rowtab[t * SK + bcell[i]];
      dma16(Bbase + ridx * p.ldb * 2 + boff[i], lds + (t % FW_NBUF) * Cfg::STAGE + Cfg::A_BYTES + bdst[i]);
    }
  };
  auto issue = [&](int t) {
#pragma unroll
    for (int i = 0; i < Cfg::A_PIECES; ++i) issueA(t, i);
    issueB(t);
  };

  f16v acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  if (ntiles > 0) issue(0);
  if (ntiles > 1) issue(1);
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, h = lane >> 5;
  const int in_gran = 32 * (gi & 1) + 8 * p4;
  const unsigned lds0 = lds_addr_of(fw_smem);
  for (int t = 0; t < ntiles; ++t) {
    if (t + 1 < ntiles) {
      if constexpr (WG_BN == 96) {
        if (wave < 4) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    const bool more = t + 2 < ntiles;
    if (more) issueB(t + 2);   // (first: its LDS-table reads are compiler-visible and must not sit between the asm reads below and their waits)
    const unsigned stA = lds0 + (t % FW_NBUF) * Cfg::STAGE, stB = stA + Cfg::A_BYTES;
    // 16 transposed reads per k-step and set: TM = 1, TN = 3 (96 genes) or TM = TN = 2 (128), hi and lo, two halves each
    s4v rr[2][16];
    auto reads = [&](int ks, int set) {
      int n = 0;
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) {
        const int row = 32 * pl + 16 * ks + 8 * h + q4;   // (row & 3) == q4
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const unsigned ad = stA + row * 512 + (((wm * TM + i) ^ q4) * 64) + in_gran;
          tr_issue(rr[set][n++], ad);
          tr_issue(rr[set][n++], ad + 4 * 512);
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const int f = (WG_BN == 128) ? q4 : 0;
          const unsigned bd = stB + row * Cfg::B_ROW + (((wn * TN + j) ^ f) * 64) + in_gran;
          tr_issue(rr[set][n++], bd);
          tr_issue(rr[set][n++], bd + 4 * Cfg::B_ROW);
        }
      }
    };
    reads(0, 0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int set = ks;
      if (ks == 0) {
        reads(1, 1);   // (16 more reads; the counter saturates at 15: "all but the newest 15" covers the first k-step's 16)
        asm volatile("s_waitcnt lgkmcnt(15)" : "+v"(rr[0][0]), "+v"(rr[0][1]), "+v"(rr[0][2]), "+v"(rr[0][3]), "+v"(rr[0][4]), "+v"(rr[0][5]), "+v"(rr[0][6]), "+v"(rr[0][7]),
                     "+v"(rr[0][8]), "+v"(rr[0][9]), "+v"(rr[0][10]), "+v"(rr[0][11]), "+v"(rr[0][12]), "+v"(rr[0][13]), "+v"(rr[0][14]), "+v"(rr[0][15]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(rr[1][0]), "+v"(rr[1][1]), "+v"(rr[1][2]), "+v"(rr[1][3]), "+v"(rr[1][4]), "+v"(rr[1][5]), "+v"(rr[1][6]), "+v"(rr[1][7]),
                     "+v"(rr[1][8]), "+v"(rr[1][9]), "+v"(rr[1][10]), "+v"(rr[1][11]), "+v"(rr[1][12]), "+v"(rr[1][13]), "+v"(rr[1][14]), "+v"(rr[1][15]));
      }
      // plane pl's reads sit at rr[set][8 pl ..]: first the TM A fragments (two halves each), then the TN B fragments
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const s8v a_hi = join8(rr[set][2 * i], rr[set][2 * i + 1]), a_lo = join8(rr[set][8 + 2 * i], rr[set][8 + 2 * i + 1]);
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const s8v b_hi = join8(rr[set][2 * TM + 2 * j], rr[set][2 * TM + 2 * j + 1]), b_lo = join8(rr[set][8 + 2 * TM + 2 * j], rr[set][8 + 2 * TM + 2 * j + 1]);
          acc[i][j] = mfma32(a_hi, b_lo, acc[i][j]);
          acc[i][j] = mfma32(a_lo, b_hi, acc[i][j]);
          acc[i][j] = mfma32(a_hi, b_hi, acc[i][j]);
        }
      }
      if (more) { issueA(t + 2, 2 * ks); issueA(t + 2, 2 * ks + 1); }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const int r = lane & 31;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * (WG_BN / Cfg::WAVES_N) + 32 * j + r;
      if (col >= p.N) continue;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = mtile * FW_BM + wm * (FW_BM / Cfg::WAVES_M) + 32 * i + crow(q, h);
        float* dst = (row >= p.c_split_row) ? p.C2 + (long)(row - p.c_split_row) * p.ldc + col : p.C + (long)row * p.ldc + col;
        *dst = acc[i][j][q];
      }
    }
}
template <int WG_BN>
__global__ __launch_bounds__(512) void fc1_wgrad_dma_split_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fw_smem_[];
  fc1_wgrad_dma_split_body<WG_BN>(p, blockIdx.x, blockIdx.y, fw_smem_);
}

template <int WG_BN>
__global__ __launch_bounds__(512) void fc1_wgrad_dma_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fw_smem_[];
  fc1_wgrad_dma_body<WG_BN>(p, blockIdx.x, blockIdx.y, fw_smem_);
}
// both groups in one grid (see fc1_fwd_dma_pair_kernel); dynamic LDS = the larger of the two groups' needs
template <int WG_BN>
__global__ __launch_bounds__(512) void fc1_wgrad_dma_pair_kernel(GemmParams p0, GemmParams p1, int n0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char fw_smem_[];
  const bool first = (int)blockIdx.x < n0;
  const GemmParams p = first ? p0 : p1;
  fc1_wgrad_dma_body<WG_BN>(p, first ? (int)blockIdx.x : (int)blockIdx.x - n0, blockIdx.y, fw_smem_);
}

}  // namespace spv
