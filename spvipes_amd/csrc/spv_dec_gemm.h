// The two 320-column GEMMs of the decoder backward in bf16 mode, LDS-DMA staged (same skeleton as spv_fc1.h):
//
//   d A_m[cell][n] = sum_gene dL[cell][gene] * W_m[gene][n]      (backward of the mixing head, nn/networks.py:322-325, wrt its input)
//   d W_m[gene][n] = sum_cell dL[cell][gene] * A_m[cell][n]      (... wrt its weight; n runs over the 320-column padded image
//                                                                 [hidden trunk 256 | z_private | z_shared | 1 | 0...])
//
// dL is the likelihood kernel's gradient wrt the mixing logits, bf16 in accumulator-tile order
// T[cell / 32][gene / 32][qq][lane][j] (gene % 32 = 8 qq + 4 h + j, lane = cell % 32 + 32 h; spv_decoder.h): a 32 x 32 tile is 2 KiB
// of contiguous memory, so the LDS-DMA moves whole tiles and the fragment reads pick 8-byte pieces ((cell, 4 genes)) out of them.
// The other operand (W_m image [Gp][320] or A_m image [Bp][320], bf16 row-major) has the contraction index on its rows in both
// GEMMs: "k-major", fragments by transposed LDS reads (ds_read_b64_tr_b16).
//
// Workgroup = 8 waves as 4 (M) x 2 (N): 128 rows x 320 columns, wave tile 32 x 160 = 1 x 5 MFMA 32x32x16 tiles (6 fragments
// for 5 MFMAs per k step -- the register-staged 64 x 320 kernel of spv_gemm.h reads 2 + 5 for 2 x 2.5).  K tile = 64, two 56 KiB
// LDS stages (16 KiB of dL tiles + 40 KiB of the k-major operand), tile t + 1 in flight while tile t is multiplied, one
// s_waitcnt vmcnt(0) + one barrier per tile.  Split-K over workgroups, fp32 slabs [split][M][320] for spv_reduce_slabs.
//
// LDS images and bank conflicts:
//   * k-major operand: lane-linear copy of 64 rows x 640 bytes.  The four k rows of a transposed read are 640 bytes apart =
//     128 bytes modulo the 256-byte bank period, rows r and r + 2 would collide: the 64-byte granule (32 columns) index inside a
//     row is XOR-ed with (row >> 1) & 1 on the DMA's source address and again on the read;
//   * dL, d A_m (cells on M): tiles copied as they are; a fragment is two 8-byte reads 256 bytes apart (h = 0, 1) per lane;
//   * dL, d W_m (cells on K, transposed reads): inside a tile the 16-byte DMA chunks ((2 cells) x (4 genes)) are re-ordered to
//     [cell / 4][gene / 4][cell % 4][4 genes], so that the 4 cells x 4 gene-pieces of a 16-lane transposed read cover 128
//     contiguous bytes (in tile order the gene pieces are 256 / 512 bytes apart: the same banks, a 4-way conflict).
#pragma once
#include "spv_common.h"
#include "spv_gemm.h"
#include "spv_fc1.h"
#include "spv_decoder.h"

namespace spv {

constexpr int DG_BM = 128, DG_BN = 320, DG_BK = 64;
constexpr int DG_A_BYTES = DG_BM * DG_BK * 2, DG_B_ROW = DG_BN * 2, DG_B_BYTES = DG_BK * DG_B_ROW, DG_STAGE = DG_A_BYTES + DG_B_BYTES;
constexpr int DG_LDS_BYTES = 2 * DG_STAGE;                       // 114 688 B: one workgroup per CU
constexpr int DG_A_PIECES = DG_A_BYTES / 1024 / 8, DG_B_PIECES = DG_B_BYTES / 1024 / 8;   // 2 + 5 LDS-DMA pieces (1 KiB) per wave and tile

__device__ __forceinline__ void lds_read8(s4v& d, unsigned addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr)); }

// A_CELLS_ON_K = false: d A_m (M = cells, K = genes);  true: d W_m (M = genes, K = cells).
// p.A = dL tiles (bf16), p.tiles_inner = gene tiles per cell tile; p.B = k-major image [K rows][320] bf16 (ldb == 320);
// p.C = slabs [split][M][ldc] fp32, p.slab_stride; p.M, p.N (<= 320 columns stored), p.K; p.k_per_split multiple of 64;
// p.c_split_row = number of K splits.  grid = ceil(M / 128) * splits (1-D, split fastest: one K range per XCD), 512 threads.
template <bool A_CELLS_ON_K>
__global__ __launch_bounds__(512) void dec_gemm320_dma_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dg_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int split = blockIdx.x % p.c_split_row, mtile = blockIdx.x / p.c_split_row;
  const int m0 = mtile * DG_BM;
  const int kbeg = split * p.k_per_split;
  const int Kpad = (p.K + DG_BK - 1) / DG_BK * DG_BK;
  int kend = kbeg + p.k_per_split;
  if (kend > Kpad) kend = Kpad;
  const int ntiles = (kend - kbeg) / DG_BK;   // may be <= 0 for a trailing split: its slab is zeros

  lds_byte* const lds = (lds_byte*)(dg_smem);
  // ---- DMA sources ---------------------------------------------------------------------------------------------------------
  // A: 8 tiles of 2 KiB per stage = 16 pieces; wave w owns pieces 2w, 2w + 1 = the two halves of stage tile w.
  //    d A_m: stage tile w = (cell tile w >> 1, gene tile w & 1);  d W_m: stage tile w = (cell tile w >> 2, gene tile w & 3)
  const glb_byte* srcA[DG_A_PIECES];
  long a_step;   // byte advance of the A sources per K tile
  {
    const long T = p.tiles_inner;
    long tile0;
    if constexpr (!A_CELLS_ON_K) { tile0 = ((long)(m0 / 32) + (wave >> 1)) * T + (kbeg / 32) + (wave & 1); a_step = 2 * 2048; }
    else { tile0 = ((long)(kbeg / 32) + (wave >> 2)) * T + (m0 / 32) + (wave & 3); a_step = 2 * T * 2048; }
#pragma unroll
    for (int i = 0; i < DG_A_PIECES; ++i) {
      int chunk = i * 64 + lane;   // 16-byte chunk of the LDS tile image this lane fills
      if constexpr (A_CELLS_ON_K) {  // LDS chunk (cq, gp, cpair) <- tile-order chunk qq * 32 + hh * 16 + c / 2, gp = 2 qq + hh, c = 4 cq + 2 cpair
        const int cq = chunk >> 4, gp = (chunk >> 1) & 7, cpair = chunk & 1;
        chunk = (gp >> 1) * 32 + (gp & 1) * 16 + 2 * cq + cpair;
      }
      srcA[i] = (glb_byte*)(p.A) + tile0 * 2048 + chunk * 16;
    }
  }
  // B: 64 rows x 640 bytes = 40 pieces; wave w owns pieces 5w .. 5w + 4
  const glb_byte* srcB[DG_B_PIECES];
#pragma unroll
  for (int i = 0; i < DG_B_PIECES; ++i) {
    const int o = (DG_B_PIECES * wave + i) * 1024 + lane * 16;     // byte offset inside the stage's B image
    const int row = o / DG_B_ROW, w = o % DG_B_ROW;
    const int gran = (w >> 6) ^ ((row >> 1) & 1);
    srcB[i] = (glb_byte*)(p.B) + ((long)(kbeg + row) * DG_B_ROW) + gran * 64 + (w & 63);
  }
  auto issue = [&](int t) {
    const int stage = (t & 1) * DG_STAGE;
#pragma unroll
    for (int i = 0; i < DG_A_PIECES; ++i) dma16(srcA[i] + (long)t * a_step, lds + stage + (DG_A_PIECES * wave + i) * 1024);
#pragma unroll
    for (int i = 0; i < DG_B_PIECES; ++i) dma16(srcB[i] + (long)t * DG_B_BYTES, lds + stage + DG_A_BYTES + (DG_B_PIECES * wave + i) * 1024);
  };

  f16v acc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  if (ntiles > 0) issue(0);
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, h = lane >> 5, r = lane & 31;
  const int in_gran = 32 * (gi & 1) + 8 * p4;   // byte offset inside the 64-byte granule (32 columns of one k row)
  const int fB = (q4 >> 1) & 1;                 // (row >> 1) & 1 of every k row this lane reads (rows 16 ks + 8 h + q4 [+ 4])
  const unsigned lds0 = lds_addr_of(dg_smem);
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's pieces of tile t have landed ...
    raw_barrier();                                     // ... everybody's have, and everybody is done reading the other buffer
    if (t + 1 < ntiles) issue(t + 1);
    const unsigned stA = lds0 + (t & 1) * DG_STAGE, stB = stA + DG_A_BYTES;
    s4v ra[2][2], rb[2][5][2];
    auto reads = [&](int ks, int set) {
      if constexpr (!A_CELLS_ON_K) {   // stage tile (wm, ks >> 1): genes 16 (ks & 1) + 8 h .. + 7 of cell r = two 8-byte pieces
        const unsigned ad = stA + (wm * 2 + (ks >> 1)) * 2048 + (2 * (ks & 1) + h) * 512 + r * 8;
        lds_read8(ra[set][0], ad);
        lds_read8(ra[set][1], ad + 256);
      } else {                         // stage tile (ks >> 1, wm): cells 16 (ks & 1) + 8 h + q4 (+ 4), gene piece 4 (gi & 1) + p4
        const unsigned ad = stA + ((ks >> 1) * 4 + wm) * 2048 + (4 * (ks & 1) + 2 * h) * 256 + (4 * (gi & 1) + p4) * 32 + q4 * 8;
        tr_issue(ra[set][0], ad);
        tr_issue(ra[set][1], ad + 256);
      }
      const unsigned rowB = stB + (16 * ks + 8 * h + q4) * DG_B_ROW + in_gran;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const unsigned ad = rowB + (((wn * 5 + j) ^ fB) * 64);
        tr_issue(rb[set][j][0], ad);
        tr_issue(rb[set][j][1], ad + 4 * DG_B_ROW);
      }
    };
    reads(0, 0);
#pragma unroll
    for (int ks = 0; ks < DG_BK / 16; ++ks) {
      const int set = ks & 1;
      if (ks + 1 < DG_BK / 16) {
        reads(ks + 1, set ^ 1);
        // all but the newest 12 reads (k-step ks + 1's) are back
        asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(ra[set][0]), "+v"(ra[set][1]), "+v"(rb[set][0][0]), "+v"(rb[set][0][1]), "+v"(rb[set][1][0]), "+v"(rb[set][1][1]),
                     "+v"(rb[set][2][0]), "+v"(rb[set][2][1]), "+v"(rb[set][3][0]), "+v"(rb[set][3][1]), "+v"(rb[set][4][0]), "+v"(rb[set][4][1]));
      } else {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[set][0]), "+v"(ra[set][1]), "+v"(rb[set][0][0]), "+v"(rb[set][0][1]), "+v"(rb[set][1][0]), "+v"(rb[set][1][1]),
                     "+v"(rb[set][2][0]), "+v"(rb[set][2][1]), "+v"(rb[set][3][0]), "+v"(rb[set][3][1]), "+v"(rb[set][4][0]), "+v"(rb[set][4][1]));
      }
      const s8v a = join8(ra[set][0], ra[set][1]);
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[j] = mfma32(a, join8(rb[set][j][0], rb[set][j][1]), acc[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // ---- fp32 partial slab [M][ldc] of this split ---------------------------------------------------------------------------------
  float* slab = p.C + (long)split * p.slab_stride;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int col = wn * 160 + 32 * j + r;
    if (col >= p.N) continue;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = m0 + wm * 32 + crow(q, h);
      if (row < p.M) slab[(long)row * p.ldc + col] = acc[j][q];
    }
  }
}

// ---- split-K fix-up inside the launch (spv_gemm_fixup) ------------------------------------------------------------------------------
// After its slab tile is stored, a workgroup publishes it and takes a ticket; the last of the row tile's `c_split_row` workgroups sums the
// slabs.  The hand-off is the release / acquire form of cdna_hip_programming.md ("In-launch split-K reduction"): every storing wave drains its
// stores, workgroup barrier, ONE lane releases at agent scope and waits for the release before the relaxed agent-scope ticket add; the last
// arriver's lane 0 acquires at agent scope, waits, workgroup barrier, then every wave reads the slabs with plain 16-byte loads.  Correct for
// any placement of the tile's workgroups on CUs / XCDs.  The sum runs in slab order 0, 1, ..: the bits spv_reduce_slabs gives.
__device__ __forceinline__ void splitk_fixup_tile(const GemmParams& p, int mtile, int m0, volatile unsigned* s_ticket) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's slab stores have left
  __syncthreads();
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // (the compiler may drop the fence's own wait: keep this one)
    *s_ticket = __hip_atomic_fetch_add(p.fix_cnt + mtile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __syncthreads();
  if (*s_ticket != (unsigned)(p.c_split_row - 1)) return;   // (workgroup-uniform)
  if (threadIdx.x == 0) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    p.fix_cnt[mtile] = 0u;   // the next launch finds its counter at zero (a kernel boundary lies between)
  }
  __syncthreads();
  const int S = p.c_split_row;
  const int ncol = max(p.fix_n0, p.fix_d1 ? p.fix_c1 + p.fix_n1 : 0);
  const int nc4 = (ncol + 3) >> 2;                       // (ldc is a multiple of 4 and the rows are 16-byte aligned: checked by the host)
  const int rows = min(DG_BM, p.M - m0);
  const float alpha = p.fix_alpha ? *p.fix_alpha : 1.f;
  const int nthreads = (int)blockDim.x;
#pragma unroll 2
  for (int e = threadIdx.x; e < rows * nc4; e += nthreads) {
    const int row = e / nc4, c4 = e - row * nc4;
    const float* src = p.C + (long)(m0 + row) * p.ldc + 4 * c4;
    f4v acc = {0.f, 0.f, 0.f, 0.f};
    for (int s0 = 0; s0 < S; s0 += 4) {   // four slabs requested at once (clamped index + select: no load behind a run-time test)
      f4v v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const f4v*>(src + (long)min(s0 + u, S - 1) * p.slab_stride);
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (s0 + u < S) { acc[0] += v[u][0]; acc[1] += v[u][1]; acc[2] += v[u][2]; acc[3] += v[u][3]; }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int col = 4 * c4 + j;
      const float val = acc[j] * alpha;
      if (col < p.fix_n0) p.fix_d0[(long)(m0 + row) * p.fix_ld0 + col] = val;
      else if (p.fix_d1 && col >= p.fix_c1 && col < p.fix_c1 + p.fix_n1) p.fix_d1[(long)(m0 + row) * p.fix_ld1 + col - p.fix_c1] = val;
    }
  }
}

// ---- the same two GEMMs with K tiles of 32 and FOUR LDS stages (three tiles = 84 KiB in flight per CU instead of one of 56 KiB) ------
// The two-stage kernel above waits for a whole tile's fill between two multiplies: it runs at the LATENCY of one 56 KiB fill per tile
// (~1.4 us) rather than at the CU's fill rate.  Here a tile is 8 KiB of dL (4 tiles' halves: one 32 x 32 tile per wave pair) + 20 KiB
// of the k-major operand (32 rows x 640 B = 20 pieces: waves 0..3 issue three, waves 4..7 two), and the wait at the top of tile t is
// the counted one: everything but this wave's pieces of the three younger tiles (12 / 9) must have landed.
constexpr int D4_BK = 32, D4_NBUF = 4;
constexpr int D4_A_BYTES = DG_BM * D4_BK * 2, D4_B_BYTES = D4_BK * DG_B_ROW, D4_STAGE = D4_A_BYTES + D4_B_BYTES;   // 8 K + 20 K
constexpr int D4_LDS_BYTES = D4_NBUF * D4_STAGE;   // 114 688 B

template <bool A_CELLS_ON_K>
__device__ __forceinline__ void dec_gemm320_dma4_body(const GemmParams& p, unsigned char* d4_smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int split = blockIdx.x % p.c_split_row, mtile = blockIdx.x / p.c_split_row;
  const int m0 = mtile * DG_BM;
  const int kbeg = split * p.k_per_split;
  const int Kpad = (p.K + DG_BK - 1) / DG_BK * DG_BK;   // (the operands are padded to multiples of 64 along K)
  int kend = kbeg + p.k_per_split;
  if (kend > Kpad) kend = Kpad;
  const int ntiles = (kend - kbeg) / D4_BK;

  lds_byte* const lds = (lds_byte*)(d4_smem);
  // A: 4 tiles of 2 KiB per stage = 8 pieces, one per wave: the half (wave & 1) of stage tile (wave >> 1).
  //    d A_m: stage tile = cell tile (one gene tile per K tile);  d W_m: stage tile = gene tile (one cell tile per K tile)
  const glb_byte* srcA;
  long a_step;
  {
    const long T = p.tiles_inner;
    long tile0;
    if constexpr (!A_CELLS_ON_K) { tile0 = ((long)(m0 / 32) + (wave >> 1)) * T + (kbeg / 32); a_step = 2048; }
    else { tile0 = (long)(kbeg / 32) * T + (m0 / 32) + (wave >> 1); a_step = T * 2048; }
    int chunk = (wave & 1) * 64 + lane;
    if constexpr (A_CELLS_ON_K) {
      const int cq = chunk >> 4, gp = (chunk >> 1) & 7, cpair = chunk & 1;
      chunk = (gp >> 1) * 32 + (gp & 1) * 16 + 2 * cq + cpair;
    }
    srcA = (glb_byte*)(p.A) + tile0 * 2048 + chunk * 16;
  }
  // B: 20 pieces: piece = wave + 8 i, i = 0..2 (the third only for waves 0..3)
  const glb_byte* srcB[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int o = ((wave + 8 * i) * 1024 + lane * 16) % D4_B_BYTES;   // (% only tames the non-existent third piece of waves 4..7)
    const int row = o / DG_B_ROW, w = o % DG_B_ROW;
    const int gran = (w >> 6) ^ ((row >> 1) & 1);
    srcB[i] = (glb_byte*)(p.B) + ((long)(kbeg + row) * DG_B_ROW) + gran * 64 + (w & 63);
  }
  auto issue = [&](int t) {
    const int stage = (t % D4_NBUF) * D4_STAGE;
    dma16(srcA + (long)t * a_step, lds + stage + wave * 1024);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i == 2 && wave >= 4) break;   // (wave-uniform)
      dma16(srcB[i] + (long)t * D4_B_BYTES, lds + stage + D4_A_BYTES + (wave + 8 * i) * 1024);
    }
  };

  f16v acc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  for (int t = 0; t < D4_NBUF - 1 && t < ntiles; ++t) issue(t);
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, h = lane >> 5, r = lane & 31;
  const int in_gran = 32 * (gi & 1) + 8 * p4;
  const int fB = (q4 >> 1) & 1;
  const unsigned lds0 = lds_addr_of(d4_smem);
  for (int t = 0; t < ntiles; ++t) {
    // tile t has landed once all but this wave's pieces of the younger tiles in flight are done
    const int younger = min(ntiles - 1 - t, D4_NBUF - 2);
    if (younger == 2) { if (wave < 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else if (younger == 1) { if (wave < 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();   // ... everybody's pieces have, and everybody is done reading buffer (t - 1) % 4, which tile t + 3 now overwrites
    if (t + D4_NBUF - 1 < ntiles) issue(t + D4_NBUF - 1);
    const unsigned stA = lds0 + (t % D4_NBUF) * D4_STAGE, stB = stA + D4_A_BYTES;
    s4v ra[2][2], rb[2][5][2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if constexpr (!A_CELLS_ON_K) {
        const unsigned ad = stA + wm * 2048 + (2 * ks + h) * 512 + r * 8;
        lds_read8(ra[ks][0], ad);
        lds_read8(ra[ks][1], ad + 256);
      } else {
        const unsigned ad = stA + wm * 2048 + (4 * ks + 2 * h) * 256 + (4 * (gi & 1) + p4) * 32 + q4 * 8;
        tr_issue(ra[ks][0], ad);
        tr_issue(ra[ks][1], ad + 256);
      }
      const unsigned rowB = stB + (16 * ks + 8 * h + q4) * DG_B_ROW + in_gran;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const unsigned ad = rowB + (((wn * 5 + j) ^ fB) * 64);
        tr_issue(rb[ks][j][0], ad);
        tr_issue(rb[ks][j][1], ad + 4 * DG_B_ROW);
      }
    }
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks == 0)
        asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(rb[0][0][0]), "+v"(rb[0][0][1]), "+v"(rb[0][1][0]), "+v"(rb[0][1][1]),
                     "+v"(rb[0][2][0]), "+v"(rb[0][2][1]), "+v"(rb[0][3][0]), "+v"(rb[0][3][1]), "+v"(rb[0][4][0]), "+v"(rb[0][4][1]));
      else
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[1][0]), "+v"(ra[1][1]), "+v"(rb[1][0][0]), "+v"(rb[1][0][1]), "+v"(rb[1][1][0]), "+v"(rb[1][1][1]),
                     "+v"(rb[1][2][0]), "+v"(rb[1][2][1]), "+v"(rb[1][3][0]), "+v"(rb[1][3][1]), "+v"(rb[1][4][0]), "+v"(rb[1][4][1]));
      const s8v a = join8(ra[ks][0], ra[ks][1]);
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[j] = mfma32(a, join8(rb[ks][j][0], rb[ks][j][1]), acc[j]);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  float* slab = p.C + (long)split * p.slab_stride;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int col = wn * 160 + 32 * j + r;
    if (col >= p.N) continue;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = m0 + wm * 32 + crow(q, h);
      if (row < p.M) slab[(long)row * p.ldc + col] = acc[j][q];
    }
  }
  if (p.fix_cnt != nullptr) splitk_fixup_tile(p, mtile, m0, reinterpret_cast<volatile unsigned*>(d4_smem));   // (kernel-uniform; the stages are dead)
}
template <bool A_CELLS_ON_K>
__global__ __launch_bounds__(512) void dec_gemm320_dma4_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char d4_smem_[];
  dec_gemm320_dma4_body<A_CELLS_ON_K>(p, d4_smem_);
}
// both groups' GEMMs in one grid: group = blockIdx.y (the grid is 1-D per group: row tile x split)
template <bool A_CELLS_ON_K>
__global__ __launch_bounds__(512) void dec_gemm320_dma4_pair_kernel(GemmParams p0, GemmParams p1) {
  extern __shared__ __attribute__((aligned(16))) unsigned char d4_smem_[];
  const GemmParams p = blockIdx.y ? p1 : p0;
  if ((int)blockIdx.x >= (p.M + DG_BM - 1) / DG_BM * p.c_split_row) return;   // (workgroup-uniform)
  dec_gemm320_dma4_body<A_CELLS_ON_K>(p, d4_smem_);
}

// ---- the same two GEMMs on split-bf16 operands ("fp32" mode: x = hi + lo, x*y ~ hi*hi + hi*lo + lo*hi, three MFMAs) -----------------------
// Both operands come as a hi and a lo plane (dL: two tile-ordered arrays; the k-major image: two [K][320] arrays).  A stage is 16 deep:
// 4 KiB of dL per plane (the K half of four 32 x 32 tiles) + 10 KiB of the k-major operand per plane = 28 KiB, FOUR stages as in the
// bf16 kernel above -- the same 8 + 20 DMA pieces per stage (wave w: one dL piece, plane w & 1; the k-major pieces: hi plane 0..9,
// lo plane 10..19), the same counted waits -- and one k-step of 3 x 5 MFMAs per wave and stage.  The register-staged 64 x 320 kernel of
// spv_gemm.h that served this mode before spends 717 / 638 us per group at C5's shard shape; see DESIGN.md section 4 for this one.
constexpr int D4S_BK = 16;
constexpr int D4S_A_BYTES = DG_BM * D4S_BK * 2, D4S_B_BYTES = D4S_BK * DG_B_ROW, D4S_STAGE = 2 * D4S_A_BYTES + 2 * D4S_B_BYTES;   // 2 x 4 K + 2 x 10 K
constexpr int D4S_LDS_BYTES = D4_NBUF * D4S_STAGE;   // 114 688 B

template <bool A_CELLS_ON_K>
__global__ __launch_bounds__(512) void dec_gemm320_dma4s_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char d4s_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int split = blockIdx.x % p.c_split_row, mtile = blockIdx.x / p.c_split_row;
  const int m0 = mtile * DG_BM;
  const int kbeg = split * p.k_per_split;
  const int Kpad = (p.K + DG_BK - 1) / DG_BK * DG_BK;   // (the operands are padded to multiples of 64 along K)
  int kend = kbeg + p.k_per_split;
  if (kend > Kpad) kend = Kpad;
  const int ntiles = (kend - kbeg) / D4S_BK;

  lds_byte* const lds = (lds_byte*)(d4s_smem);
  // A: per plane four half tiles of 1 KiB (stage tile i = the i-th 32-row block of the workgroup's M range; the K half alternates with
  // the stage index).  Wave w copies half tile w >> 1 of plane w & 1.
  //    d A_m: stage tile = cell tile, its K half = genes 16 (t & 1) .. + 15 = the first / second KiB of the tile;
  //    d W_m: stage tile = gene tile, its K half = cells 16 (t & 1) .. + 15 = the first / second KiB of the RE-ORDERED tile image
  //           [cell / 4][gene / 4][cell % 4][4 genes] (see dec_gemm320_dma_kernel), gathered 16 bytes at a time
  const glb_byte* srcA;
  long a_pair_step;   // byte advance of the A source per PAIR of stages (one 32-deep tile)
  {
    const long T = p.tiles_inner;
    long tile0;
    if constexpr (!A_CELLS_ON_K) { tile0 = ((long)(m0 / 32) + (wave >> 1)) * T + (kbeg / 32); a_pair_step = 2048; }
    else { tile0 = (long)(kbeg / 32) * T + (m0 / 32) + (wave >> 1); a_pair_step = T * 2048; }
    srcA = (glb_byte*)((wave & 1) ? p.A_lo : p.A) + tile0 * 2048;
  }
  // this lane's 16 bytes of the first / second K half of a tile
  int a_half_off[2];
#pragma unroll
  for (int hf = 0; hf < 2; ++hf) {
    int chunk = hf * 64 + lane;   // 16-byte chunk of the (LDS-order) tile image
    if constexpr (A_CELLS_ON_K) {
      const int cq = chunk >> 4, gp = (chunk >> 1) & 7, cpair = chunk & 1;
      chunk = (gp >> 1) * 32 + (gp & 1) * 16 + 2 * cq + cpair;
    }
    a_half_off[hf] = chunk * 16;
  }
  // B: per plane 16 rows x 640 B = 10 pieces; piece q = wave + 8 i (i = 0..2, the third only for waves 0..3): plane q / 10, piece q % 10
  const glb_byte* srcB[3];
  int dstB[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int q = (wave + 8 * i) % 20;   // (% only tames the non-existent third piece of waves 4..7)
    const int plane = q / 10, o = (q % 10) * 1024 + lane * 16;
    const int row = o / DG_B_ROW, w = o % DG_B_ROW;
    const int gran = (w >> 6) ^ ((row >> 1) & 1);
    srcB[i] = (glb_byte*)(plane ? p.B_lo : p.B) + ((long)(kbeg + row) * DG_B_ROW) + gran * 64 + (w & 63);
    dstB[i] = 2 * D4S_A_BYTES + plane * D4S_B_BYTES + (q % 10) * 1024;
  }
  auto issue = [&](int t) {
    const int stage = (t % D4_NBUF) * D4S_STAGE;
    dma16(srcA + (long)(t >> 1) * a_pair_step + a_half_off[t & 1], lds + stage + (wave & 1) * D4S_A_BYTES + (wave >> 1) * 1024);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      if (i == 2 && wave >= 4) break;   // (wave-uniform)
      dma16(srcB[i] + (long)t * D4S_B_BYTES, lds + stage + dstB[i]);
    }
  };

  f16v acc[5];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[j][q] = 0.f;

  for (int t = 0; t < D4_NBUF - 1 && t < ntiles; ++t) issue(t);
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, h = lane >> 5, r = lane & 31;
  const int in_gran = 32 * (gi & 1) + 8 * p4;
  const int fB = (q4 >> 1) & 1;
  const unsigned lds0 = lds_addr_of(d4s_smem);
  for (int t = 0; t < ntiles; ++t) {
    const int younger = min(ntiles - 1 - t, D4_NBUF - 2);
    if (younger == 2) { if (wave < 4) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); }
    else if (younger == 1) { if (wave < 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    if (t + D4_NBUF - 1 < ntiles) issue(t + D4_NBUF - 1);
    const unsigned st = lds0 + (t % D4_NBUF) * D4S_STAGE;
    s4v ra[2][2], rb[2][5][2];   // [plane][..]
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
      const unsigned stA = st + pl * D4S_A_BYTES, stB = st + 2 * D4S_A_BYTES + pl * D4S_B_BYTES;
      if constexpr (!A_CELLS_ON_K) {   // half tile wm: genes 8 h .. + 7 of cell r = two 8-byte pieces
        const unsigned ad = stA + wm * 1024 + h * 512 + r * 8;
        lds_read8(ra[pl][0], ad);
        lds_read8(ra[pl][1], ad + 256);
      } else {                         // half tile wm: cells 8 h + q4 (+ 4), gene piece 4 (gi & 1) + p4
        const unsigned ad = stA + wm * 1024 + (2 * h) * 256 + (4 * (gi & 1) + p4) * 32 + q4 * 8;
        tr_issue(ra[pl][0], ad);
        tr_issue(ra[pl][1], ad + 256);
      }
      const unsigned rowB = stB + (8 * h + q4) * DG_B_ROW + in_gran;
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        const unsigned ad = rowB + (((wn * 5 + j) ^ fB) * 64);
        tr_issue(rb[pl][j][0], ad);
        tr_issue(rb[pl][j][1], ad + 4 * DG_B_ROW);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[0][0]), "+v"(ra[0][1]), "+v"(ra[1][0]), "+v"(ra[1][1]),
                 "+v"(rb[0][0][0]), "+v"(rb[0][0][1]), "+v"(rb[0][1][0]), "+v"(rb[0][1][1]), "+v"(rb[0][2][0]), "+v"(rb[0][2][1]), "+v"(rb[0][3][0]), "+v"(rb[0][3][1]),
                 "+v"(rb[0][4][0]), "+v"(rb[0][4][1]), "+v"(rb[1][0][0]), "+v"(rb[1][0][1]), "+v"(rb[1][1][0]), "+v"(rb[1][1][1]), "+v"(rb[1][2][0]), "+v"(rb[1][2][1]),
                 "+v"(rb[1][3][0]), "+v"(rb[1][3][1]), "+v"(rb[1][4][0]), "+v"(rb[1][4][1]));
    const s8v a_hi = join8(ra[0][0], ra[0][1]), a_lo = join8(ra[1][0], ra[1][1]);
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      const s8v b_hi = join8(rb[0][j][0], rb[0][j][1]), b_lo = join8(rb[1][j][0], rb[1][j][1]);
      acc[j] = mfma32(a_hi, b_lo, acc[j]);   // small terms first, as the register-staged kernel orders them (spv_gemm.h)
      acc[j] = mfma32(a_lo, b_hi, acc[j]);
      acc[j] = mfma32(a_hi, b_hi, acc[j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  float* slab = p.C + (long)split * p.slab_stride;
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int col = wn * 160 + 32 * j + r;
    if (col >= p.N) continue;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int row = m0 + wm * 32 + crow(q, h);
      if (row < p.M) slab[(long)row * p.ldc + col] = acc[j][q];
    }
  }
  if (p.fix_cnt != nullptr) splitk_fixup_tile(p, mtile, m0, reinterpret_cast<volatile unsigned*>(d4s_smem));   // (kernel-uniform; the stages are dead)
}

// ---- regressor weight gradients of both rate heads in one pass --------------------------------------------------------------------
//   d [W'_p | c_p][gene][0..15] = sum_cell tP[cell][gene] * Aps[cell][0..15]
//   d [W'_s | c_s][gene][0..31] = sum_cell tS[cell][gene] * Aps[cell][16..47]        (backward of the two factor regressors,
//                                                                                     nn/networks.py:314-320, BatchNorm folded)
// tP / tS: the (softmax-corrected) gradients wrt the two heads' pre-softmax outputs, bf16 tile order; Aps: the latent operand
// image [Bp][48] bf16 = [z_private | 1 | 0.. | z_shared | 1 | 0..].  Pure streaming (168 MB per group for 0.8 GFLOP): what matters
// is that every byte of tP / tS crosses a CU once and with enough in flight.  Workgroup = 128 genes x both arrays, K tile = 64 cells:
// 32 KiB of tiles + 8 KiB of latents per stage, two stages = 80 KiB, so TWO workgroups share a CU (two tiles in flight per CU and
// one workgroup's fill overlaps the other's multiply; the first version -- 256 genes, one workgroup per CU -- ran at 42 us, 4.0 TB/s).
// 8 waves: wave w owns gene tile w & 3 of array w >> 2 (0: tP / private head, 1: tS / shared head): one MFMA 32x32x16 per 16-cell step
// (the private head multiplies against latent columns 0..31 and keeps 0..15).
// The latent rows are staged as TWO 64-byte-pitch images (columns 0..31 and 16..47): four consecutive k rows then cover the 256-byte
// bank period exactly and the transposed reads need no swizzle.
constexpr int DH_BM = 128, DH_BK = 64;
constexpr int DH_T_BYTES = DH_BM * DH_BK * 2, DH_L_BYTES = 2 * DH_BK * 64, DH_STAGE = 2 * DH_T_BYTES + DH_L_BYTES;   // 16 K + 16 K + 8 K
constexpr int DH_LDS_BYTES = 2 * DH_STAGE;   // 81 920 B: two workgroups per CU

// tP, tS: tiles, T = gene tiles per cell tile; Aps [>= K cells][48]; slabP [splits][G][16], slabS [splits][G][32];
// K = cells (multiple of 64), k_per_split multiple of 64.  grid = ceil(G / 128) * splits (split fastest), 512 threads.
__global__ __launch_bounds__(512, 2) void dec_heads_wgrad_dma_kernel(const bf16_t* tP, const bf16_t* tS, int T, const bf16_t* Aps, int G, int K, int k_per_split,
                                                                     int splits, float* slabP, float* slabS) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dh_smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_wave_id();
  const int arr = wave >> 2, gt = wave & 3;
  const int split = blockIdx.x % splits, mtile = blockIdx.x / splits;
  const int m0 = mtile * DH_BM;
  const int kbeg = split * k_per_split;
  int kend = kbeg + k_per_split;
  if (kend > K) kend = K;
  const int ntiles = (kend - kbeg) / DH_BK;

  lds_byte* const lds = (lds_byte*)(dh_smem);
  // stage image [array 2][cell tile 2][gene tile 4][2 KiB]; wave w copies ITS gene tile of ITS array for both cell tiles (4 pieces),
  // each tile re-ordered to [cell / 4][gene / 4][cell % 4][4 genes] (see dec_gemm320_dma_kernel)
  const glb_byte* srcT[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ct = i >> 1, half = i & 1;
    const int lc = half * 64 + lane, cq = lc >> 4, gp = (lc >> 1) & 7, cpair = lc & 1;
    const int chunk = (gp >> 1) * 32 + (gp & 1) * 16 + 2 * cq + cpair;
    const long tile = ((long)(kbeg / 32) + ct) * T + (m0 / 32) + gt;
    srcT[i] = (glb_byte*)(arr ? tS : tP) + tile * 2048 + chunk * 16;
  }
  const long t_step = 2L * T * 2048;
  // latent piece of this wave: waves 0..3 rows 16 w .. of the private image (source columns 0..31), waves 4..7 of the shared one (16..47)
  const int lrow = 16 * gt + (lane >> 2);
  const glb_byte* srcL = (glb_byte*)(Aps) + ((long)(kbeg + lrow) * DEC_KPS + arr * 16) * 2 + (lane & 3) * 16;
  auto issue = [&](int t) {
    const int stage = (t & 1) * DH_STAGE;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int ct = i >> 1, half = i & 1;
      dma16(srcT[i] + (long)t * t_step, lds + stage + arr * DH_T_BYTES + ((ct * 4 + gt) * 2 + half) * 1024);
    }
    dma16(srcL + (long)t * (DH_BK * DEC_KPS * 2), lds + stage + 2 * DH_T_BYTES + wave * 1024);
  };

  f16v acc;
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.f;
  if (ntiles > 0) issue(0);
  const int gi = lane >> 4, q4 = (lane & 15) >> 2, p4 = lane & 3, h = lane >> 5, r = lane & 31;
  const int in_gran = 32 * (gi & 1) + 8 * p4;
  const unsigned lds0 = lds_addr_of(dh_smem);
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    if (t + 1 < ntiles) issue(t + 1);
    const unsigned st = lds0 + (t & 1) * DH_STAGE, stT = st + arr * DH_T_BYTES, stL = st + 2 * DH_T_BYTES + arr * (DH_BK * 64);
    s4v a[4][2], b[4][2];
#pragma unroll
    for (int ks = 0; ks < DH_BK / 16; ++ks) {
      const unsigned adT = stT + ((ks >> 1) * 4 + gt) * 2048 + (4 * (ks & 1) + 2 * h) * 256 + (4 * (gi & 1) + p4) * 32 + q4 * 8;
      tr_issue(a[ks][0], adT); tr_issue(a[ks][1], adT + 256);
      const unsigned adL = stL + (16 * ks + 8 * h + q4) * 64 + in_gran;
      tr_issue(b[ks][0], adL); tr_issue(b[ks][1], adL + 256);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0][0]), "+v"(a[0][1]), "+v"(a[1][0]), "+v"(a[1][1]), "+v"(a[2][0]), "+v"(a[2][1]), "+v"(a[3][0]), "+v"(a[3][1]),
                 "+v"(b[0][0]), "+v"(b[0][1]), "+v"(b[1][0]), "+v"(b[1][1]), "+v"(b[2][0]), "+v"(b[2][1]), "+v"(b[3][0]), "+v"(b[3][1]));
#pragma unroll
    for (int ks = 0; ks < DH_BK / 16; ++ks) acc = mfma32(join8(a[ks][0], a[ks][1]), join8(b[ks][0], b[ks][1]), acc);
  }
#pragma unroll
  for (int q = 0; q < 16; ++q) {
    const int gene = m0 + gt * 32 + crow(q, h);
    if (gene < G) {
      if (arr == 0) { if (r < DEC_KP) slabP[((long)split * G + gene) * DEC_KP + r] = acc[q]; }
      else slabS[((long)split * G + gene) * DEC_KS + r] = acc[q];
    }
  }
}

// ---- mixing logits, bf16 mode --------------------------------------------------------------------------------------------------
//   out(cell, gene) = sum_k W_m[gene][k] * A_m[cell][k]      (nn/networks.py:322-325 with the bias folded into a ones column), K <= 320,
// written f16 in accumulator-tile order (genes on MFMA rows: the layout the likelihood kernel reads).  Both operands have K contiguous
// ("natural"): plain 16-byte fragment reads as in fc1_fwd_dma_kernel.  256 genes x 128 cells per workgroup, 8 waves as 4 (M) x 2 (N),
// wave tile 64 x 64; K tiles of 32 (64-byte LDS rows: 16 KiB + 8 KiB per stage), two stages = 48 KiB and <= 128 registers, so two
// workgroups share a CU: the K loop is only K / 32 = 10 tiles long and one workgroup's fill / epilogue overlaps the other's MFMAs.
// A 64-byte row holds four 16-byte chunks; chunk c of row r sits at position c ^ ((r >> 2) & 3): the 16 rows a ds_read_b128 lane
// group touches ({0-3, 12-15, 20-27} / {4-11, 16-19, 28-31}, MI355X_MICROARCH.md) then cover all 64 banks.
constexpr int DL_BM = 256, DL_BN = 128, DL_BK = 32;
constexpr int DL_A_BYTES = DL_BM * DL_BK * 2, DL_B_BYTES = DL_BN * DL_BK * 2, DL_STAGE = DL_A_BYTES + DL_B_BYTES;   // 16 K + 8 K
constexpr int DL_LDS_BYTES = 2 * DL_STAGE;

// Wm bf16 [Gp][K], Am bf16 [Bp][K] (K % 32 == 0, rows zero padded), T = Gp / 32, out f16 tiles.  grid = (Gp / 256, Bp / 128), 512 threads.
__device__ __forceinline__ void dec_logits_dma_body(const bf16_t* Wm, const bf16_t* Am, int K, int T, _Float16* out, unsigned char* dl_smem) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = uniform_wave_id();
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * DL_BM, n0 = blockIdx.y * DL_BN;
  const int ntiles = K / DL_BK;
  lds_byte* const lds = (lds_byte*)(dl_smem);
  // pieces of 1 KiB = 16 rows of 64 bytes: lane -> row 16 piece + (lane >> 2), position lane & 3 <- source chunk pos ^ ((row >> 2) & 3)
  const glb_byte* src[3];
  int dst[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int piece = (i < 2) ? 2 * wave + i : wave;          // A pieces 0..15 (two per wave), B pieces 0..7 (one per wave)
    const int row = 16 * piece + (lane >> 2), c = (lane & 3) ^ ((row >> 2) & 3);
    src[i] = (i < 2) ? (glb_byte*)(Wm) + ((long)(m0 + row) * K + 8 * c) * 2 : (glb_byte*)(Am) + ((long)(n0 + row) * K + 8 * c) * 2;
    dst[i] = (i < 2 ? 0 : DL_A_BYTES) + piece * 1024;
  }
  auto issue = [&](int t) {
    const int stage = (t & 1) * DL_STAGE;
#pragma unroll
    for (int i = 0; i < 3; ++i) dma16(src[i] + (long)t * (DL_BK * 2), lds + stage + dst[i]);
  };
  f16v acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
  issue(0);
  const int r = lane & 31, h = lane >> 5;
  const int a_row = wm * 64 + r, b_row = wn * 64 + r;   // rows + 32 for the second tile: (row >> 2) & 3 is the same for both
  const int sw = (r >> 2) & 3;
  for (int t = 0; t < ntiles; ++t) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    raw_barrier();
    if (t + 1 < ntiles) issue(t + 1);
    const unsigned char* st = dl_smem + (t & 1) * DL_STAGE;
#pragma unroll
    for (int ks = 0; ks < DL_BK / 16; ++ks) {
      const int cpos = ((2 * ks + h) ^ sw) * 16;
      s8v a[2], b[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) a[i] = *reinterpret_cast<const s8v*>(st + (a_row + 32 * i) * 64 + cpos);
#pragma unroll
      for (int j = 0; j < 2; ++j) b[j] = *reinterpret_cast<const s8v*>(st + DL_A_BYTES + (b_row + 32 * j) * 64 + cpos);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = mfma32(a[i], b[j], acc[i][j]);
    }
  }
  typedef __attribute__((ext_vector_type(4))) _Float16 h4v;
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const long gt = (m0 + wm * 64 + 32 * i) / 32, ct = (n0 + wn * 64 + 32 * j) / 32;
      _Float16* o = out + (ct * T + gt) * 1024 + lane * 4;
#pragma unroll
      for (int qq = 0; qq < 4; ++qq) {
        const int q = 4 * qq;
        *reinterpret_cast<h4v*>(o + qq * 256) = h4v{(_Float16)acc[i][j][q], (_Float16)acc[i][j][q + 1], (_Float16)acc[i][j][q + 2], (_Float16)acc[i][j][q + 3]};
      }
    }
}
__global__ __launch_bounds__(512, 2) void dec_logits_dma_kernel(const bf16_t* Wm, const bf16_t* Am, int K, int T, _Float16* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char dl_smem_[];
  dec_logits_dma_body(Wm, Am, K, T, out, dl_smem_);
}
struct LogitsArgs { const bf16_t* Wm; const bf16_t* Am; int K, T, Bp; _Float16* out; };
__global__ __launch_bounds__(512, 2) void dec_logits_dma_pair_kernel(LogitsArgs a0, LogitsArgs a1) {   // group = blockIdx.z
  extern __shared__ __attribute__((aligned(16))) unsigned char dl_smem_[];
  const LogitsArgs a = blockIdx.z ? a1 : a0;
  if ((int)blockIdx.x * DL_BM >= a.T * 32 || (int)blockIdx.y * DL_BN >= a.Bp) return;   // (workgroup-uniform)
  dec_logits_dma_body(a.Wm, a.Am, a.K, a.T, a.out, dl_smem_);
}

}  // namespace spv
