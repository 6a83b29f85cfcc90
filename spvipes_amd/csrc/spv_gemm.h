// Tiled bf16 MFMA GEMM core shared by the encoder / decoder contractions.
//
//   C[m][n] (+)= sum_k A(m,k) * B(k,n)        fp32 accumulate, fp32 output
//
// Operand sources
//   SRC_PLAIN : a bf16 image in HBM (optionally a hi/lo pair, NSPLIT == 3), zero padded by the
//               producer so that every tile the grid touches is in bounds.
//               natural  : mem[row][k]   (k contiguous)   -> 16-byte fragment reads from LDS
//               k-major  : mem[k][col]   (col contiguous) -> ds_read_b64_tr_b16 fragment reads
//   SRC_COUNTS: the resident count matrix X[cell][gene] (f32 or u16), rows gathered through an
//               index vector, log1p applied in registers on the way to LDS (A1 of SURVEY.md 8a:
//               module/spVIPESmodule.py:428-435).  As a natural A operand (m = cell, k = gene)
//               it also produces the per-cell sum of log1p(x) (the library, :435).
//               As a k-major B operand (k = cell, n = gene) it feeds the fc1 weight gradient.
//
//   SRC_TILED : a [cells][genes] array kept in "accumulator-tile" order
//               T[cell/32][gene/32][qq][lane = cell%32 + 32h][j],  gene%32 = 8 qq + 4 h + j,
//               i.e. each 32x32 tile is stored as the MFMA accumulator that produced it (genes on
//               MFMA rows, cells on MFMA columns; register q = 4 qq + j), register-group major: one
//               wave instruction moving registers 4qq..4qq+3 of every lane touches 64 x 8 B = 512
//               contiguous bytes (16-bit elements) instead of 32 strided row pieces.
//
// 256 threads = 4 waves arranged WM x WN; each wave owns (BM/WM) x (BN/WN) of the tile as TM x TN
// MFMA 32x32 tiles.  BK = 32 or 64.  Global->register prefetch of tile t+1 overlaps the MFMAs of tile t
// (register staging: the count operand has to pass through VALU for log1p anyway).
#pragma once
#include "spv_common.h"

namespace spv {

enum { SRC_PLAIN = 0, SRC_COUNTS = 1, SRC_TILED = 2, SRC_GATHER = 3 };
// SRC_GATHER: bf16 image of the WHOLE data set [all cells][ld] (zero padded along the genes), the minibatch's rows picked
// through p.rows exactly like SRC_COUNTS does -- the precomputed bf16 log1p(x) of a resident count matrix.
enum { EPI_STORE = 0, EPI_ATOMIC = 1, EPI_TILED_F16 = 2, EPI_TILED_F32 = 3 };

struct GemmParams {
  const void* A; const void* A_lo; long lda;
  const void* B; const void* B_lo; long ldb;
  const int* rows;   // gather index of the count operand's cells (nullable = identity)
  int col_off;       // first gene column of this group inside the count matrix
  int n_cells;       // logical number of cells (M for natural-A counts, K for k-major-B counts)
  int n_genes;       // logical number of genes (K for natural-A counts, N for k-major-B counts)
  float* rowsum;     // [splits][M] partial sums of log1p(x) (natural-A counts), nullable
  float* C; long ldc; long slab_stride;
  float* C2; int c_split_row;   // optional: rows >= c_split_row of an EPI_STORE output go to C2 (row - c_split_row), same ldc
  const float* out_scale;       // optional (EPI_STORE): device scalar the accumulators are multiplied by on the way out (1 / scale of an f16 dh image)
  int M, N, K;
  int k_per_split;   // multiple of 32
  int epi;
  int tiles_inner;   // gene tiles per cell tile of a SRC_TILED operand / EPI_TILED_* output (= Gp / 32)
  bf16_t* xb_out; long ld_xb;  // optional (natural-A counts, NSPLIT 1): bf16 log1p(x) image [cells][ld_xb] written as a by-product
  int counts_aligned;  // count matrix base, row pitch and col_off all multiples of 16 bytes: every 8-gene chunk is one 16-B load
  // optional in-launch split-K fix-up (spv_gemm_fixup; LDS-DMA 320-column kernels only): fix_cnt != nullptr turns it on
  unsigned* fix_cnt; const float* fix_alpha; float* fix_d0; long fix_ld0; int fix_n0; float* fix_d1; long fix_ld1; int fix_c1, fix_n1;
};

__host__ __device__ constexpr int kmajor_pitch(int cols) {
  // pitch (bf16 elements) whose dword count is 16 or 48 mod 64: the four k-rows a transposed
  // read touches then fall on disjoint bank ranges (MI355X LDS: 64 banks for ds_read_b64_tr_b16)
  int p = cols / 2;
  while ((p % 64) != 16 && (p % 64) != 48) p += 4;
  return p * 2;
}
__host__ __device__ constexpr int nat_pitch(int bk) { return bk + 8; }  // BK k + 8 pad (16-B aligned rows, conflict-free 16-B reads)

// PF_: operand tiles each thread keeps in flight in registers (HBM latency is ~10 MFMA phases of a tile: one tile
// ahead leaves the kernel latency-bound at ~1 workgroup per CU).  With PF_ > 1 the LDS image is double buffered
// when it fits, which also drops one of the two barriers per tile.
// HALF_: the 16-bit operand words are IEEE f16 instead of bf16 (the encoder's first layer in the NSPLIT == 1 mode, spv_common.h):
// counts decode to f16(log1p), the MFMA is v_mfma_f32_32x32x16_f16.
template <int BM_, int BN_, int WM_, int WN_, bool A_KMAJ_, bool B_KMAJ_, int A_SRC_, int B_SRC_, typename CT_, int NSPLIT_, int BK_ = 32, int PF_ = 1, int OCC_ = 1, bool HALF_ = false>
struct GemmCfg {
  static constexpr bool HALF = HALF_;
  static_assert(!HALF_ || NSPLIT_ == 1, "f16 operands have no hi/lo split");
  static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_, BK = BK_, PF = PF_, OCC = OCC_;  // OCC: waves per SIMD to fit (register budget)
  static constexpr int NAT_PITCH = nat_pitch(BK_);
  static constexpr bool A_KMAJ = A_KMAJ_, B_KMAJ = B_KMAJ_;
  static constexpr int A_SRC = A_SRC_, B_SRC = B_SRC_, NSPLIT = NSPLIT_;
  typedef CT_ CT;
  static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;
  static constexpr int A_PITCH = A_KMAJ ? kmajor_pitch(BM) : NAT_PITCH;
  static constexpr int B_PITCH = B_KMAJ ? kmajor_pitch(BN) : NAT_PITCH;
  static constexpr int A_ELEMS = A_KMAJ ? BK * A_PITCH : BM * NAT_PITCH;
  static constexpr int B_ELEMS = B_KMAJ ? BK * B_PITCH : BN * NAT_PITCH;
  static constexpr int NIMG = (NSPLIT == 3) ? 2 : 1;
  static constexpr int STAGE_ELEMS = (A_ELEMS + B_ELEMS) * NIMG;
  static constexpr int NBUF = (PF_ > 1 && STAGE_ELEMS * 4 <= 96 * 1024) ? 2 : 1;
  static constexpr int LDS_BYTES = STAGE_ELEMS * 2 * NBUF;
  // 16-byte chunks each thread moves per tile
  static constexpr int A_CHUNKS = (BM * BK / 8 + 255) / 256;
  static constexpr int B_CHUNKS = (BN * BK / 8 + 255) / 256;
  static_assert(WM * WN == 4, "4 waves");
  static_assert(BM % (32 * WM) == 0 && BN % (32 * WN) == 0, "tile shape");
};

// ---- staging: one 16-byte chunk (8 bf16) of an operand tile ------------------
// chunk c of a tile with `rows` x `cols8` chunks (cols8 = fast-dimension chunks per slow row)
template <typename Cfg, bool KMAJ, int SRC, int EXT /*BM or BN*/>
struct Stager {
  static constexpr int FAST8 = KMAJ ? EXT / 8 : Cfg::BK / 8;   // chunks along the contiguous dim
  static constexpr int SLOW = KMAJ ? Cfg::BK : EXT;            // rows of the LDS image
  static constexpr int NCHUNKS = SLOW * FAST8;
  static constexpr int NCH = (NCHUNKS + 255) / 256;
  static constexpr int PITCH = KMAJ ? kmajor_pitch(EXT) : Cfg::NAT_PITCH;
  // per-chunk register payload
  u4v hi[NCH], lo[NCH];
  float csum[NCH];
  int ridx[NCH];    // SRC_COUNTS: row of the count matrix this thread's chunk i reads next (prefetched: see prep_rows)
  bool valid[NCH];  // SRC_COUNTS: chunk holds real counts (else it decodes as zeros)

  // slow/fast coordinates of chunk i for this thread
  __device__ __forceinline__ static void coord(int i, int tid, int& s, int& f, bool& ok) {
    const int c = tid + 256 * i;
    ok = c < NCHUNKS;
    s = c / FAST8;   // row of the LDS image
    f = c % FAST8;   // 16-byte chunk within the row: consecutive lanes fill a row contiguously (conflict-free stores)
  }

  // tile origin: `ext0` along the M/N dimension, `k0` along K
  __device__ __forceinline__ void load(const GemmParams& p, const void* ptr, const void* ptr_lo, long ld, int ext0, int k0, int tid) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int s, f; bool ok;
      coord(i, tid, s, f, ok);
      hi[i] = u4v{0u, 0u, 0u, 0u};
      lo[i] = u4v{0u, 0u, 0u, 0u};
      csum[i] = 0.f;
      valid[i] = false;
      if (!ok) continue;
      if constexpr (SRC == SRC_TILED) {
        // The source is stored in 32x32 accumulator tiles T[cell/32][gene/32][qq][lane = cell%32 + 32h][j], gene%32 = 8qq+4h+j.
        // A chunk = 8 consecutive genes (8qq .. 8qq+7) of one cell = the 8-byte words of storage lanes r and r + 32.
        // k-major (k = cell, ext = gene): image row s = cell, f = gene chunk; natural (ext = cell, k = gene): row s = cell too.
        const int r = s & 31, st = s >> 5, ft = f >> 2, qq = f & 3;   // cell within its tile / cell tile / gene tile / gene chunk
        const long ct = KMAJ ? k0 / 32 + st : ext0 / 32 + st, gt = KMAJ ? ext0 / 32 + ft : k0 / 32 + ft;
        const long off = (ct * p.tiles_inner + gt) * 1024 + qq * 256 + r * 4;
        const u2v a = *reinterpret_cast<const u2v*>(reinterpret_cast<const bf16_t*>(ptr) + off);
        const u2v b = *reinterpret_cast<const u2v*>(reinterpret_cast<const bf16_t*>(ptr) + off + 128);
        hi[i] = u4v{a[0], a[1], b[0], b[1]};
        if constexpr (Cfg::NSPLIT == 3) {
          const u2v c2 = *reinterpret_cast<const u2v*>(reinterpret_cast<const bf16_t*>(ptr_lo) + off);
          const u2v d2 = *reinterpret_cast<const u2v*>(reinterpret_cast<const bf16_t*>(ptr_lo) + off + 128);
          lo[i] = u4v{c2[0], c2[1], d2[0], d2[1]};
        }
      } else if constexpr (SRC == SRC_GATHER) {
        const int cell = KMAJ ? k0 + s : ext0 + s;
        const int gene = KMAJ ? ext0 + 8 * f : k0 + 8 * f;
        valid[i] = cell < p.n_cells;   // rows beyond the minibatch read row ridx = 0 and are zeroed in store()
        hi[i] = *reinterpret_cast<const u4v*>(reinterpret_cast<const bf16_t*>(ptr) + (long)ridx[i] * ld + gene);
      } else if constexpr (SRC == SRC_PLAIN) {
        // natural: mem[ext0 + s][k0 + 8f]; k-major: mem[k0 + s][ext0 + 8f]; producer-padded
        const long off = KMAJ ? (long)(k0 + s) * ld + ext0 + 8 * f : (long)(ext0 + s) * ld + k0 + 8 * f;
        hi[i] = *reinterpret_cast<const u4v*>(reinterpret_cast<const bf16_t*>(ptr) + off);
        if constexpr (Cfg::NSPLIT == 3) lo[i] = *reinterpret_cast<const u4v*>(reinterpret_cast<const bf16_t*>(ptr_lo) + off);
      } else {
        // counts: slow index = cell, fast index = gene.  Only the RAW words are fetched here (u16: 8 counts in
        // hi[i]; f32: 4 + 4 floats in hi[i], lo[i]); log1p / bf16 split happen in store(), PF tiles later, so the
        // gather's latency hides under the MFMAs in between.  Branch-free on the fast path (a divergent branch
        // around a load makes the compiler wait for it inside the branch, which drains every prefetch in flight);
        // lanes with nothing to read fetch the first element of the matrix and store() zeroes them.
        typedef typename Cfg::CT CT;
        const int cell = KMAJ ? k0 + s : ext0 + s;
        const int gene = KMAJ ? ext0 + 8 * f : k0 + 8 * f;
        const bool inb = cell < p.n_cells && gene < p.n_genes;
        const bool full = gene + 8 <= p.n_genes;
        const CT* src = reinterpret_cast<const CT*>(ptr) + (long)ridx[i] * ld + p.col_off + gene;
        valid[i] = inb;
        if (p.counts_aligned) {  // uniform
          const CT* s2 = (inb && full) ? src : reinterpret_cast<const CT*>(ptr);
          hi[i] = *reinterpret_cast<const u4v*>(s2);
          if constexpr (sizeof(CT) == 4) lo[i] = *reinterpret_cast<const u4v*>(s2 + 4);
        }
        const bool slow = inb && !(p.counts_aligned && full);  // the group's last, partial chunk -- or an unaligned matrix
        if (__builtin_expect(__any(slow), 0)) {
          if (slow) {
            if constexpr (sizeof(CT) == 2) {
              unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (gene + j < p.n_genes) w[j >> 1] |= (unsigned)reinterpret_cast<const unsigned short*>(src)[j] << (16 * (j & 1));
              hi[i] = u4v{w[0], w[1], w[2], w[3]};
            } else {
              unsigned w[8] = {0u, 0u, 0u, 0u, 0u, 0u, 0u, 0u};
#pragma unroll
              for (int j = 0; j < 8; ++j)
                if (gene + j < p.n_genes) w[j] = __float_as_uint(reinterpret_cast<const float*>(src)[j]);
              hi[i] = u4v{w[0], w[1], w[2], w[3]};
              lo[i] = u4v{w[4], w[5], w[6], w[7]};
            }
          }
        }
      }
    }
    if constexpr ((SRC == SRC_COUNTS || SRC == SRC_GATHER) && KMAJ) prep_rows(p, ext0, k0 + Cfg::PF * Cfg::BK, tid);  // this stage's next tile
  }

  // row indices of the count chunks of tile (ext0, k0): fetched one use ahead so that load() never waits on them
  __device__ __forceinline__ void prep_rows(const GemmParams& p, int ext0, int k0, int tid) {
    if constexpr (SRC == SRC_COUNTS || SRC == SRC_GATHER) {
#pragma unroll
      for (int i = 0; i < NCH; ++i) {
        int s, f; bool ok;
        coord(i, tid, s, f, ok);
        const int cell = KMAJ ? k0 + s : ext0 + s;
        const int c2 = (ok && cell < p.n_cells) ? cell : 0;  // clamped: the load below is unconditional (no wait at issue)
        if (p.rows) ridx[i] = p.rows[c2];
        else ridx[i] = c2;
      }
    }
  }

  // decode one raw count chunk into log1p values, their bf16 hi/lo split and their sum
  __device__ __forceinline__ static void decode_counts(const u4v& r0, const u4v& r1, u4v& o_hi, u4v& o_lo, float& sum) {
    float v[8];
    if constexpr (sizeof(typename Cfg::CT) == 2) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[2 * j] = (float)(r0[j] & 0xFFFFu); v[2 * j + 1] = (float)(r0[j] >> 16); }
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = __uint_as_float(r0[j]); v[4 + j] = __uint_as_float(r1[j]); }
    }
    sum = 0.f;
    unsigned hw[4], lw[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = log1p_count(v[2 * j]), x1 = log1p_count(v[2 * j + 1]);
      sum += x0 + x1;
      if constexpr (Cfg::HALF) {
        hw[j] = pack2h(x0, x1);
        lw[j] = 0u;
        continue;
      }
      bf16_t h0, l0, h1, l1;
      split_bf16(x0, h0, l0);
      split_bf16(x1, h1, l1);
      hw[j] = (unsigned)h0 | ((unsigned)h1 << 16);
      lw[j] = (unsigned)l0 | ((unsigned)l1 << 16);
    }
    o_hi = u4v{hw[0], hw[1], hw[2], hw[3]};
    o_lo = u4v{lw[0], lw[1], lw[2], lw[3]};
  }

  // (ext0, k0): origin of the tile being stored (only used by the log1p by-product)
  __device__ __forceinline__ void store(bf16_t* img_hi, bf16_t* img_lo, int tid, const GemmParams& p, int ext0, int k0) {
#pragma unroll
    for (int i = 0; i < NCH; ++i) {
      int s, f; bool ok;
      coord(i, tid, s, f, ok);
      if (!ok) continue;
      if constexpr (SRC == SRC_COUNTS) {
        u4v o_hi, o_lo;
        const u4v zero = u4v{0u, 0u, 0u, 0u};
        decode_counts(valid[i] ? hi[i] : zero, valid[i] ? lo[i] : zero, o_hi, o_lo, csum[i]);
        *reinterpret_cast<u4v*>(img_hi + s * PITCH + 8 * f) = o_hi;
        if constexpr (Cfg::NSPLIT == 3) *reinterpret_cast<u4v*>(img_lo + s * PITCH + 8 * f) = o_lo;
        if constexpr (!KMAJ && Cfg::NSPLIT == 1) {  // the weight-gradient GEMM re-reads log1p(x) as plain bf16 instead of decoding the counts again
          if (p.xb_out != nullptr && blockIdx.y == 0 && ext0 + s < p.n_cells) *reinterpret_cast<u4v*>(p.xb_out + (long)(ext0 + s) * p.ld_xb + k0 + 8 * f) = o_hi;
        }
        continue;
      }
      if constexpr (SRC == SRC_GATHER) {
        *reinterpret_cast<u4v*>(img_hi + s * PITCH + 8 * f) = valid[i] ? hi[i] : u4v{0u, 0u, 0u, 0u};
        continue;
      }
      *reinterpret_cast<u4v*>(img_hi + s * PITCH + 8 * f) = hi[i];
      if constexpr (Cfg::NSPLIT == 3) *reinterpret_cast<u4v*>(img_lo + s * PITCH + 8 * f) = lo[i];
    }
  }
};

template <typename Cfg>
__device__ __forceinline__ void gemm_body(const GemmParams& p, const int bx, unsigned char* smem_raw) {   // bx: this workgroup's M tile
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* sA_lo = sA + Cfg::A_ELEMS * (Cfg::NIMG - 1);
  bf16_t* sB = sA + Cfg::A_ELEMS * Cfg::NIMG;
  bf16_t* sB_lo = sB + Cfg::B_ELEMS * (Cfg::NIMG - 1);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / Cfg::WN, wn = wave % Cfg::WN;
  const int m0 = bx * Cfg::BM, n0 = blockIdx.y * Cfg::BN;
  const int split = blockIdx.z;
  const int kbeg = split * p.k_per_split;
  int kend = kbeg + p.k_per_split;
  const int Kpad = (p.K + Cfg::BK - 1) / Cfg::BK * Cfg::BK;
  if (kend > Kpad) kend = Kpad;

  f16v acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;

  typedef Stager<Cfg, Cfg::A_KMAJ, Cfg::A_SRC, Cfg::BM> StA;
  typedef Stager<Cfg, Cfg::B_KMAJ, Cfg::B_SRC, Cfg::BN> StB;
  StA stA[Cfg::PF];
  StB stB[Cfg::PF];
  float rowsum_acc[StA::NCH];
#pragma unroll
  for (int i = 0; i < StA::NCH; ++i) rowsum_acc[i] = 0.f;

#pragma unroll
  for (int s = 0; s < Cfg::PF; ++s)
    if (kbeg + s * Cfg::BK < kend) {
      stA[s].prep_rows(p, m0, kbeg + s * Cfg::BK, tid);
      stB[s].prep_rows(p, n0, kbeg + s * Cfg::BK, tid);
      stA[s].load(p, p.A, p.A_lo, p.lda, m0, kbeg + s * Cfg::BK, tid);
      stB[s].load(p, p.B, p.B_lo, p.ldb, n0, kbeg + s * Cfg::BK, tid);
    }
  int buf = 0;
  for (int k0 = kbeg; k0 < kend; k0 += Cfg::PF * Cfg::BK) {
#pragma unroll
    for (int s = 0; s < Cfg::PF; ++s) {
      const int k = k0 + s * Cfg::BK;
      if (k < kend) {  // uniform
        const int bo = buf * Cfg::STAGE_ELEMS;
        bf16_t* const a_hi_img = sA + bo; bf16_t* const a_lo_img = sA_lo + bo;
        bf16_t* const b_hi_img = sB + bo; bf16_t* const b_lo_img = sB_lo + bo;
        // single buffer: wait until the previous tile's fragment reads are done.  Double buffer: this image was last
        // read two tiles ago and every wave has passed the previous tile's barrier since.
        if constexpr (Cfg::NBUF == 1) __syncthreads();
        stA[s].store(a_hi_img, a_lo_img, tid, p, m0, k);
        stB[s].store(b_hi_img, b_lo_img, tid, p, n0, k);
        if constexpr (Cfg::A_SRC == SRC_COUNTS && !Cfg::A_KMAJ) {
#pragma unroll
          for (int i = 0; i < StA::NCH; ++i) rowsum_acc[i] += stA[s].csum[i];
        }
        __syncthreads();
        if (k + Cfg::PF * Cfg::BK < kend) {  // refill this register stage; lands under the MFMAs of the next PF tiles
          stA[s].load(p, p.A, p.A_lo, p.lda, m0, k + Cfg::PF * Cfg::BK, tid);
          stB[s].load(p, p.B, p.B_lo, p.ldb, n0, k + Cfg::PF * Cfg::BK, tid);
        }
#pragma unroll
        for (int ks = 0; ks < Cfg::BK; ks += 16) {
          s8v a_hi[Cfg::TM], a_lo[Cfg::TM];
#pragma unroll
          for (int i = 0; i < Cfg::TM; ++i) {
            const int r0 = wm * (Cfg::BM / Cfg::WM) + 32 * i;
            if constexpr (Cfg::A_KMAJ) {
              a_hi[i] = frag_kmajor(a_hi_img, Cfg::A_PITCH, r0, ks, lane);
              if constexpr (Cfg::NSPLIT == 3) a_lo[i] = frag_kmajor(a_lo_img, Cfg::A_PITCH, r0, ks, lane);
            } else {
              a_hi[i] = frag_natural(a_hi_img, Cfg::A_PITCH, r0, ks, lane);
              if constexpr (Cfg::NSPLIT == 3) a_lo[i] = frag_natural(a_lo_img, Cfg::A_PITCH, r0, ks, lane);
            }
          }
#pragma unroll
          for (int j = 0; j < Cfg::TN; ++j) {
            const int c0 = wn * (Cfg::BN / Cfg::WN) + 32 * j;
            s8v b_hi, b_lo;
            if constexpr (Cfg::B_KMAJ) {
              b_hi = frag_kmajor(b_hi_img, Cfg::B_PITCH, c0, ks, lane);
              if constexpr (Cfg::NSPLIT == 3) b_lo = frag_kmajor(b_lo_img, Cfg::B_PITCH, c0, ks, lane);
            } else {
              b_hi = frag_natural(b_hi_img, Cfg::B_PITCH, c0, ks, lane);
              if constexpr (Cfg::NSPLIT == 3) b_lo = frag_natural(b_lo_img, Cfg::B_PITCH, c0, ks, lane);
            }
#pragma unroll
            for (int i = 0; i < Cfg::TM; ++i) {
              if constexpr (Cfg::HALF) acc[i][j] = mfma32h(a_hi[i], b_hi, acc[i][j]);
              else acc[i][j] = mfma32_split<Cfg::NSPLIT>(a_hi[i], a_lo[i], b_hi, b_lo, acc[i][j]);
            }
          }
        }
        if constexpr (Cfg::NBUF == 2) buf ^= 1;
      }
    }
  }

  // ---- epilogue ---------------------------------------------------------------
  float* C = p.C + (long)split * p.slab_stride;
  const int h = lane >> 5, r = lane & 31;
  if (p.epi == EPI_TILED_F16 || p.epi == EPI_TILED_F32) {
    // M = genes, N = cells: every 32x32 accumulator tile goes out as 64 lanes x 16 registers
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int j = 0; j < Cfg::TN; ++j) {
        const long gt = (m0 + wm * (Cfg::BM / Cfg::WM) + 32 * i) / 32, ct = (n0 + wn * (Cfg::BN / Cfg::WN) + 32 * j) / 32;
        const long off = (ct * p.tiles_inner + gt) * 1024 + lane * 4;
#pragma unroll
        for (int qq = 0; qq < 4; ++qq) {
          const int q = 4 * qq;
          if (p.epi == EPI_TILED_F32) {
            *reinterpret_cast<f4v*>(p.C + off + qq * 256) = f4v{acc[i][j][q], acc[i][j][q + 1], acc[i][j][q + 2], acc[i][j][q + 3]};
          } else {
            typedef __attribute__((ext_vector_type(4))) _Float16 h4v;
            const h4v v = {(_Float16)acc[i][j][q], (_Float16)acc[i][j][q + 1], (_Float16)acc[i][j][q + 2], (_Float16)acc[i][j][q + 3]};
            *reinterpret_cast<h4v*>(reinterpret_cast<_Float16*>(p.C) + off + qq * 256) = v;
          }
        }
      }
    return;
  }
  const float oscale = *(p.out_scale ? p.out_scale : &g_spv_one);
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j) {
      const int col = n0 + wn * (Cfg::BN / Cfg::WN) + 32 * j + r;
#pragma unroll
      for (int q = 0; q < 16; ++q) {
        const int row = m0 + wm * (Cfg::BM / Cfg::WM) + 32 * i + crow(q, h);
        if (row < p.M && col < p.N) {
          float* dst = (p.C2 != nullptr && row >= p.c_split_row) ? p.C2 + (long)(row - p.c_split_row) * p.ldc + col : C + (long)row * p.ldc + col;
          if (p.epi == EPI_ATOMIC) atomicAdd(dst, acc[i][j][q] * oscale);
          else *dst = acc[i][j][q] * oscale;
        }
      }
    }
  if constexpr (Cfg::A_SRC == SRC_COUNTS && !Cfg::A_KMAJ) {
    // per-cell sum of log1p(x): the 4 threads of a row hold disjoint 8-gene chunks
    if (p.rowsum != nullptr && blockIdx.y == 0) {
#pragma unroll
      for (int i = 0; i < StA::NCH; ++i) {
        float v = rowsum_acc[i];
        v += __shfl_xor(v, 1, 64);
        v += __shfl_xor(v, 2, 64);
        int s, f; bool ok;
        StA::coord(i, tid, s, f, ok);
        if (ok && f == 0 && m0 + s < p.M) p.rowsum[(long)split * p.M + m0 + s] = v;
      }
    }
  }
}
template <typename Cfg>
__global__ __launch_bounds__(256, Cfg::OCC) void gemm_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw_[];
  gemm_body<Cfg>(p, blockIdx.x, smem_raw_);
}
// two problems of one configuration in ONE grid (workgroups 0 .. nx0 - 1 along x: the first problem's M tiles); the y extent covers the
// wider problem's N tiles, both problems have the same split count (z)
template <typename Cfg>
__global__ __launch_bounds__(256, Cfg::OCC) void gemm_pair_kernel(GemmParams p0, GemmParams p1, int nx0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw_[];
  const bool first = (int)blockIdx.x < nx0;   // (workgroup-uniform)
  const GemmParams p = first ? p0 : p1;
  if ((int)blockIdx.y * Cfg::BN >= p.N) return;   // (the grid's y extent is the larger problem's)
  gemm_body<Cfg>(p, first ? (int)blockIdx.x : (int)blockIdx.x - nx0, smem_raw_);
}

template <typename Cfg>
inline int launch_gemm_pair(const GemmParams& p0, const GemmParams& p1, int splits, hipStream_t stream) {
  const int nx0 = (p0.M + Cfg::BM - 1) / Cfg::BM, nx1 = (p1.M + Cfg::BM - 1) / Cfg::BM;
  const int nmax = p0.N > p1.N ? p0.N : p1.N;
  dim3 grid(nx0 + nx1, (nmax + Cfg::BN - 1) / Cfg::BN, splits);
  if constexpr (Cfg::LDS_BYTES > 65536) {
    static bool raised = false;
    if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pair_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES); raised = true; }
  }
  hipLaunchKernelGGL(gemm_pair_kernel<Cfg>, grid, dim3(256), Cfg::LDS_BYTES, stream, p0, p1, nx0);
  return hipGetLastError() == hipSuccess ? SPV_OK : SPV_ERR_LAUNCH;
}

template <typename Cfg>
inline int launch_gemm(const GemmParams& p, int splits, hipStream_t stream) {
  dim3 grid((p.M + Cfg::BM - 1) / Cfg::BM, (p.N + Cfg::BN - 1) / Cfg::BN, splits);
  if constexpr (Cfg::LDS_BYTES > 65536) {
    static bool raised = false;  // one-time opt-in to more than 64 KiB of dynamic LDS for this instantiation
    if (!raised) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_kernel<Cfg>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::LDS_BYTES); raised = true; }
  }
  hipLaunchKernelGGL(gemm_kernel<Cfg>, grid, dim3(256), Cfg::LDS_BYTES, stream, p);
  return hipGetLastError() == hipSuccess ? SPV_OK : SPV_ERR_LAUNCH;
}

}  // namespace spv
