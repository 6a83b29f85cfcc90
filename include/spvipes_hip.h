/* spvipes_hip.h -- C ABI of the MI355X (gfx950) hot-path library  libspvipes_hip.so
 *
 * Drop-in boundary for the per-minibatch VAE step of nrclaudio/spVIPES.  The reference has no
 * native code and no FFI of its own: its hot path is a sequence of stock torch ops inside
 * src/spVIPES/module/spVIPESmodule.py and src/spVIPES/nn/networks.py.  Each entry point below
 * names the reference lines whose arithmetic it replaces; spvipes_amd/_abi.py is the ctypes
 * binding, INTEGRATION.md shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless marked host; no allocation happens inside
 *     (workspaces are passed in); every call only enqueues work on `stream`
 *     (a hipStream_t passed as void*; NULL = the default stream) and returns immediately;
 *   - return value: SPV_OK (0) or a negative SPV_ERR_*; spv_last_error() gives the text;
 *   - thread-compatible, not thread-safe; no exceptions cross the ABI;
 *   - "bf16" = raw 16-bit words (uint16_t).  Packed operand images are zero padded by
 *     spv_pack_bf16 so that whole tiles are always in bounds;
 *   - precision mode `nsplit`: 1 = bf16 operands, fp32 accumulate;
 *                              3 = split-bf16 (hi/lo) operands, ~fp32 products ("fp32" mode).
 */
#ifndef SPVIPES_HIP_H
#define SPVIPES_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SPV_OK 0
#define SPV_ERR_ARG (-1)
#define SPV_ERR_LAUNCH (-2)
#define SPV_ERR_UNSUPPORTED (-3)

#define SPV_COUNT_F32 0
#define SPV_COUNT_U16 1

#define SPV_NB_CMAX 64      /* rows of the (count, gene) lgamma/digamma table         */
#define SPV_DEC_KP 16       /* K slots of the private factor regressor (n_p + 1 <= 16) */
#define SPV_DEC_KS 32       /* K slots of the shared factor regressor  (n_s + 1 <= 32) */
#define SPV_DEC_CELLS_PER_WG 128

int spv_version(void);
const char* spv_last_error(void);
/* Fingerprint (32 hex digits) of the sources and compiler flags this library was built from; the Python binding refuses a
 * library whose fingerprint differs from the sources next to it (spvipes_amd/build.py: build_id). */
const char* spv_build_id(void);

/* Count matrix of one group: X[cell][gene], row-major, resident in HBM.
 * rows == NULL means "cells 0..B-1"; otherwise cell b of the minibatch is row rows[b]
 * (the AnnDataLoader row gather, dataloaders/_ann_dataloader.py, fused into the kernels).
 * col_off selects the group's own genes inside an outer-joined matrix
 * (x[:, groups_var_indices[g]], module/spVIPESmodule.py:428-430). */
typedef struct spv_counts {
  const void* X;
  int64_t ld;       /* elements per row                    */
  const int32_t* rows;
  int32_t col_off;
  int32_t dtype;    /* SPV_COUNT_F32 | SPV_COUNT_U16       */
} spv_counts;

/* fp32 [R][C] (+ one optional extra column: a per-row vector, or the constant 1) -> bf16 hi (and
 * lo = bf16(x - hi) when dst_lo != NULL) image dst[Rp][ld_dst], written at column dst_col_off
 * over `cslot` columns and `Rp` rows, zero filled outside the source. */
int spv_pack_bf16(const float* src, int64_t ld_src, int32_t R, int32_t C,
                  const float* extra_col, int32_t extra_one,
                  uint16_t* dst_hi, uint16_t* dst_lo, int64_t ld_dst, int32_t dst_col_off,
                  int32_t Rp, int32_t cslot, void* stream);

/* The encoder's first layer in the nsplit == 1 mode keeps its 16-bit operand images as IEEE f16 words (11 significant bits instead
 * of bf16's 8, same MFMA rate; a 10 000 .. 30 000-term contraction with bf16 operands left the latent means 5e-3 off the fp32
 * reference): the weight image W1_hi = f16(W * spv_fc1_w_scale()) (a power of two, undone by the forward epilogue), the resident
 * log1p image of spv_prepare_log1p, and the dh image of the weight gradient scaled per step by a power of two (spv_enc_fc1_bwd_prep).
 * spv_pack_f16: fp32 [R][C] * scale -> f16 image dst[Rp][ld_dst] over `cslot` columns, zero filled outside the source, saturating
 * (cslot and ld_dst multiples of 8, dst 16-byte aligned). */
float spv_fc1_w_scale(void);
int spv_pack_f16(const float* src, int64_t ld_src, int32_t R, int32_t C, uint16_t* dst, int64_t ld_dst, int32_t Rp, int32_t cslot,
                 float scale, void* stream);

/* 1 when spv_enc_fc1_fwd will take its LDS-DMA kernel for these shapes (resident bf16 log1p image, N1 = 256, nsplit 1, operand rows
 * zero padded to multiples of 64 genes).  The caller needs to know because that path wants `slabs` sized
 * [splits][round_up(B, 128)][N1] fp32 (it keeps the partial sums in MFMA accumulator-tile order) and splits chosen as
 * clamp(256 / ceil(B / 128), 1, 16) with at least four 64-gene steps per split. */
int spv_enc_fc1_fwd_uses_dma(int32_t B, int32_t G, int32_t N1, int32_t nsplit, int32_t have_xb, int64_t ldw, int64_t ld_xb);

/* A1 + first layer of both encoders of a group (module/spVIPESmodule.py:428-435,
 * nn/networks.py:119):  h1 = relu(log1p(X[rows]) @ W1^T + b1) for the concatenated
 * [private ; shared] fc1 (N1 = 2 * n_hidden output columns), library = log(sum_g log1p(x)).
 *   W1_hi/lo : [N1p][ldw] (rows = output units, K = genes contiguous, ldw % 32 == 0): nsplit 3: bf16 hi / lo of W;
 *              nsplit 1: W1_hi = f16(W * spv_fc1_w_scale()) from spv_pack_f16 / spv_adam_step_images, W1_lo unused
 *   bias     : fp32 [N1], or the two encoders' biases in place: bias [n_first] and bias2 [N1 - n_first]
 *              (bias2 == NULL: one vector)
 *   slabs    : fp32 workspace [splits][B][N1];  rowsum_ws: fp32 [splits][B]
 *   h1       : fp32 [B][N1];  library: fp32 [B]
 *   xb_all, library_all : optional (nsplit 1 only): f16(log1p(x)) of the WHOLE resident count matrix [n_cells][ld_xb] (zero
 *              padded to ld_xb >= round_up(G, 128)) and log(sum_g log1p(x)) per cell, both from spv_prepare_log1p; the GEMM
 *              then gathers plain bf16 rows through x->rows instead of decoding counts, and library is a table lookup
 *   cov, cov_idx : optional (both or neither): one-hot batch covariates appended to the layer's input (nn/networks.py:105-119,
 *              module/spVIPESmodule.py:133,440-445) as what they amount to: cov fp32 [n_batch][N1] = the weight columns of the
 *              covariates, transposed, both encoders side by side; cov_idx int32 [B] = the batch code of every minibatch cell;
 *              cov[cov_idx[b]][:] is added before the ReLU */
int spv_enc_fc1_fwd(const spv_counts* x, int32_t B, int32_t G,
                    const uint16_t* W1_hi, const uint16_t* W1_lo, int64_t ldw, int32_t N1,
                    const float* bias, const float* bias2, int32_t n_first, int32_t nsplit, int32_t splits,
                    float* slabs, float* rowsum_ws, float* h1, float* library,
                    const uint16_t* xb_all, int64_t ld_xb, const float* library_all,
                    const float* cov, const int32_t* cov_idx, void* stream);

/* Once per resident count matrix: xb[c][g] = f16(log1p(X[c][g])) (zero for g >= G, row pitch ld_xb) and
 * library[c] = log(sum_g log1p(X[c][g])) (module/spVIPESmodule.py:428-435 evaluated for every cell of the data set).
 * x->rows is ignored.                                                                                               */
int spv_prepare_log1p(const spv_counts* x, int32_t n_cells, int32_t G, uint16_t* xb, int64_t ld_xb, float* library, void* stream);
/* The same for the "fp32" (nsplit 3) mode: bf16 hi words of log1p(x) and lo words (bf16(x - hi)), interleaved in blocks of 32 genes --
 * words [64 b, 64 b + 32) of a row are the hi words of genes 32 b .. 32 b + 31, words [64 b + 32, 64 b + 64) their lo words (one
 * 128-byte line = what one K tile of the LDS-DMA fc1 kernel takes from a row); ld_xb (words per row) a multiple of 128,
 * >= 2 round_up(G, 64) for that kernel, zero padded.  spv_enc_fc1_fwd takes it as xb_all with nsplit 3 and that ld_xb. */
int spv_prepare_log1p_split(const spv_counts* x, int32_t n_cells, int32_t G, uint16_t* xb, int64_t ld_xb, float* library, void* stream);

/* fc1 weight gradient (autograd of nn/networks.py:119):  dW1[N1][G] = dh^T @ log1p(X[rows]).
 *   dh_hi/lo : [round_up(B,64)][ld_dh] (ld_dh >= round_up(N1,128), zero padded) from spv_enc_fc1_bwd_prep: nsplit 3 bf16 hi / lo,
 *              nsplit 1 f16(dpre * scale) with dh_scale = that call's scale_ws ({scale, 1 / scale}: the result is multiplied by [1])
 *   xb       : optional: nsplit 1: the resident f16 log1p image of spv_prepare_log1p ([n_cells][ld_xb], gathered through x->rows);
 *              nsplit 3: the resident split image of spv_prepare_log1p_split, where spv_enc_fc1_wgrad_split_uses_dma says so
 *              (B <= 4096-ish: the row-index table of the minibatch sits in LDS); NULL = decode the counts here
 *   dW2      : optional second destination: rows >= rows_first go to dW2[row - rows_first] (the shared
 *              encoder's weight gradient, stored apart from the private encoder's)  */
int spv_enc_fc1_wgrad_split_uses_dma(int32_t B, int32_t G, int32_t N1, int64_t ld_dh, int64_t ld_xb);   /* nsplit 3: may xb be passed? */
int spv_enc_fc1_wgrad(const spv_counts* x, int32_t B, int32_t G,
                      const uint16_t* dh_hi, const uint16_t* dh_lo, int64_t ld_dh, int32_t N1,
                      int32_t nsplit, float* dW, float* dW2, int32_t rows_first, int64_t ldc,
                      const uint16_t* xb, int64_t ld_xb, const float* dh_scale, void* stream);

/* Backward of fc1's ReLU + bias (nn/networks.py:119): dpre = dh1 * (h1 > 0) packed as the 16-bit image
 * spv_enc_fc1_wgrad consumes ([Bp][ld_img], zero padded), and the bias gradients
 * db[col] = sum_b dpre[b][col] (cols >= n_first to db2[col - n_first] when db2 != NULL).
 *   img_lo != NULL ("fp32" mode): bf16 hi / lo images;  img_lo == NULL: img_hi = f16(dpre * scale), scale = the power of two that
 *   brings max |dpre| of this call into [4096, 8192) (a pure function of dpre: no state across steps, no atomics)
 *   part     : fp32 workspace [Bp / 16][N1]
 *   scale_ws : fp32 [2 + Bp / 16] (required when img_lo == NULL): [0] = scale, [1] = 1 / scale, rest = per-block maxima */
int spv_enc_fc1_bwd_prep(const float* dh1, const float* h1, int32_t B, int32_t N1, uint16_t* img_hi, uint16_t* img_lo,
                         int64_t ld_img, int32_t Bp, float* part, float* db, float* db2, int32_t n_first, float* scale_ws, void* stream);

/* Grouped forms of the three fc1 entry points above: one argument block per group (same meaning as the parameters of
 * spv_enc_fc1_fwd / spv_enc_fc1_bwd_prep + spv_enc_fc1_wgrad).  Pairs of groups whose shapes take the LDS-DMA kernels run as ONE launch
 * per kernel (both groups' tiles in one grid: the 2 x 256 one-per-CU workgroups of two separate launches can only run one after the
 * other anyway, and a single stream needs no fork / join); anything else falls back to the per-group entry points in order. */
typedef struct spv_fc1_fwd_args {
  const spv_counts* x; int32_t B, G;
  const uint16_t* W1_hi; const uint16_t* W1_lo; int64_t ldw; int32_t N1;
  const float* bias; const float* bias2; int32_t n_first, nsplit, splits;
  float* slabs; float* rowsum_ws; float* h1; float* library;
  const uint16_t* xb_all; int64_t ld_xb; const float* library_all;
  const float* cov; const int32_t* cov_idx;   /* optional batch covariates (see spv_enc_fc1_fwd) */
} spv_fc1_fwd_args;
int spv_enc_fc1_fwd_grouped(const spv_fc1_fwd_args* groups, int32_t n_groups, void* stream);

typedef struct spv_fc1_bwd_args {
  const float* dh1; const float* h1; const spv_counts* x;
  int32_t B, G, N1, n_first, nsplit, Bp;
  uint16_t* dh_hi; uint16_t* dh_lo; int64_t ld_dh;      /* workspace: packed 16-bit image of relu'(h1) * dh1, [Bp][ld_dh] */
  float* part;                                          /* workspace: [Bp / 16][N1] column partial sums */
  float* db; float* db2; float* dW; float* dW2; int64_t ldc;
  const uint16_t* xb; int64_t ld_xb;
  float* scale_ws;                                      /* workspace: [2 + Bp / 16] (nsplit 1: scale record of the f16 dh image) */
} spv_fc1_bwd_args;
int spv_enc_fc1_bwd_grouped(const spv_fc1_bwd_args* groups, int32_t n_groups, void* stream);

/* "Accumulator-tile" order of the [cells][genes] arrays exchanged between the decoder kernels
 * (mixing logits, dL, tP, tS):  T[cell/32][gene/32][qq][lane][j], lane = cell%32 + 32 h,
 * gene%32 = 8 qq + 4 h + j, qq, j = 0..3 -- each 32x32 tile is stored as the MFMA accumulator that
 * produced it (genes on MFMA rows, register 4 qq + j), register-group major, so one wave
 * instruction moves 512 contiguous bytes.  n_gene_tiles = Gp / 32. */

/* 1 when spv_gemm_bf16 will take its LDS-DMA kernel (spvipes_amd/csrc/spv_dec_gemm.h) for these arguments: nsplit 1, a tile-ordered
 * A operand, 32 < N <= 320, ldb == 320.  It works on 128-row x 320-column workgroup tiles, so the caller should pick `splits`
 * with ceil(M / 128) * splits close to (a multiple of) the 256 CUs, and the tiled array must cover round_up(M, 128) rows /
 * round_up(K, 64) contraction indices (the decoder's Bp / Gp paddings do). */
int spv_gemm_bf16_uses_dma(int32_t a_kmajor, int32_t M, int32_t N, int32_t K, int32_t nsplit, int32_t a_tiles, int64_t ldb);

/* Plain bf16 MFMA GEMM, fp32 out:  C[M][N] (+)= sum_k A(m,k) B(k,n).
 *   a_kmajor == 0: A is mem[m][k] (k contiguous); a_kmajor == 1: A is mem[k][m];
 *   a_tiles > 0: A is a tiled [cells][genes] array (above) with a_tiles gene tiles per cell tile;
 *                a_kmajor then says whether K runs over cells (1) or over genes (0); lda is unused.
 *   B is always k-major: mem[k][n].   Operands zero padded to tile multiples (64 x 320 / 128 x 32,
 *   K to a multiple of 64).
 *   splits > 1: C is a stack of `splits` fp32 slabs (slab_stride elements apart), one per K range. */
int spv_gemm_bf16(int32_t a_kmajor, const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda,
                  const uint16_t* B_hi, const uint16_t* B_lo, int64_t ldb,
                  float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                  int32_t nsplit, int32_t splits, int64_t slab_stride, int32_t a_tiles, void* stream);

/* The same GEMM with its split-K slabs summed INSIDE the launch (only where spv_gemm_bf16_uses_dma() says 1; else SPV_ERR_UNSUPPORTED):
 * every workgroup stores its slab tile, releases it (agent scope) and takes a ticket on its row tile's counter; the workgroup that draws
 * the last ticket acquires, adds the `splits` slabs of the tile IN SLAB ORDER (for splits <= 8 what spv_reduce_slabs produces, bit for bit), scales
 * by *alpha and writes columns [0, n0) to dst0 and columns [c1, c1 + n1) to dst1 -- the separate reduction launch and its re-read of the
 * slabs from HBM are gone (the decoder backward's d A_m: 2 x 21 MB per step at B 4096, on the critical chain).
 * counters: uint32 [ceil(M / 128)], ZERO when the call is made; the call leaves them zero (the last arriver resets its counter). */
typedef struct spv_gemm_fixup {
  uint32_t* counters;
  const float* alpha;                          /* device scalar, or NULL = 1                        */
  float* dst0; int64_t ld0; int32_t n0;        /* columns [0, n0) of the summed C                    */
  float* dst1; int64_t ld1; int32_t c1, n1;    /* columns [c1, c1 + n1) (dst1 == NULL: none)         */
} spv_gemm_fixup;
int spv_gemm_bf16_fix(int32_t a_kmajor, const uint16_t* A_hi, const uint16_t* A_lo, int64_t lda,
                      const uint16_t* B_hi, const uint16_t* B_lo, int64_t ldb,
                      float* C, int64_t ldc, int32_t M, int32_t N, int32_t K,
                      int32_t nsplit, int32_t splits, int64_t slab_stride, int32_t a_tiles, const spv_gemm_fixup* fix, void* stream);

/* spv_gemm_bf16 for the two groups of a step in ONE grid (group = blockIdx.y) where both calls take the bf16 LDS-DMA kernel in the same
 * direction; otherwise, and for an odd group left over, the per-group entry point in order.  Same arguments, same slabs. */
typedef struct spv_gemm_args {
  int32_t a_kmajor; int32_t pad0_;
  const uint16_t* A_hi; const uint16_t* A_lo; int64_t lda;
  const uint16_t* B_hi; const uint16_t* B_lo; int64_t ldb;
  float* C; int64_t ldc; int32_t M, N, K, nsplit, splits, a_tiles; int64_t slab_stride;
} spv_gemm_args;
int spv_gemm_bf16_grouped(const spv_gemm_args* groups, int32_t n_groups, void* stream);

/* Weight gradients of both rate heads' regressors (BatchNorm folded) in one streaming pass, bf16 mode:
 *   slabP[split][g][0..15] = sum_{cells of the split} tP(cell, g) * Aps[cell][0..15]
 *   slabS[split][g][0..31] = sum_{cells of the split} tS(cell, g) * Aps[cell][16..47]        (backward of nn/networks.py:314-320)
 * tP / tS: bf16 [Bp][Gp] in accumulator-tile order (a_tiles = Gp / 32 >= round_up(G, 256) / 32), already softmax-corrected
 * (spv_dec_softmax_bwd); Aps: bf16 [Bp][48] latent operand image; Bp % 64 == 0.  Sum the slabs with spv_reduce_slabs. */
int spv_dec_heads_wgrad(const uint16_t* tP, const uint16_t* tS, int32_t a_tiles, const uint16_t* Aps, int32_t G, int32_t Bp,
                        int32_t splits, float* slabP, float* slabS, void* stream);

/* Mixing logits of the decoder (nn/networks.py:322-325 with the bias folded into a ones column):
 *   out(b, g) = sum_k Am[b][k] * Wm[g][k],  Am bf16 [Bp][K], Wm bf16 [Gp][K], K % 32 == 0,
 *   Bp, Gp multiples of 128; out is f16 (out_f32 == 0) or f32 in accumulator-tile order. */
int spv_dec_logits(const uint16_t* Am_hi, const uint16_t* Am_lo, const uint16_t* Wm_hi, const uint16_t* Wm_lo,
                   int32_t K, int32_t Bp, int32_t Gp, int32_t nsplit, void* out, int32_t out_f32, void* stream);

/* Decoder + NB-mixture likelihood (nn/networks.py:314-325, module/spVIPESmodule.py:758-759,
 * :817-824).  Field meanings are documented in spvipes_amd/csrc/spv_decoder.h (DecParams has
 * exactly this layout). */
typedef struct spv_dec_params {
  const void* X; int64_t ldx; const int32_t* rows; int32_t col_off; int32_t count_is_u16;
  int32_t B, G, Bp, Gp;
  const void* logits; int32_t n_gene_tiles; int32_t logits_f32; /* tiled mixing logits, f16 | f32 (see below) */
  const uint16_t* Wps_hi; const uint16_t* Wps_lo;
  const uint16_t* Aps_hi; const uint16_t* Aps_lo;
  const void* gene_tab;      /* float4 [Gp]              */
  const void* cnt_tab;       /* float2 [SPV_NB_CMAX][Gp] */
  const float* a_p; const float* a_s; const float* lse_p; const float* lse_s;
  const float* w_row;
  int32_t gene_splits; int32_t genes_per_split;
  float* part_max_p; float* part_sum_p; float* part_max_s; float* part_sum_s;
  float* rec_part; float* tp_part; float* ts_part;
  float* dtheta_part;
  void* dL; void* tP; void* tS; int32_t grads_f32;              /* tiled like logits; bf16 [Bp][Gp], or (grads_f32) a bf16 hi plane
                                                                  followed by a bf16 lo plane, [2][Bp][Gp]: value = hi + lo */
  int32_t nb_splits; int32_t nb_genes_per_split;               /* gene splits of spv_dec_nb_fwd: multiple of 32, <= SPV_NB_GSPL_MAX */
  int32_t nb_cell_tiles;   /* 64-cell tiles one likelihood workgroup walks (0 = 1); dtheta_part then holds ceil(Bp / 64 / nb_cell_tiles) rows */
} spv_dec_params;

/* theta = exp(px_r) and the per-(count, gene) lgamma / digamma table (module/spVIPESmodule.py:758
 * and the lgamma terms of scvi-tools' log_mixture_nb). */
int spv_dec_tables(const float* px_r, int32_t G, int32_t Gp, void* gene_tab, void* cnt_tab, void* stream);

/* softmax denominators of both rate heads; writes lse_p/lse_s and a_k = library - lse_k
 * (the cast away of const on those four pointers is deliberate: they are this call's outputs). */
int spv_dec_lse(const spv_dec_params* p, const float* library, void* stream);

/* rec_part/tp_part/ts_part [nb_splits][Bp], dtheta_part [Bp/64][Gp]; when train != 0 also the
 * per-element gradients dL, tP, tS (accumulator-tile order; bf16, or bf16 hi + lo planes when grads_f32). */
int spv_dec_nb_fwd(const spv_dec_params* p, int32_t train, void* stream);

/* The decoder's per-group launches for the two groups of a step in ONE grid each (group = blockIdx.z; csrc/spv_decoder.h *_pair_kernel):
 * workgroups outside a group's own grid extent leave at once, the others run the single-group kernel body unchanged -- the results are
 * bit-identical to the per-group calls.  A pair goes out as one grid where both groups take the same kernel variant (count storage,
 * logits and gradient word types; bf16 LDS-DMA logits GEMM; bf16 words for the one-pass backward); otherwise, and for an odd group left
 * over, the per-group entry points run in order.  Each entry point reads the fields of spv_dec_group its per-group twin takes as arguments:
 *   spv_dec_tables_grouped   px_r  (writes p.gene_tab / p.cnt_tab)              spv_dec_logits_grouped   Am_*, Wm_*, K, nsplit (writes p.logits)
 *   spv_dec_lse_grouped      library                                            spv_dec_nb_fwd_grouped   --
 *   spv_dec_heads_bwd_grouped  Tp, Ts, dz_part, dw_part
 * At 128 cells per minibatch a likelihood launch is ~15 us of fixed cost: one grid instead of two per kernel is what shortens the step. */
typedef struct spv_dec_group {
  spv_dec_params p;
  const float* library;
  const float* px_r;
  const uint16_t* Am_hi; const uint16_t* Am_lo; const uint16_t* Wm_hi; const uint16_t* Wm_lo; int32_t K, nsplit;
  const float* Tp; const float* Ts; float* dz_part; float* dw_part;
} spv_dec_group;
int spv_dec_tables_grouped(const spv_dec_group* groups, int32_t n_groups, void* stream);
int spv_dec_logits_grouped(const spv_dec_group* groups, int32_t n_groups, void* stream);
int spv_dec_lse_grouped(const spv_dec_group* groups, int32_t n_groups, void* stream);
int spv_dec_nb_fwd_grouped(const spv_dec_group* groups, int32_t n_groups, int32_t train, void* stream);
int spv_dec_heads_bwd_grouped(const spv_dec_group* groups, int32_t n_groups, void* stream);

/* Materialises the decoder outputs the reference's generative() returns (module/spVIPESmodule.py:751-768,
 * nn/networks.py:314-325) for one group: px_scale_k = softmax_G(BN(z_k W_k^T)), px_rate_k = exp(library) * px_scale_k
 * (k = private, shared) and the mixing logits, as row-major fp32 [B][ld] (ld >= G).  Off the hot path: the fused
 * likelihood (spv_dec_nb_fwd) never stores them.  Needs spv_dec_logits and spv_dec_lse to have run on `p`. */
int spv_dec_materialize(const spv_dec_params* p, float* scale_private, float* scale_shared, float* rate_private, float* rate_shared,
                        float* mixing_logits, int64_t ld, void* stream);

/* The latent gradient of the two rate heads alone: what spv_dec_softmax_bwd computes into dz_part, WITHOUT the in-place correction of
 * tP / tS (a read-only pass: bf16 gradient arrays only).  Lets the caller put the write-back pass, which only the regressor
 * weight-gradient GEMMs wait for, beside the backward pass's critical chain instead of on it. */
int spv_dec_dz(const spv_dec_params* p, const float* Tp, const float* Ts, float* dz_part, void* stream);

/* in place: tP <- tP - softmax_p * Tp[b],  tS <- tS - softmax_s * Ts[b].
 * dz_part (optional, bf16 gradient arrays only): fp32 [gene_splits][Bp][48]; per gene split the gradient reaching the latents
 * through the two rate heads, columns 0..15 = sum_g tP[b][g] * W'_p[g][.], columns 16..47 = sum_g tS[b][g] * W'_s[g][.]
 * (sum the slabs with spv_reduce_slabs): replaces the two [B,G] x [G,K] GEMMs over tP and tS. */
int spv_dec_softmax_bwd(const spv_dec_params* p, const float* Tp, const float* Ts, float* dz_part, void* stream);
/* ONE read-only pass over t_P / t_S (bf16 words, or the hi / lo planes of split-bf16 words when p->grads_f32) for everything the backward pass needs from them (nn/networks.py:314-320 through
 * autograd): the latent gradient of the two rate heads (dz_part, as above) and the two regressor weight gradients
 * d [W'_p | c_p] = t'_P^T [z_p | 1], d [W'_s | c_s] = t'_S^T [z_s | 1] as one partial slab per 128-cell workgroup row:
 * dw_part [Bp / 128][Gp][48], a row = [d W'_p | d c_p | 0.. (16 columns) | d W'_s | d c_s | 0.. (32 columns)] like the regressor operand
 * image (sum the first G rows with spv_reduce_slabs, columns 0..15 and 16..47).  t_P / t_S stay uncorrected -- nothing reads
 * them afterwards: replaces spv_dec_softmax_bwd + spv_dec_heads_wgrad (336 MB less [B, G] traffic per group at B 4096 x G 10 000). */
int spv_dec_heads_bwd(const spv_dec_params* p, const float* Tp, const float* Ts, float* dz_part, float* dw_part, void* stream);

/* One 16-bit operand image kept in step with a parameter by spv_adam_step_images: the `count` fp32 values at flat offset `begin`
 * (a multiple of 4) are a row-major matrix with `cols` columns; element (r, c) is also written to
 * dst[(r + row_off) * ld + c + col_off], rounded to bf16 (fmt SPV_IMAGE_BF16) or as f16(value * scale) (fmt SPV_IMAGE_F16: the fc1
 * weight images, scale = spv_fc1_w_scale()).  A vector that forms one column of an image is cols = 1. */
#define SPV_ADAM_MAX_IMAGES 16
#define SPV_IMAGE_BF16 0
#define SPV_IMAGE_F16 1
typedef struct spv_adam_image {
  int64_t begin, count;
  int32_t cols, row_off, col_off, fmt;
  int64_t ld;
  uint16_t* dst;
  float scale; int32_t _pad;
} spv_adam_image;

/* spv_adam_step that also rewrites the bf16 images of the matrices it updates (the packed operands spv_pack_bf16 would otherwise
 * rebuild at the start of the next step: fc1 weights, [W_m | b_m]).  Image padding (rows / columns outside the matrix) is not
 * touched: the image must have been produced by spv_pack_bf16 once.  n_images may be 0. */
int spv_adam_step_images(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2, float eps,
                         float weight_decay, float bc1, float bc2, float grad_scale, const spv_adam_image* images, int32_t n_images,
                         int64_t* step_counter, const int64_t* t_dev, double beta1_d, double beta2_d, void* stream);
/* step_counter (nullable): device int64 incremented by one per call (the counter spv_randn / the dropout masks are keyed by).
 * t_dev (nullable): device int64 = optimiser steps ALREADY taken; when given, bc1 / bc2 are ignored and the bias corrections
 * 1 - beta^(t_dev + 1) are formed on the device from beta1_d / beta2_d (double, as torch.optim.Adam forms them on the host): no
 * argument of the launch changes from step to step, so it can be part of the step's captured graph.  The kernel does not write
 * t_dev: advance it with spv_counter_bump after the launch. */

/* Up to SPV_MAXP gathers / copies of 4-byte words in one launch: dst[i] = src[idx ? idx[i] : i], i < count (the minibatch's label
 * gathers, labels[rows] of every group -- the reference's loader does this on the host: data/_multi_datasplitter.py:65-98 -- and the
 * copies of the row indices into the buffers a captured graph reads). */
typedef struct spv_gather_prob { const uint32_t* src; const int32_t* idx; uint32_t* dst; int64_t count; } spv_gather_prob;
int spv_gather_u32(const spv_gather_prob* probs, int32_t nprob, void* stream);

/* n standard-normal draws (the reparameterisation noise of a step: nn/networks.py:128-134, spVIPESmodule.py:346-349 sample with
 * torch's generator) from a counter-based generator: Philox 4x32-10 on (element index / 4, *counter) keyed by `key`, Box-Muller.
 * `counter` (nullable = 0) is read on the device when the kernel runs -- pass the step counter spv_adam_step_images increments and
 * a captured graph draws fresh noise at every replay without any host-side generator state. */
int spv_randn(float* out, int64_t n, const int64_t* counter, uint64_t key, void* stream);
/* *counter += 1 in stream order.  Forward passes that no optimiser step follows (the reference draws fresh torch.randn noise for
 * every batch of get_latent_representation, model/spvipes.py:536-538, nn/networks.py:128-134) advance the counter with this, so
 * that consecutive inference / validation batches do not share one noise matrix. */
int spv_counter_bump(int64_t* counter, void* stream);

/* One Adam step (torch.optim.Adam semantics, L2 weight decay folded into the gradient) over a
 * flat fp32 parameter buffer; grad_scale multiplies g first (1/world for data-parallel means).
 * bc1 = 1 - beta1^t, bc2 = 1 - beta2^t.  Replaces scvi TrainingPlan's optimiser step
 * (constructed at model/base/training_mixin.py:111). */
int spv_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, float bc1, float bc2, float grad_scale,
                  void* stream);

/* ---------------------------------------------------------------------------------------------
 * Small dense primitives for the [B, <=256] layers (encoder tails nn/networks.py:120-129, decoder
 * trunk nn/networks.py:322-323) and their backward.  fp32, batched: one launch serves up to
 * SPV_MAXP independent problems (2 groups x {private, shared} x {mu, logvar} heads).  All arrays row-major.
 * ------------------------------------------------------------------------------------------- */
#define SPV_MAXP 8

typedef struct spv_linear_prob {
  const float* X; int64_t ldx;     /* [B][K]                                     */
  const float* W;                  /* [N][K] (torch Linear weight)               */
  const float* bias;               /* [N] or NULL                                */
  float* Y; int64_t ldy;           /* [B][N] forward output (read again by the backward masks) */
  const float* dY; int64_t lddy;   /* backward: upstream gradient [B][N]         */
  float* dX; int64_t lddx;         /* backward: [B][K]                           */
  float* dW; float* db;            /* backward: [N][K], [N] (db may be NULL)     */
  int32_t N, K;
  const float* keep;               /* forward, drop_p > 0 only: optional [B][N] keep-mask (1 = keep, row pitch N) used INSTEAD of the
                                      counter-based draw -- the dropout mask of nn/networks.py:121 injected by a caller (parity tests) */
  const float* W2; int32_t n_w2;   /* dgrad only, optional: rows n >= n_w2 of the [N][K] weight are W2[n - n_w2] (two layers that read the
                                      same input -- the mu / logvar heads, nn/networks.py:123-124 -- back-propagate into it in ONE launch:
                                      dY = their gradients side by side [B][N], N = the two widths together); W2 == NULL: one matrix */
  /* forward only, optional: Y also as bf16 hi (and lo: split-bf16) words into columns 0..N-1 of a packed operand image [img_rows][ld_img]
   * (the decoder's mixing-logits operand); rows B..img_rows-1 of those columns are zero filled.  img_hi == NULL: no image */
  uint16_t* img_hi; uint16_t* img_lo; int64_t ld_img; int32_t img_rows;
} spv_linear_prob;
typedef struct spv_linear_batch {
  spv_linear_prob p[SPV_MAXP];
  int32_t nprob, B;
  int32_t relu;       /* forward: Y = relu(.); backward: dY masked with (Y > 0)                      */
  float drop_p;       /* inverted dropout after the relu (0 = none); backward rescales where Y > 0   */
  uint64_t seed;      /* counter-based mask: (seed + *seed_ptr, problem, element)                    */
  const uint64_t* seed_ptr; /* optional device-resident addend (lets a captured hipGraph draw fresh masks) */
  int32_t accumulate; /* dgrad: dX += instead of dX =                                                */
} spv_linear_batch;
int spv_linear_fwd(const spv_linear_batch* a, void* stream);    /* Y = dropout(relu(X W^T + b))      */
int spv_linear_dgrad(const spv_linear_batch* a, void* stream);  /* dX (+)= mask(dY) W                */
/* dW = mask(dY)^T X, db = colsum(mask(dY)); wpart: fp32 workspace of >= nprob * 16 * Nmax * (Kmax + 1)
 * elements (the batch is reduced in 16 slices, then summed in slice order)                          */
int spv_linear_wgrad(const spv_linear_batch* a, float* wpart, int64_t wpart_elems, void* stream);

typedef struct spv_bn_prob {
  const float* X; int64_t ldx;     /* [B][N], N <= 256                                              */
  float* Y; int64_t ldy;
  const float* gamma; const float* beta;
  float* running_mean; float* running_var;   /* updated in training (momentum, unbiased variance)   */
  float* stats;                    /* [N][2] saved (mean, 1/sqrt(var+eps))                          */
  float* part;                     /* workspace [ceil(B/64)][N][2]                                  */
  const float* dY; int64_t lddy;   /* backward                                                     */
  float* dX; int64_t lddx;
  float* dgamma; float* dbeta;
  int32_t N;
  /* optional forward by-product: Y also as a packed bf16 operand image (hi, and lo = bf16(Y - hi) when img_lo != NULL)
   * img[b][c], b < img_rows (rows >= B are written as zeros), c < N, row pitch ld_img                              */
  uint16_t* img_hi; uint16_t* img_lo; int64_t ld_img; int32_t img_rows;
} spv_bn_prob;
typedef struct spv_bn_batch {
  spv_bn_prob p[SPV_MAXP];
  int32_t nprob, B, training, relu;  /* relu: Y = relu(bn(X)) and the backward masks dY with (Y > 0) */
  float eps, momentum;
} spv_bn_batch;
int spv_bn_fwd(const spv_bn_batch* a, void* stream);
int spv_bn_bwd(const spv_bn_batch* a, void* stream);

/* post = BatchNorm output [B][2n] = (loc | logvar): scale = exp(logvar/2), log_z = loc + scale*eps,
 * theta = softmax(log_z), kl_b = KL(N(loc,scale) || N(0,1)) summed over n (nn/networks.py:125-129,
 * module/spVIPESmodule.py:841-854).  Backward: d_post from the (nullable) upstream gradients. */
typedef struct spv_sample_prob {
  const float* post; int32_t n;
  const float* eps;
  float* scale; float* logz; float* theta; float* kl;
  const float* g_loc; const float* g_logvar; const float* g_scale; const float* g_logz; const float* g_kl;
  float* d_post;
  int64_t g_ld;   /* row pitch of g_loc / g_logvar (0 = n): they may be column blocks of a wider matrix */
} spv_sample_prob;
typedef struct spv_sample_batch { spv_sample_prob p[SPV_MAXP]; int32_t nprob, B; } spv_sample_batch;
int spv_enc_sample_fwd(const spv_sample_batch* a, void* stream);
int spv_enc_sample_bwd(const spv_sample_batch* a, void* stream);
/* The encoder heads' BatchNorm + sampling as ONE pair of launches per direction (nn/networks.py:123-129 and its autograd):
 * bn->p[2e], bn->p[2e + 1] = the BatchNorm problems of encoder e's mu / logvar heads (N = n <= 32, their outputs Y the two halves of
 * sb->p[e].post [B][2n], in the backward their dY the two halves of sb->p[e].d_post), sb->p[e] the sampling problem.
 *   forward : bn_stats + one kernel that finalises the statistics, normalises, samples and forms the KL (same bits as
 *             spv_bn_fwd + spv_enc_sample_fwd, two launches fewer)
 *   backward: one kernel for d_post and the BatchNorm partial sums + one that finalises d gamma / d beta and applies
 *             (training-mode statistics only; eval mode: spv_enc_sample_bwd + spv_bn_bwd) */
int spv_enc_heads_fwd(const spv_bn_batch* bn, const spv_sample_batch* sb, void* stream);
int spv_enc_heads_bwd(const spv_bn_batch* bn, const spv_sample_batch* sb, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Label-based Product of Experts on device (module/spVIPESmodule.py:583-718 + _poe2 :282-379).
 * ------------------------------------------------------------------------------------------- */
#define SPV_NB_GSPL_MAX 160 /* genes per likelihood split: their regressor weights sit in LDS */
#define SPV_POE_LMAX 1024   /* label codes are integers in [0, SPV_POE_LMAX) */

/* rank-within-label pairing of two minibatches; labels are float32 codes (as the reference carries
 * them); order*, rank*: int32 scratch [B*]; tables: int32 scratch [2][2][SPV_POE_LMAX];
 * partner*: int32 [B*] (-1 = none); mode*: int32 [B*] (0 partner, 1 ones/zeros padding, 2 label
 * absent from the other group); *err != 0 on a label code outside [0, SPV_POE_LMAX). */
int spv_poe_partner(const float* labels0, const float* labels1, int32_t B0, int32_t B1, int32_t* order0, int32_t* order1,
                    int32_t* rank0, int32_t* rank1, int32_t* tables, int32_t* partner0, int32_t* mode0,
                    int32_t* partner1, int32_t* mode1, int32_t* err, void* stream);

/* the ranking half of spv_poe_partner alone (one launch): rank of every cell within its label, the cells of each label in batch
 * order, tables[2][2][SPV_POE_LMAX] = per group (count, start) per label */
int spv_poe_rank(const float* labels0, const float* labels1, int32_t B0, int32_t B1, int32_t* order0, int32_t* order1,
                 int32_t* rank0, int32_t* rank1, int32_t* tables, int32_t* err, void* stream);

typedef struct spv_poe_args {
  const float* stats[2]; int64_t ld[2];   /* shared encoders' (loc | logvar) rows, [B][ld], logvar at column n */
  const int32_t* partner[2]; const int32_t* mode[2];
  const float* eps[2];                     /* [B][n] */
  float* loc[2]; float* logvar[2]; float* scale[2]; float* logz[2]; float* theta[2]; float* kl[2];
  const float* g_loc[2]; const float* g_logvar[2]; const float* g_scale[2]; const float* g_logz[2]; const float* g_kl[2];
  float* d_stats[2];                       /* backward output, same layout as stats (zeroed by the call itself) */
  int32_t B[2]; int32_t n;
  int32_t clamp_scale;                     /* paired / cluster PoE: the draw and the KL use scale.clamp(min = 1e-6)  */
  int32_t lone_passthrough;                /* cluster PoE: mode 2 keeps the cell's own encoder statistics           */
  const float* expert[2]; int64_t ld_expert[2];  /* cluster PoE: plan-weighted experts [B][ld] (loc | logvar) fused in
                                                    place of the encoder statistics; NULL = label / paired PoE      */
  float* d_expert[2];                      /* backward output for expert (zeroed by the call itself)                */
  /* label PoE: when lab[0] != NULL spv_poe_fuse_fwd derives every cell's partner / mode itself from the ranking left
   * by spv_poe_rank (labels, rank within label, cells of each label in batch order, per-label count / start tables) and STORES them
   * through partner / mode (which are then outputs) for the backward pass -- the separate lookup launch of spv_poe_partner is gone.
   * spv_poe_fuse_bwd reads lab[0] only as a flag: non-NULL = partners are one to one (label pairing: one atomic per cell and column);
   * NULL = partners are arg maxima / component ranks that may coincide (paired, cluster PoE): contributions to one partner are first
   * summed inside a 32-cell workgroup */
  const float* lab[2]; const int32_t* order[2]; const int32_t* rank[2]; const int32_t* tables;
} spv_poe_args;
int spv_poe_fuse_fwd(const spv_poe_args* a, void* stream);
int spv_poe_fuse_bwd(const spv_poe_args* a, void* stream);

/* N-group cluster-matched Product of Experts (BASELINE config 4: 3 groups; the N-expert generalisation of
 * _product_of_experts, module/spVIPESmodule.py:573-581, that SURVEY.md 8d prescribes -- throughput only, the reference stops
 * at two groups: data/prepare_adatas.py:94-95).  For cell i of group g with component c: experts = N(0, 1) prior, the cell's
 * own shared-encoder posterior, and for every other group h with cells of component c in its minibatch the COMPONENT MEAN of
 * h's (loc, logvar).  Fixed-order reductions (no atomics).  Up to SPV_POE_MAXG groups, latent dimension <= 32, component codes
 * integral in [0, ncomp), ncomp <= SPV_POE_COMP_CMAX.                                                                   */
#define SPV_POE_MAXG 4
#define SPV_POE_COMP_SEG 16
#define SPV_POE_COMP_CMAX 64
typedef struct spv_poe_comp_args {
  int32_t ngroups, n, ncomp, pad_;
  int32_t B[SPV_POE_MAXG];
  const float* stats[SPV_POE_MAXG]; int64_t ld[SPV_POE_MAXG];   /* (loc | logvar) rows [B][ld], logvar at column n      */
  const float* comp[SPV_POE_MAXG];                               /* component code of every cell [B] (fp32, integral)     */
  const float* eps[SPV_POE_MAXG];                                /* [B][n] standard-normal draws                         */
  float* part;   /* workspace fp32 [ngroups][SPV_POE_COMP_SEG][ncomp][2n + 1]                                              */
  float* mean;   /* workspace fp32 [ngroups][ncomp][2n + 1]: mean loc | mean logvar | count (kept for the backward pass)   */
  float* loc[SPV_POE_MAXG]; float* logvar[SPV_POE_MAXG]; float* scale[SPV_POE_MAXG]; float* logz[SPV_POE_MAXG];
  float* theta[SPV_POE_MAXG]; float* kl[SPV_POE_MAXG];
  const float* g_loc[SPV_POE_MAXG]; const float* g_logvar[SPV_POE_MAXG]; const float* g_scale[SPV_POE_MAXG];
  const float* g_logz[SPV_POE_MAXG]; const float* g_kl[SPV_POE_MAXG];
  float* dpn[SPV_POE_MAXG];       /* backward workspace [B][2n]                                                           */
  float* d_stats[SPV_POE_MAXG];   /* backward output, layout as stats                                                     */
} spv_poe_comp_args;
int spv_poe_comp_fwd(const spv_poe_comp_args* a, void* stream);
int spv_poe_comp_bwd(const spv_poe_comp_args* a, void* stream);

/* Sparse transport plan: CSR of the plan (rows = dataset cells of group 0, int32 indptr / column
 * indices, fp32 values >= 0) and CSR of its transpose (rows = dataset cells of group 1).  Replaces the dense
 * plan[idx_0][:, idx_1] gather of module/spVIPESmodule.py:474-482.                                                    */
typedef struct spv_plan {
  const int32_t* ptr0; const int32_t* ind0; const float* val0; int32_t n0;   /* plan   : n0 rows */
  const int32_t* ptr1; const int32_t* ind1; const float* val1; int32_t n1;   /* plan^T : n1 rows */
} spv_plan;
/* inv0[c] / inv1[c] = position of dataset cell c in the minibatch (idx0 [B0] / idx1 [B1], unique), -1 if absent */
int spv_plan_invmap(const int32_t* idx0, int32_t B0, const int32_t* idx1, int32_t B1, int32_t* inv0, int32_t n0, int32_t* inv1,
                    int32_t n1, void* stream);
/* paired PoE partners (module/spVIPESmodule.py:520-523): partner0[i] = argmax_j block[i][j], partner1[j] = argmax_i
 * block[i][j] of the minibatch block of the plan; first maximum, 0 for an all-zero row / column                       */
/* Cluster PoE experts (module/spVIPESmodule.py:213-229) from the stored plan entries of the minibatch:
 *   expert[g][i] = sum_j w'_ij * stats[g][j]   (loc | logvar),  j = positions of the other minibatch in i's component,
 *   w' = row-normalised plan weights (row sums clamped at 1e-10, kept in rowsum[g] for the backward).
 * comp[g]: component codes as fp32 [B]; both minibatches have B cells.  The backward ADDS into d_stats[g].            */
typedef struct spv_plan_expert_args {
  spv_plan plan;
  const int32_t* idx[2]; const int32_t* inv[2]; const float* comp[2];
  const float* stats[2]; int64_t ld[2];
  float* expert[2]; int64_t ld_expert[2];
  float* rowsum[2];
  const float* d_expert[2]; float* d_stats[2];
  int32_t B, n;
} spv_plan_expert_args;
int spv_plan_expert_fwd(const spv_plan_expert_args* a, void* stream);
int spv_plan_expert_bwd(const spv_plan_expert_args* a, void* stream);
int spv_plan_argmax(const spv_plan* plan, const int32_t* idx0, const int32_t* idx1, const int32_t* inv0, const int32_t* inv1,
                    int32_t B0, int32_t B1, int32_t* partner0, int32_t* partner1, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Decoder preparation: latent slicing (spVIPESmodule.py:733-754) and BatchNorm folding of the two
 * factor regressors (nn/networks.py:314,318) into the packed operand image of the decoder kernels.
 * ------------------------------------------------------------------------------------------- */
typedef struct spv_zsplit_args {
  const float* priv[2]; const float* poe[2];   /* private log_z [B][n_p], PoE log_z [B][n_s] */
  float* zcat[2];                               /* [B][n_p + n_s] = [z_private | z_shared]    */
  const float* d_zcat[2]; float* d_priv[2]; float* d_poe[2];
  int32_t B, n_p, n_s, ngroups;
  /* optional forward by-products (all rows < Bp, zero beyond B): the decoder's packed bf16 operand images
   *   am  [Bp][ld_am]  columns am_col .. am_col + am_cols - 1  <-  [ zcat | 1 | 0 .. ]   (mixture-logit operand tail)
   *   aps [Bp][48]     <-  [ z_private | 1 | 0 ..(16) | z_shared | 1 | 0 ..(32) ]        (rate-regressor operand)     */
  uint16_t* am_hi[2]; uint16_t* am_lo[2]; int64_t ld_am; int32_t am_col; int32_t am_cols;
  uint16_t* aps_hi[2]; uint16_t* aps_lo[2]; int32_t Bp;
} spv_zsplit_args;
int spv_zsplit_fwd(const spv_zsplit_args* a, void* stream);
int spv_zsplit_bwd(const spv_zsplit_args* a, void* stream);

typedef struct spv_fold_prob {
  const float* W; const float* gamma; const float* beta;   /* [G][K], [G], [G]            */
  float* running_mean; float* running_var;                 /* [G]                         */
  const float* zsum; const float* zz;                      /* [K] column sums of z, [K][K] z^T z */
  const float* z; int64_t ldz;                             /* [B][ldz] (backward)         */
  float* stat;                                             /* [G][2] saved (mean, var)    */
  uint16_t* img_hi; uint16_t* img_lo; int64_t ld_img; int32_t col_off; int32_t slot;   /* packed [Gp][ld_img] */
  const float* dWeff; int64_t ld_dw;                       /* backward in: [G][ld_dw], col K = d c */
  float* dW; float* dgamma; float* dbeta;                  /* backward out                */
  float* red_part;                                         /* [ceil(G/256) + 1][K + K*K]  */
  float* dz; int64_t lddz;                                 /* backward out (+=): [B][lddz] */
  int32_t G, Gp, K;
  /* optional: the latent-slicing backward (spVIPESmodule.py:733-754, see spv_zsplit_args) done by the same kernel -- dz then is this
   * problem's column block (first column zcol) of d zcat [B][n_p + n_s], which is only READ, and the finished column c of d zcat goes
   * to d private_log_z / d poe_log_z (out_priv [B][n_p], out_poe [B][n_s]) instead of back into dz: no spv_zsplit_bwd launch.  The two
   * problems of a group (private block, shared block) together cover every column exactly once. */
  int32_t zcol; float* out_priv; float* out_poe; int32_t n_p, n_s;
} spv_fold_prob;
typedef struct spv_fold_batch { spv_fold_prob p[SPV_MAXP]; int32_t nprob, B, training; float eps, momentum; } spv_fold_batch;
int spv_bn_fold_fwd(const spv_fold_batch* a, void* stream);
int spv_bn_fold_bwd(const spv_fold_batch* a, void* stream);   /* fold backward + the z-statistics backward */

/* The decoder's mixing trunk m = relu(BN_256(cat(z) W_a^T + b_a)) (nn/networks.py:322-323; scvi FCLayers: BatchNorm1d eps 1e-3, momentum
 * 0.01) with its training-mode BatchNorm folded the same way: batch mean and variance of a LINEAR map of z follow from zbar and cov(z)
 * (mean_j = w_j . zbar + b_j, var_j = w_j^T C w_j), so the layer is one affine map W'_j = (gamma_j / sqrt(var_j + eps)) w_j,
 * c'_j = beta_j - W'_j . zbar (the Linear's own bias cancels in the batch mean: its gradient is exactly zero, as in the reference) followed
 * by the rectifier: spv_trunk_fold_fwd writes (W', c') for spv_linear_fwd(relu); three launches (statistics, finalise, normalise) and the
 * pre-activation round trip are gone from the forward pass, three (partial sums, finalise, apply) from the backward pass.
 * Backward: from d W' [N][K] and d c' [N] (spv_linear_wgrad of the folded layer, masked by the rectifier) spv_trunk_fold_bwd forms d W_a,
 * d gamma, d beta (d b_a = 0) and (d zbar | d C) [K + K*K], then adds d z[b] += d zbar / B + (d C + d C^T)(z[b] - zbar) / B into dz.
 * Eval mode (training == 0): running statistics instead of batch statistics, no z-statistics term.  N <= 256, K <= 48. */
#define SPV_TRUNK_KMAX 48
typedef struct spv_trunk_prob {
  const float* W; const float* bias; const float* gamma; const float* beta;   /* Linear [N][K], [N]; BatchNorm affine [N], [N]          */
  float* running_mean; float* running_var;                                    /* [N] (updated by the forward call in training mode)    */
  const float* zsum; const float* zz;                                         /* [K] column sums of z, [K][K] z^T z over the batch     */
  float* Wf; float* cf; float* stat;                                          /* forward out: W' [N][K], c' [N], (mean, var) [N][2]    */
  const float* dWf; const float* dcf;                                         /* backward in: d W' [N][K], d c' [N]                    */
  float* dW; float* dbias; float* dgamma; float* dbeta;                       /* backward out: [N][K], [N] (zeros), [N], [N]           */
  float* dred;                                                                /* backward scratch: [ceil(N / 64)][K + K*K] partials of (d zbar | d C) */
  const float* z; int64_t ldz; float* dz; int64_t lddz;                       /* z [B][K] (read), d z [B][K] (accumulated into)        */
  int32_t N, K;
} spv_trunk_prob;
typedef struct spv_trunk_batch { spv_trunk_prob p[2]; int32_t nprob, B, training; float eps, momentum; } spv_trunk_batch;
int spv_trunk_fold_fwd(const spv_trunk_batch* a, void* stream);
int spv_trunk_fold_bwd(const spv_trunk_batch* a, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Deterministic slab reduction (the tail of every split-K GEMM and of the per-wave partials):
 *   dst[r][c] (+)= alpha * escale(c) * sum_s src[s * slab_stride + r * ld_src + col_off + c]
 * up to 8 slabs are added in index order; more than 8 in index order inside four fixed interleaved groups
 * (s = y, y + 4, ...; then ((p0 + p1) + p2) + p3).  rows * cols < 2^31.  alpha: optional device scalar
 * (the upstream gradient of the loss); exp_scale: optional per-column log-scale, multiplies by
 * exp(exp_scale[c]) (d px_r = exp(px_r) * d theta, module/spVIPESmodule.py:758).
 * ------------------------------------------------------------------------------------------- */
#define SPV_MAXR 16
typedef struct spv_reduce_prob {
  const float* src; int64_t slab_stride; int64_t ld_src; int32_t nslabs; int32_t col_off;
  int32_t rows, cols;
  float* dst; int64_t ld_dst; int32_t accumulate; int32_t pad_;
  const float* alpha; const float* exp_scale;
} spv_reduce_prob;
typedef struct spv_reduce_batch { spv_reduce_prob p[SPV_MAXR]; int32_t nprob; } spv_reduce_batch;
int spv_reduce_slabs(const spv_reduce_batch* b, void* stream);

/* Loss assembly (module/spVIPESmodule.py:870-872): rec_sum = sum_g sum_b w[b] * rec[g][b] and
 *   loss = rec_sum + (kl_weight / B) * sum_b sum_i kl[i][b]   (i < nkl <= 4 KL vectors of length B);
 * also writes gkl[b] = kl_weight / B, the gradient of the loss w.r.t. every kl[i][b].
 * kl_weight is read from device memory (one float) so a captured graph sees later updates. */
int spv_loss_assemble(const float* rec0, const float* rec1, const float* w, const float* const* kl, int32_t nkl, int32_t B,
                      const float* kl_weight, float* loss, float* rec_sum, float* gkl, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SPVIPES_HIP_H */
