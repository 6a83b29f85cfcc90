// Probe (dev tool): the NB-mixture likelihood kernel (spv_decoder.h: dec_nb_kernel) alone at the C2 shape, on synthetic
// operands of realistic magnitude, so that kernel variants (-DSPV_NB_OCC=2|3|4, source edits) can be timed on the GPU box
// without the Python stack:
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -DSPV_NB_OCC=3 -o nb_bench nb_bench.hip && ./nb_bench [B G]
// Prints the average launch time, the per-SIMD issue rate implied by the instruction count given on the command line
// (optional 3rd arg, wave-instructions per launch from a --pmc SQ_INSTS_VALU pass) and a checksum of the outputs (so that
// two variants can be compared for equal results).
#include "../../spvipes_amd/csrc/spv_decoder.h"
#include <algorithm>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <random>
#include <vector>

using namespace spv;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

static unsigned short f2bf_host(float f) {
  unsigned u; std::memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1);
  return (unsigned short)(u >> 16);
}
static float bf2f_host(unsigned short b) { unsigned u = (unsigned)b << 16; float f; std::memcpy(&f, &u, 4); return f; }

template <typename T> static T* dalloc(size_t n) { T* p; CK(hipMalloc(&p, n * sizeof(T))); CK(hipMemset(p, 0, n * sizeof(T))); return p; }
template <typename T> static T* upload(const std::vector<T>& v) { T* p = dalloc<T>(v.size()); CK(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice)); return p; }

int main(int argc, char** argv) {
  const int B = argc > 1 ? atoi(argv[1]) : 4096, G = argc > 2 ? atoi(argv[2]) : 10000;
  const double instrs = argc > 3 ? atof(argv[3]) : 0.0;
  const int NCELLS = 20000;  // resident matrix the minibatch rows are gathered from
  const int Bp = (B + 127) / 128 * 128, Gp = (G + 255) / 256 * 256;
  std::mt19937 rng(1);
  std::normal_distribution<float> nrm(0.f, 1.f);
  std::uniform_real_distribution<float> uni(0.f, 1.f);
  // counts: 80 % zeros, else small (geometric-ish), a few large ones beyond the table
  std::vector<unsigned short> X((size_t)NCELLS * G);
  for (auto& x : X) { const float u = uni(rng); x = (u < 0.8f) ? 0 : (unsigned short)(1 + (int)(-3.0f * logf(uni(rng) + 1e-6f))); }
  std::vector<int> rows(B);
  for (auto& r : rows) r = (int)(uni(rng) * (NCELLS - 1));
  std::vector<unsigned short> Whi((size_t)Gp * DEC_KPS), Wlo((size_t)Gp * DEC_KPS), Ahi((size_t)Bp * DEC_KPS), Alo((size_t)Bp * DEC_KPS);
  auto put = [&](std::vector<unsigned short>& hi, std::vector<unsigned short>& lo, size_t i, float v) { hi[i] = f2bf_host(v); lo[i] = f2bf_host(v - bf2f_host(hi[i])); };
  for (int g = 0; g < G; ++g) {
    for (int k = 0; k < 10; ++k) put(Whi, Wlo, (size_t)g * DEC_KPS + k, 0.3f * nrm(rng));
    put(Whi, Wlo, (size_t)g * DEC_KPS + 10, 0.5f * nrm(rng));
    for (int k = 0; k < 25; ++k) put(Whi, Wlo, (size_t)g * DEC_KPS + DEC_KP + k, 0.3f * nrm(rng));
    put(Whi, Wlo, (size_t)g * DEC_KPS + DEC_KP + 25, 0.5f * nrm(rng));
  }
  for (int b = 0; b < B; ++b) {
    for (int k = 0; k < 10; ++k) put(Ahi, Alo, (size_t)b * DEC_KPS + k, nrm(rng));
    put(Ahi, Alo, (size_t)b * DEC_KPS + 10, 1.f);
    for (int k = 0; k < 25; ++k) put(Ahi, Alo, (size_t)b * DEC_KPS + DEC_KP + k, nrm(rng));
    put(Ahi, Alo, (size_t)b * DEC_KPS + DEC_KP + 25, 1.f);
  }
  std::vector<_Float16> logits((size_t)Bp * Gp);
  for (auto& l : logits) l = (_Float16)(1.5f * nrm(rng));
  std::vector<float> pxr(G);
  for (auto& v : pxr) v = nrm(rng);
  std::vector<float> ap(Bp), as(Bp), w(Bp, 0.f);
  for (int b = 0; b < B; ++b) { ap[b] = 6.f - logf((float)G) - 0.5f; as[b] = 6.f - logf((float)G) - 0.5f; w[b] = 1.f / B; }

  DecParams p{};
  p.X = upload(X); p.ldx = G; p.rows = upload(rows); p.col_off = 0; p.count_is_u16 = 1;
  p.B = B; p.G = G; p.Bp = Bp; p.Gp = Gp;
  p.logits = upload(logits); p.n_gene_tiles = Gp / 32; p.logits_f32 = 0;
  p.Wps_hi = upload(Whi); p.Wps_lo = upload(Wlo); p.Aps_hi = upload(Ahi); p.Aps_lo = upload(Alo);
  float* d_pxr = upload(pxr);
  float4* gene_tab = dalloc<float4>(Gp); float2* cnt_tab = dalloc<float2>((size_t)NB_CMAX * Gp);
  hipLaunchKernelGGL(nb_tables_kernel, dim3((Gp + 255) / 256, NB_CMAX), dim3(256), 0, 0, d_pxr, G, Gp, gene_tab, cnt_tab);
  p.gene_tab = gene_tab; p.cnt_tab = cnt_tab;
  p.a_p = upload(ap); p.a_s = upload(as); p.lse_p = p.a_p; p.lse_s = p.a_s; p.w_row = upload(w);
  const int per = Gp < NB_GSPL_MAX ? Gp : NB_GSPL_MAX, nbs = (Gp + per - 1) / per;
  p.nb_splits = nbs; p.nb_genes_per_split = per; p.gene_splits = 1; p.genes_per_split = Gp;
  p.nb_cell_tiles = getenv("NB_CELL_TILES") ? atoi(getenv("NB_CELL_TILES")) : 1;
  p.rec_part = dalloc<float>((size_t)nbs * Bp); p.tp_part = dalloc<float>((size_t)nbs * Bp); p.ts_part = dalloc<float>((size_t)nbs * Bp);
  p.dtheta_part = dalloc<float>((size_t)(Bp / 64) * Gp);
  p.dL = dalloc<unsigned short>((size_t)Bp * Gp); p.tP = dalloc<unsigned short>((size_t)Bp * Gp); p.tS = dalloc<unsigned short>((size_t)Bp * Gp);
  p.grads_f32 = 0;
  CK(hipDeviceSynchronize());

  dim3 grid(((Bp + NB_CELLS_PER_WG - 1) / NB_CELLS_PER_WG + p.nb_cell_tiles - 1) / p.nb_cell_tiles, nbs);
#ifdef SPV_NB_STAMPS
  unsigned long long* d_st = dalloc<unsigned long long>((size_t)grid.x * grid.y * 8);   // set BEFORE the first launch: every launch of this build stamps
  CK(hipMemcpyToSymbol(HIP_SYMBOL(g_nb_stamps), &d_st, sizeof(d_st)));
  CK(hipDeviceSynchronize());
#endif
  auto launch = [&]() { hipLaunchKernelGGL((dec_nb_kernel<true, bf16_t, _Float16, CNT_U16_ALIGNED>), grid, dim3(256), 0, 0, p); };
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 50;
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
#ifdef SPV_NB_STAMPS
  {  // phases of a workgroup's lifetime (wave 0), median over workgroups, in shader cycles
    launch(); CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)grid.x * grid.y * 8);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    const char* names[5] = {"stage W slice + gene table into LDS (to barrier)", "cell fragments, first loads, first table gather", "chunk 0", "chunks 1..n-1", "row sums, barrier, d-theta store"};
    unsigned long long tmin = ~0ull, tmax = 0;
    for (size_t b = 0; b < (size_t)grid.x * grid.y; ++b) { tmin = std::min(tmin, st[b * 8]); tmax = std::max(tmax, st[b * 8 + 5]); }
    printf("  kernel span %.1f k cycles (first workgroup start -> last end)\n", (double)(tmax - tmin) / 1e3);
    for (int ph = 0; ph < 5; ++ph) {
      std::vector<double> d;
      for (size_t b = 0; b < (size_t)grid.x * grid.y; ++b) if (st[b * 8 + ph + 1] > st[b * 8 + ph]) d.push_back((double)(st[b * 8 + ph + 1] - st[b * 8 + ph]));
      std::sort(d.begin(), d.end());
      if (!d.empty()) printf("  phase %d %-52s median %8.0f  p10 %8.0f  p90 %8.0f cycles\n", ph, names[ph], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
    }
    std::vector<double> life;
    for (size_t b = 0; b < (size_t)grid.x * grid.y; ++b) life.push_back((double)(st[b * 8 + 5] - st[b * 8]));
    std::sort(life.begin(), life.end());
    printf("  workgroup lifetime median %.0f cycles; %d workgroups, %d slots at 3 per CU\n", life[life.size() / 2], grid.x * grid.y, 768);
  }
#endif
  // checksum
  std::vector<float> rec((size_t)nbs * Bp), dth((size_t)(Bp / 64) * Gp);
  CK(hipMemcpy(rec.data(), p.rec_part, rec.size() * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(dth.data(), p.dtheta_part, dth.size() * 4, hipMemcpyDeviceToHost));
  std::vector<unsigned short> tS((size_t)Bp * Gp);
  CK(hipMemcpy(tS.data(), p.tS, tS.size() * 2, hipMemcpyDeviceToHost));
  double s1 = 0, s2 = 0, s3 = 0;
  for (float v : rec) s1 += v;
  for (float v : dth) s2 += v;
  for (size_t i = 0; i < tS.size(); i += 7) s3 += bf2f_host(tS[i]);
  printf("dec_nb_kernel<train,bf16,f16,u16-aligned> OCC=%d  B=%d G=%d grid=%dx%d: %.1f us/launch   rec=%.6e dtheta=%.6e tS~=%.6e\n", SPV_NB_OCC, B, G, grid.x,
         grid.y, us, s1, s2, s3);
  if (instrs > 0) printf("  %.1f M wave-instr / 1024 SIMDs / (%.1f us x 2.4 GHz) = %.3f instr/cycle/SIMD\n", instrs / 1e6, us, instrs / 1024.0 / (us * 2400.0));
  return 0;
}
