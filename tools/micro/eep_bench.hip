// Stand-alone timing of the fp32 pair embedder (edge_embed_f32p_kernel vs the tiled edge_embed_kernel<PrecF32>).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w tools/micro/eep_bench.hip -o tools/micro/eep_bench ; eep_bench [N] [B]
#include "../../framedipt_amd/csrc/pair_mlp.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int B = argc > 2 ? atoi(argv[2]) : 8, N = argc > 1 ? atoi(argv[1]) : 300, NB = 22;
  const long P = (long)B * N * N, R = (long)B * N;
  const int n_rel = 2 * N - 1;
  EdgeEmbedArgs a;
  float *pi, *pj, *rt, *dt, *ed, *ca, *vecs, *rm, *w, *z; int32_t* si;
  (void)hipMalloc(&pi, R * 128 * 4); (void)hipMalloc(&pj, R * 128 * 4); (void)hipMalloc(&rt, (size_t)B * n_rel * 128 * 4);
  (void)hipMalloc(&dt, (NB + 1) * 128 * 4); (void)hipMalloc(&ed, NB * 4); (void)hipMalloc(&ca, R * 3 * 4); (void)hipMalloc(&vecs, 4 * 128 * 4);
  (void)hipMalloc(&rm, R * 4); (void)hipMalloc(&si, R * 4); (void)hipMalloc(&w, 2 * 128 * 128 * 4); (void)hipMalloc(&z, P * 128 * 4);
  std::vector<float> hw(2 * 128 * 128), hp(R * 128), hed(NB), hca(R * 3), hrm(R, 1.f), hv(4 * 128, 0.5f); std::vector<int32_t> hsi(R);
  for (size_t k = 0; k < hw.size(); ++k) hw[k] = 0.05f * (float)((int)((k * 2654435761u) >> 24) - 128) / 128.f;
  for (size_t k = 0; k < hp.size(); ++k) hp[k] = (float)((int)((k * 40503u) >> 8 & 255) - 100) / 128.f;
  for (int k = 0; k < NB; ++k) hed[k] = 1e-5f + k * (20.f - 1e-5f) / (NB - 1);
  for (long r = 0; r < R; ++r) { hsi[r] = (int)(r % N); for (int c = 0; c < 3; ++c) hca[r * 3 + c] = 3.8f * (r % N) * (c == 0) * 0.3f + 0.1f * ((r * 7 + c) % 13); }
  (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(pi, hp.data(), hp.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(pj, hp.data(), hp.size() * 4, hipMemcpyHostToDevice); (void)hipMemset(rt, 0, (size_t)B * n_rel * 128 * 4);
  (void)hipMemcpy(dt, hp.data(), (NB + 1) * 128 * 4, hipMemcpyHostToDevice); (void)hipMemcpy(vecs, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
  (void)hipMemcpy(ed, hed.data(), NB * 4, hipMemcpyHostToDevice); (void)hipMemcpy(ca, hca.data(), R * 12, hipMemcpyHostToDevice);
  (void)hipMemcpy(rm, hrm.data(), R * 4, hipMemcpyHostToDevice); (void)hipMemcpy(si, hsi.data(), R * 4, hipMemcpyHostToDevice);
  a.B = B; a.N = N; a.n_rel = n_rel; a.rel_off = N - 1; a.num_bins = NB; a.pi = pi; a.pj = pj; a.rtab = rt; a.dtab = dt; a.edges = ed;
  a.seq_idx = si; a.sc_ca = ca; a.w2 = w; a.w3 = w + 128 * 128; a.b2 = vecs; a.b3 = vecs + 128; a.gamma = vecs + 256; a.beta = vecs + 384;
  a.res_mask = rm; a.z_out = z; a.trace = nullptr; a.wb_img = nullptr; a.bb = nullptr; a.bias_out = nullptr; a.H = 8;
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  std::vector<float> ref(4096), got(4096);
  for (int variant = 0; variant < 2; ++variant) {
    if (variant == 1) setenv("FDIPT_EE_F32_TILED", "1", 1);
    for (int i = 0; i < 2; ++i) if (fd_edge_embed(FDIPT_PREC_F32, 128, a, 0)) { printf("launch failed\n"); return 1; }
    (void)hipEventRecord(t0, 0);
    const int iters = 10;
    for (int i = 0; i < iters; ++i) fd_edge_embed(FDIPT_PREC_F32, 128, a, 0);
    (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
    float ms; (void)hipEventElapsedTime(&ms, t0, t1);
    printf("%s N=%d B=%d: %.1f us/launch, %.1f TFLOP/s (65536 FLOP per pair)\n", variant ? "tiled     " : "persistent", N, B, ms / iters * 1e3,
           P * 65536.0 / (ms / iters) / 1e9);
    (void)hipMemcpy(variant ? ref.data() : got.data(), z + (P / 2) * 128, 4096 * 4, hipMemcpyDeviceToHost);
  }
#ifdef EEP_PROF
  {
    unsigned long long h[2][8];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(eep_prof), sizeof(h));
    const char* nm[7] = {"P0 gather", "P1 layer 2", "P2 relu store", "P3 layer 3", "P4 y store", "P5 LayerNorm", "barrier wait"};
    const double tiles = (double)((P + 31) / 32) / (2.0 * 256) * 12;  // per team of block 0, over the 12 persistent launches
    for (int t = 0; t < 2; ++t)
      for (int k = 0; k < 7; ++k) printf("  team %d %-14s %8.0f cycles per tile\n", t, nm[k], h[t][k] / tiles);
  }
#endif
  double md = 0; for (int k = 0; k < 4096; ++k) md = fmax(md, fabs(ref[k] - got[k]));
  printf("max |persistent - tiled| over 32 rows: %.3g\n", md);
  return 0;
}
