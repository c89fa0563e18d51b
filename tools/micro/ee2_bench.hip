// Stand-alone timing + in-kernel phase profile of edge_embed2_kernel (build with -DEE2_PROF for the profile).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w [-DEE2_PROF] [-DEE2_RESIDENT=k] [-DEE2_EARLY=k] tools/micro/ee2_bench.hip -o ee2_bench
#include "../../framedipt_amd/csrc/edge_embed2.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int B = argc > 2 ? atoi(argv[2]) : 8, N = argc > 1 ? atoi(argv[1]) : 300, NB = 22;
  const long P = (long)B * N * N, R = (long)B * N, Np = (N + 31) / 32 * 32;
  const int n_rel = 2 * N - 1;
  EdgeEmbedArgs a;
  float *pi, *pj, *rt, *dt, *ed, *ca, *vecs, *rm, *bo, *bb; int32_t* si; void *img, *wb; half_t* z;
  (void)hipMalloc(&pi, R * 128 * 4); (void)hipMalloc(&pj, R * 128 * 4); (void)hipMalloc(&rt, (size_t)B * n_rel * 128 * 4);
  (void)hipMalloc(&dt, (NB + 1) * 128 * 4); (void)hipMalloc(&ed, NB * 4); (void)hipMalloc(&ca, R * 3 * 4); (void)hipMalloc(&vecs, 4 * 128 * 4);
  (void)hipMalloc(&rm, R * 4); (void)hipMalloc(&si, R * 4); (void)hipMalloc(&img, fd_ee2_image_bytes()); (void)hipMalloc(&wb, 8192);
  (void)hipMalloc(&z, P * 128 * 2); (void)hipMalloc(&bo, (size_t)B * 8 * Np * Np * 4); (void)hipMalloc(&bb, 64);
  (void)hipMemset(pi, 0, R * 128 * 4); (void)hipMemset(pj, 0, R * 128 * 4); (void)hipMemset(rt, 0, (size_t)B * n_rel * 128 * 4);
  (void)hipMemset(dt, 0, (NB + 1) * 128 * 4); (void)hipMemset(vecs, 0, 4 * 128 * 4); (void)hipMemset(img, 0, fd_ee2_image_bytes());
  (void)hipMemset(wb, 0, 8192); (void)hipMemset(bb, 0, 64);
  std::vector<float> hed(NB), hca(R * 3), hrm(R, 1.f); std::vector<int32_t> hsi(R);
  for (int k = 0; k < NB; ++k) hed[k] = 1e-5f + k * (20.f - 1e-5f) / (NB - 1);
  for (long r = 0; r < R; ++r) { hsi[r] = (int)(r % N); for (int c = 0; c < 3; ++c) hca[r * 3 + c] = 3.8f * (r % N) * (c == 0) * 0.3f + 0.1f * ((r * 7 + c) % 13); }
  (void)hipMemcpy(ed, hed.data(), NB * 4, hipMemcpyHostToDevice); (void)hipMemcpy(ca, hca.data(), R * 12, hipMemcpyHostToDevice);
  (void)hipMemcpy(rm, hrm.data(), R * 4, hipMemcpyHostToDevice); (void)hipMemcpy(si, hsi.data(), R * 4, hipMemcpyHostToDevice);
  a.B = B; a.N = N; a.n_rel = n_rel; a.rel_off = N - 1; a.num_bins = NB; a.pi = pi; a.pj = pj; a.rtab = rt; a.dtab = dt; a.edges = ed;
  a.seq_idx = si; a.sc_ca = ca; a.w2 = a.w3 = nullptr; a.b2 = vecs; a.b3 = vecs + 128; a.gamma = vecs + 256; a.beta = vecs + 384;
  a.res_mask = rm; a.z_out = z; a.trace = nullptr; a.wb_img = (argc > 3 && atoi(argv[3]) == 0) ? nullptr : wb; a.bb = bb; a.bias_out = bo; a.H = 8;
  // round 6: + pair_z of the first block (argv[3] = 2: bias only, as round 5 ran it)
  if (a.wb_img && !(argc > 3 && atoi(argv[3]) == 2)) {
    void *dzh, *dzl; half_t* pz;
    (void)hipMalloc(&dzh, 8192); (void)hipMalloc(&dzl, 8192); (void)hipMalloc(&pz, fd_pz_bytes(B, N));
    (void)hipMemset(dzh, 0, 8192); (void)hipMemset(dzl, 0, 8192);
    a.wdz_img = dzh; a.wdz_img_lo = dzl; a.bdz = vecs; a.pz_out = pz;
  }
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 3; ++i) if (fd_edge_embed2(a, img, 0)) { printf("launch failed\n"); return 1; }
  (void)hipEventRecord(t0, 0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) fd_edge_embed2(a, img, 0);
  (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
  float ms; (void)hipEventElapsedTime(&ms, t0, t1);
  printf("EE2 N=%d B=%d resident=%d early=%d: %.1f us/launch (z write %.2f TB/s)\n", N, B, EE2_RESIDENT, EE2_EARLY, ms / iters * 1e3,
         P * 256.0 / (ms / iters) / 1e9);
#ifdef EE2_PROF
  {
    std::vector<unsigned> h(256 * 8);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(ee2_prof), h.size() * 4);
    const char* names[6] = {"kernel prologue (images -> LDS)", "group setup + first request", "gather (wait + sum + stage)", "row ids + H1 + layer 2",
                            "layer 3", "request + LN epilogue + stores"};
    const double tiles = (double)B * ((N + 31) / 32) * N / (256.0 * 8);
    double tot = 0;
    for (int k = 0; k < 6; ++k) {
      double s = 0;
      for (int b = 0; b < 256; ++b) s += h[b * 8 + k];
      s /= 256.0; tot += s;
      printf("  %-36s %9.0f cyc per wave-launch, %7.0f per tile\n", names[k], s, s / tiles);
    }
    printf("  %-36s %9.0f cyc (wave 0), %.1f tiles per wave\n", "total", tot, tiles);
  }
#endif
  return 0;
}
