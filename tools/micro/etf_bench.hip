// Stand-alone timing + phase profile of edge_transition_f32_kernel (fp32 mode, pair_mlp.hip); -DETF_PROF for the profile.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w [-DETF_PROF] tools/micro/etf_bench.hip -o etf_bench
#include "../../framedipt_amd/csrc/pair_mlp.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int N = argc > 1 ? atoi(argv[1]) : 300, B = argc > 2 ? atoi(argv[2]) : 8;
  const long P = (long)B * N * N, R = (long)B * N;
  float *z, *e, *w, *v, *rm;
  (void)hipMalloc(&z, P * 128 * 4); (void)hipMalloc(&e, R * 128 * 4); (void)hipMalloc(&w, (2 * 384 * 384 + 128 * 384) * 4);
  (void)hipMalloc(&v, 2048 * 4); (void)hipMalloc(&rm, R * 4);
  (void)hipMemset(z, 0, P * 128 * 4); (void)hipMemset(e, 0, R * 128 * 4); (void)hipMemset(w, 0, (2 * 384 * 384 + 128 * 384) * 4);
  (void)hipMemset(v, 0, 2048 * 4); (void)hipMemset(rm, 0, R * 4);
  EdgeTransArgs a; a.B = B; a.N = N; a.z_in = z; a.z_out = z; a.e = e; a.w1 = w; a.w2 = w + 384 * 384; a.wf = w + 2 * 384 * 384;
  a.b1 = v; a.b2 = v + 384; a.bf = v + 768; a.gamma = v + 1024; a.beta = v + 1280; a.res_mask = rm; a.trace = nullptr;
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 2; ++i) if (fd_edge_transition(FDIPT_PREC_F32, 128, 128, a, 0)) { printf("launch failed\n"); return 1; }
  (void)hipEventRecord(t0, 0);
  const int iters = 5;
  for (int i = 0; i < iters; ++i) fd_edge_transition(FDIPT_PREC_F32, 128, 128, a, 0);
  (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
  float ms; (void)hipEventElapsedTime(&ms, t0, t1);
  const double tf = 688128.0 * P / (ms / iters) / 1e9;
  printf("ET fp32 N=%d B=%d: %.3f ms/launch, %.1f TFLOP/s = %.1f %% of 157.3\n", N, B, ms / iters, tf, tf / 1.573);
#ifdef ETF_PROF
  {
    std::vector<unsigned> h(256 * 8);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(etf_prof), h.size() * 4);
    const char* names[5] = {"stage X0 + first tiles", "layer 1 (36 tiles, 576 MFMA/wave)", "layer 2 (36 tiles)", "final layer (12 tiles)", "LayerNorm + stores"};
    double tot = 0;
    for (int k = 0; k < 5; ++k) {
      double s = 0;
      for (int b = 0; b < 256; ++b) s += h[b * 8 + k];
      s /= 256.0; tot += s;
      printf("  %-36s %9.0f cyc per block\n", names[k], s);
    }
    printf("  %-36s %9.0f cyc (matrix work: 86016)\n", "total", tot);
  }
#endif
  return 0;
}
