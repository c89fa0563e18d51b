// Stand-alone timing of the fp32 EdgeTransition kernels (fp32 mode, pair_mlp.hip), their in-kernel clock, a bit-exact check of the
// wave-specialised kernel against the fused 4-wave kernel, and two probes:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w [-DETF_PROF | -DETF_PROF2] [-DETF_DMA=0] [-DETF_ABL=<bits>] tools/micro/etf_bench.hip -o etf_bench
//   etf_bench N B        timing (zero operands: the matrix cores at their best clock)
//   etf_bench N B 1      check on random operands (exit code 1 on any differing bit)
//   -DETF_PROF           phase profile of the fused kernel;  -DETF_PROF2  s_memtime probe of the wave-specialised kernel: cycles per step /
//                        at the step barrier per layer (multiplier wave 0) and cycles a mover needs to issue a step's requests
//   -DETF_ABL bits (timing only, results wrong): 1 no step barrier, 4 no weight requests, 16 weight tiles as consecutive memory, 32 nobody
//                        waits for the tiles, 64 no LayerNorm / stores, 128 no operand reads, 256 no pass epilogues (pair_mlp.hip)
#include "../../framedipt_amd/csrc/pair_mlp.hip"
#include <cstdio>
#include <vector>
#include <cstring>
#include <cmath>
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
  if (argc > 3 && atoi(argv[3]) == 1) {
    // check: the wave-specialised kernel (movers as LDS-DMA when ETF_DMA) against the fused 4-wave kernel on random operands, bit for bit
    std::vector<float> hz(P * 128), he(R * 128), hw(2 * 384 * 384 + 128 * 384), hv(2048), hrm(R);
    unsigned sd = 12345u;
    auto rnd = [&]() { sd = sd * 1664525u + 1013904223u; return ((sd >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& x : hz) x = rnd();
    for (auto& x : he) x = rnd();
    for (auto& x : hw) x = rnd() * 0.05f;
    for (auto& x : hv) x = rnd() * 0.5f;
    for (long r = 0; r < R; ++r) hrm[r] = (r % 11) ? 1.f : 0.f;
    (void)hipMemcpy(z, hz.data(), hz.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(e, he.data(), he.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(w, hw.data(), hw.size() * 4, hipMemcpyHostToDevice); (void)hipMemcpy(v, hv.data(), hv.size() * 4, hipMemcpyHostToDevice);
    (void)hipMemcpy(rm, hrm.data(), hrm.size() * 4, hipMemcpyHostToDevice);
    float *o1, *o2;
    (void)hipMalloc(&o1, P * 128 * 4); (void)hipMalloc(&o2, P * 128 * 4);
    (void)hipFuncSetAttribute((const void*)edge_transition_f32_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, ETF_LDS);
    (void)hipFuncSetAttribute((const void*)edge_transition_f32ws_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, ETF_LDS);
    const int nb = (int)((P + 31) / 32), grid = nb < 256 ? nb : 256;
    a.z_out = o1;
    hipLaunchKernelGGL(edge_transition_f32_kernel<float>, dim3(grid), dim3(FD_THREADS), ETF_LDS, 0, a, nb);
    a.z_out = o2;
    hipLaunchKernelGGL(edge_transition_f32ws_kernel<float>, dim3(grid), dim3(2 * FD_THREADS), ETF_LDS, 0, a, nb);
    if (hipDeviceSynchronize() != hipSuccess) { printf("check: launch failed\n"); return 1; }
    std::vector<float> h1(P * 128), h2(P * 128);
    (void)hipMemcpy(h1.data(), o1, h1.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(h2.data(), o2, h2.size() * 4, hipMemcpyDeviceToHost);
    long bad = 0; double ss = 0;
    for (size_t i = 0; i < h1.size(); ++i) { bad += memcmp(&h1[i], &h2[i], 4) != 0; ss += (double)h1[i] * h1[i]; }
    printf("check N=%d B=%d: %ld of %zu values differ between the fused and the wave-specialised kernel (rms %.3f)\n", N, B, bad, h1.size(), sqrt(ss / h1.size()));
    return bad != 0;
  }
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 2; ++i) if (fd_edge_transition(FDIPT_PREC_F32, 128, 128, a, 0)) { printf("launch failed\n"); return 1; }
  (void)hipEventRecord(t0, 0);
  const int iters = 5;
  for (int i = 0; i < iters; ++i) fd_edge_transition(FDIPT_PREC_F32, 128, 128, a, 0);
  (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
  float ms; (void)hipEventElapsedTime(&ms, t0, t1);
  {  // shader clock inside the kernel (s_memtime cycles per 100 MHz s_memrealtime tick, one more launch)
    unsigned long long* clk; (void)hipMalloc(&clk, 32); (void)hipMemset(clk, 0, 32);
    a.clock = clk;
    fd_edge_transition(FDIPT_PREC_F32, 128, 128, a, 0);
    unsigned long long h[3]; (void)hipMemcpy(h, clk, 24, hipMemcpyDeviceToHost);
    printf("  in-kernel clock %.2f GHz (%llu blocks)\n", h[1] ? 0.1 * h[0] / h[1] : 0.0, h[2]);
    a.clock = nullptr;
  }
  const double tf = 688128.0 * P / (ms / iters) / 1e9;
  printf("ET fp32 N=%d B=%d: %.3f ms/launch, %.1f TFLOP/s = %.1f %% of 157.3\n", N, B, ms / iters, tf, tf / 1.573);
#ifdef ETF_PROF2
  {
    unsigned long long h[16];
    (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(etf_prof2), 128);
    for (int l = 0; l < 3; ++l) {
      const double n = (double)h[2] / 84 * (l < 2 ? 36 : 12);
      printf("  layer %d: %.0f cycles per step, %.0f at the barrier\n", l + 1, h[9 + 2 * l] / n, h[8 + 2 * l] / n);
    }
    printf("  row tile: %.0f cycles, %.0f of them in its 84 steps\n", (double)h[6] / h[7], (double)h[1] / h[7]);
    if (h[5]) printf("  mover wave 0 (layers 1, 2): %.0f cycles from the step barrier to its wait's end, %.0f of them issuing the five requests\n", (double)h[3] / h[5], (double)h[4] / h[5]);
    printf("  multiplier wave 0: %.0f cycles per step, %.0f of them at the step barrier (%llu steps; matrix work 1024)\n", (double)h[1] / h[2], (double)h[0] / h[2], h[2]);
  }
#endif
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
