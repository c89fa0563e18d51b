// Stand-alone timing + phase profile (-DFD_PROF) of ipa_attn4_kernel (tools/micro/attention4_experiment.hip: measured, not adopted).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w [-DFD_PROF] tools/micro/attn4_bench.hip -o attn4_bench
#include "attention4_experiment.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int B = 8, H = 8, N = argc > 1 ? atoi(argv[1]) : 300, Np = (N + 31) / 32 * 32;
  const int split = argc > 2 ? atoi(argv[2]) : 1;
  Attn3Args a; a.out_h16 = nullptr;
  a.B = B; a.N = N; a.H = H; a.Np = Np;
  auto dz = [](size_t bytes) { void* p; (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); return p; };
  a.Qb = (const half_t*)dz((size_t)B * H * Np * 256 * 2); a.Kb = (const half_t*)dz((size_t)B * H * Np * 256 * 2);
  a.Vt = (const half_t*)dz((size_t)B * H * 256 * Np * 2); a.Vt_lo = split ? (const half_t*)dz((size_t)B * H * 256 * Np * 2) : nullptr;
  a.bias = (const float*)dz((size_t)B * H * Np * Np * 4);
  a.res_mask = (const float*)dz((size_t)B * N * 4); a.qp = (const float*)dz((size_t)B * N * H * 24 * 4);
  a.kp = (const float*)dz((size_t)B * N * H * 24 * 4); a.vp = (const float*)dz((size_t)B * N * H * 36 * 4);
  a.vpt = (const half_t*)dz((size_t)B * H * 96 * Np * 2);
  a.gamma = (const float*)dz(64); a.rot = (const float*)dz((size_t)B * N * 9 * 4); a.trans = (const float*)dz((size_t)B * N * 3 * 4);
  a.probs = nullptr; a.probs_h16 = (half_t*)dz((size_t)B * H * N * Np * 2);
  a.out_ld = 2432; a.out = (float*)dz((size_t)B * N * a.out_ld * 4); a.pt_off = H * 256;
  if (!fd_attention4_supported(a)) { printf("unsupported\n"); return 1; }
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 3; ++i) fd_attention4(a, 0);
  (void)hipEventRecord(t0, 0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) fd_attention4(a, 0);
  (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
  float ms; (void)hipEventElapsedTime(&ms, t0, t1);
  printf("attn4 N=%d split=%d: %.1f us/launch\n", N, split, ms * 1000 / iters);
#ifdef FD_PROF
  const int nt = Np / 32, nb = 8 * ((B * H + 7) / 8) * ((nt + 3) / 4);
  std::vector<unsigned long long> h((size_t)nb * 16);
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(fd_prof), h.size() * 8);
  const char* names[5] = {"", "query-side loads", "pass 1 (stats)", "pass 2 (weights, P V)", "epilogue"};
  double tot = 0;
  for (int k = 1; k < 5; ++k) {
    double s = 0;
    for (int b = 0; b < nb; ++b) s += (double)(h[(size_t)b * 16 + k] - h[(size_t)b * 16 + (k - 1)]);
    s /= nb; tot += s;
    printf("  %-30s %8.0f cyc\n", names[k], s);
  }
  printf("  %-30s %8.0f cyc (wave 0 of each block)\n", "total per block", tot);
  const char* n2[16] = {"","","","","","", "p1 t=4: vm_wait", "p1: barrier", "p1: requests", "p1: logits", "p1: stats", "(p1 tail + p2 head)", "p2 t=4: wait+barrier", "p2: requests", "p2: logits", "p2: softmax + P V"};
  for (int k = 6; k < 16; ++k) {
    double s = 0;
    for (int b = 0; b < nb; ++b) s += (double)(h[(size_t)b * 16 + k] - h[(size_t)b * 16 + (k - 1)]);
    printf("  %-30s %8.0f cyc\n", n2[k], s / nb);
  }
#endif
  return 0;
}
