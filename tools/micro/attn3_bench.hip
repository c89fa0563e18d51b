// Stand-alone timing + phase profile (-DFD_PROF) of ipa_attn3_kernel.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w [-DFD_PROF] tools/micro/attn3_bench.hip -o attn3_bench
#include "../../framedipt_amd/csrc/attention3.hip"
#include <cstdio>
#include <vector>
#include <algorithm>
int main(int argc, char** argv) {
  const int B = argc > 2 ? atoi(argv[2]) : 8, H = 8, N = argc > 1 ? atoi(argv[1]) : 300, Np = (N + 31) / 32 * 32;
  Attn3Args a; a.out_h16 = nullptr;
  a.B = B; a.N = N; a.H = H; a.Np = Np;
  auto dz = [](size_t bytes) { void* p; (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); return p; };
  a.Qb = (const half_t*)dz((size_t)B * H * N * 256 * 2); a.Kb = (const half_t*)dz((size_t)B * H * N * 256 * 2);
  a.Vt = (const half_t*)dz((size_t)B * H * 256 * Np * 2); a.bias = (const float*)dz((size_t)B * H * Np * Np * 4);
  a.res_mask = (const float*)dz((size_t)B * N * 4); a.qp = (const float*)dz((size_t)B * N * H * 24 * 4);
  a.kp = (const float*)dz((size_t)B * N * H * 24 * 4); a.vp = (const float*)dz((size_t)B * N * H * 36 * 4);
  a.vpt = (const half_t*)dz((size_t)B * H * 96 * Np * 2);
  // the product configuration (rounds 3 - 4): merged projection (K / V images per sample), split P V (V_lo), key-point fragment image, fp16 probs rows
  a.kv_per_sample = 1; a.Vt_lo = (const half_t*)dz((size_t)B * H * 256 * Np * 2);
  a.kpf = (const half_t*)dz((size_t)B * H * (Np / 32) * FD_KPF_FRAGS * 1024);
  a.probs_h16 = (half_t*)dz((size_t)B * N * H * Np * 2);
  a.gamma = (const float*)dz(64); a.rot = (const float*)dz((size_t)B * N * 9 * 4); a.trans = (const float*)dz((size_t)B * N * 3 * 4);
  a.probs = (float*)dz((size_t)B * H * N * N * 4); a.out_ld = 2432; a.out = (float*)dz((size_t)B * N * a.out_ld * 4); a.pt_off = H * 256;
  if (!fd_attention3_supported(a)) { printf("unsupported\n"); return 1; }
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 3; ++i) fd_attention3(a, 0);
  (void)hipEventRecord(t0, 0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) fd_attention3(a, 0);
  (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
  float ms; (void)hipEventElapsedTime(&ms, t0, t1);
  printf("attn3 N=%d B=%d: %.1f us/launch\n", N, B, ms * 1000 / iters);
  {  // resident blocks per CU at this launch's dynamic LDS (N <= 384: ipa_attn3_kernel<3, 3, false, true>)
    const int nt = Np / 32;
    const size_t pf = (size_t)2 * nt * 64 * 16, base = 2 * 128 * 4 + (size_t)32 * 96 * 4 + pf + 16, smem = base + (16384 > pf ? 16384 : pf);
    for (size_t sm : {smem, smem - 1024, smem - 2048, (size_t)49152}) {
      int nb = -1;
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ipa_attn3_kernel<3, 3, false, true>, 256, sm);
      printf("  occupancy at %zu B of dynamic LDS: %d blocks per CU\n", sm, nb);
    }
  }
#ifdef FD_PROF
  const int nb = (Np / 32) * H * B;
  std::vector<unsigned long long> h((size_t)8 * ((B * H + 7) / 8) * (Np / 32) * 16);
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(fd_prof), h.size() * 8);
  const char* names[8] = {"", "prologue (Q -> LDS, q_pts)", "phase 1 logits", "phase 2 softmax", "phase 3 probs/P", "barrier", "-", "phase 4 PV + o_pt"};
  double tot = 0;
  for (int k = 1; k < 8; ++k) {
    double s = 0;
    for (int b = 0; b < nb; ++b) s += (double)(h[(size_t)b * 16 + k] - h[(size_t)b * 16 + (k - 1)]);
    s /= nb; tot += s;
    printf("  %-30s %8.0f cyc\n", names[k], s);
  }
  printf("  %-30s %8.0f cyc (wave 0 of each block)\n", "total per block", tot);
  {  // dispatch timeline of the last launch: start / end of every block relative to the earliest start (s_memtime: shader clock)
    // (chip-wide 100 MHz clock, slots 8 .. 15 of a block's stamps: s_memtime counters differ between XCDs)
    const int nbl = (int)h.size() / 16;
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < nbl; ++b) if (h[(size_t)b * 16 + 8]) t0 = std::min(t0, h[(size_t)b * 16 + 8]);
    std::vector<double> st, en;
    for (int b = 0; b < nbl; ++b) if (h[(size_t)b * 16 + 8]) { st.push_back((double)(h[(size_t)b * 16 + 8] - t0) * 0.01); en.push_back((double)(h[(size_t)b * 16 + 15] - t0) * 0.01); }
    std::sort(st.begin(), st.end()); std::sort(en.begin(), en.end());
    auto q = [&](std::vector<double>& v, double f) { return v[(size_t)(f * (v.size() - 1))]; };
    printf("  %zu blocks; starts (us after the first): median %.2f, 75%% %.2f, 90%% %.2f, max %.2f; ends: min %.2f median %.2f max %.2f\n", st.size(),
           q(st, 0.5), q(st, 0.75), q(st, 0.9), q(st, 1.0), q(en, 0.0), q(en, 0.5), q(en, 1.0));
  }
#endif
  return 0;
}
