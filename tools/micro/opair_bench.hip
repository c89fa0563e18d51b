// Stand-alone timing + phase profile (-DFD_PROF) of opair_kernel<bf16, 128>.
#include "../../framedipt_amd/csrc/attention.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int B = 8, H = 8, N = argc > 1 ? atoi(argv[1]) : 300;
  auto dz = [](size_t bytes) { void* p; (void)hipMalloc(&p, bytes); (void)hipMemset(p, 0, bytes); return p; };
  OPairArgs a; a.probs_h16 = nullptr; a.probs_np = 0; a.out_h16 = nullptr;
  a.B = B; a.N = N; a.H = H; a.CZ = 128; a.CD = 32; a.z = dz((size_t)B * N * N * 128 * 2); a.probs = (const float*)dz((size_t)B * H * N * N * 4);
  a.wdz = (const float*)dz(128 * 32 * 4); a.wdz_img = dz(8192); a.bdz = (const float*)dz(32 * 4); a.out_ld = 2688; a.out = (float*)dz((size_t)B * N * a.out_ld * 4);
  a.off = 2048 + 384;
  const int Np64 = (N + 63) / 64 * 64;
  if (argc > 2) {  // product configuration: half-precision weight rows from the attention, split down-projection
    a.probs_h16 = (const half_t*)dz((size_t)B * N * H * Np64 * 2); a.probs_np = Np64; a.wdz_img_lo = dz(8192);
    (void)hipMemset((void*)a.z, 0x3c, (size_t)B * N * N * 128 * 2);  // (non-zero operands)
  }
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 3; ++i) fd_opair(FDIPT_PREC_HALF, a, 0);
  (void)hipEventRecord(t0, 0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) fd_opair(FDIPT_PREC_HALF, a, 0);
  (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
  float ms; (void)hipEventElapsedTime(&ms, t0, t1);
  printf("opair N=%d: %.1f us/launch (%.2f TB/s of z)\n", N, ms * 1000 / iters, (double)B * N * N * 256 / (ms / iters * 1e-3) / 1e12);
#ifdef FD_PROF
  const int nb = B * N;
  std::vector<unsigned long long> h((size_t)nb * 16);
  (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(fd_prof), h.size() * 8);
  const char* names[6] = {"", "prologue (probs, wdz, z0, barrier)", "main loop", "wave fold + psum + barrier", "cross-wave sum + barrier", "down-projection"};
  double tot = 0;
  for (int k = 1; k < 6; ++k) {
    double s = 0;
    for (int b = 0; b < nb; ++b) s += (double)(h[(size_t)b * 16 + k] - h[(size_t)b * 16 + k - 1]);
    s /= nb; tot += s;
    printf("  %-36s %8.0f cyc\n", names[k], s);
  }
  printf("  %-36s %8.0f cyc\n", "total per block", tot);
#endif
  return 0;
}
