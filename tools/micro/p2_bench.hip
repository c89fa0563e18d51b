// Stand-alone timing + phase profile (-DFD_PROF) of ipa_proj2_kernel, zero inputs, B = 8, N = 300 (M = 2400), c_s = 256.
#include "../../framedipt_amd/csrc/ipa_proj2.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int split = argc > 1 ? atoi(argv[1]) : 0;
  const int B = 8, N = 300, H = 8, C = 256, K = 256, PT = 672, Np = 320, M = B * N, NOUT = 3 * H * C + PT;
  float *A, *bias, *pts; void *wimg, *wimg_lo; half_t *Qb, *Kb, *Vt, *Vtl;
  const size_t img = (size_t)((NOUT + 127) / 128) * 65536, qs = (size_t)B * H * Np * C * 2;
  (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&bias, NOUT * 4); (void)hipMalloc(&pts, (size_t)M * PT * 4); (void)hipMalloc(&wimg, img);
  (void)hipMalloc(&Qb, qs); (void)hipMalloc(&Kb, qs); (void)hipMalloc(&Vt, qs); (void)hipMalloc(&Vtl, qs); (void)hipMalloc(&wimg_lo, img); (void)hipMemset(wimg_lo, 0, img);
  (void)hipMemset(A, 0, (size_t)M * K * 4); (void)hipMemset(bias, 0, NOUT * 4); (void)hipMemset(wimg, 0, img);
  ProjArgs a; a.B = B; a.N = N; a.H = H; a.C = C; a.K = K; a.PT = PT; a.Np = Np; a.A = A; a.lda = K; a.W = nullptr; a.W_img = wimg; a.bias = bias;
  a.qscale = 1.f; a.Qb = Qb; a.Kb = Kb; a.Vt = Vt; a.pts = pts; a.zero_pads = 0; if (split) { a.W_img_lo = wimg_lo; a.Vt_lo = Vtl; }
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 3; ++i) if (fd_ipa_proj2(a, 0)) { printf("launch failed\n"); return 1; }
  (void)hipEventRecord(t0, 0);
  const int iters = 50;
  for (int i = 0; i < iters; ++i) fd_ipa_proj2(a, 0);
  (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
  float ms; (void)hipEventElapsedTime(&ms, t0, t1);
  printf("ipa_proj2 M=%d: %.2f us/launch\n", M, ms / iters * 1e3);
#ifdef FD_PROF
  {
    const int nx = split ? 10 : 19, ny = split ? 25 : 13;
    std::vector<unsigned long long> h((size_t)8192 * 16);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(fd_prof), h.size() * 8);
    for (int y : {0, 7, 13, 14, ny - 1}) {
      const unsigned long long* p = &h[(size_t)(0 + nx * y) * 16];
      printf("  walker %2d:", y);
      for (int k = 1; k < 16; ++k) printf(" %7lld", p[k] > p[k - 1] ? (long long)(p[k] - p[k - 1]) : -1LL);
      printf("\n");
    }
    (void)ny;
  }
#endif
  return 0;
}
