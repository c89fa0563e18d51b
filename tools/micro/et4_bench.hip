// Stand-alone timing + in-kernel phase profile of edge_transition4_kernel (build with -DFD_PROF for the profile).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w [-DFD_PROF] [-DE4_ABL=k] tools/micro/et4_bench.hip -o et4_bench
#include "../../framedipt_amd/csrc/edge_transition4.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int B = 8, N = argc > 1 ? atoi(argv[1]) : 300;
  const long P = (long)B * N * N, R = (long)B * N;
  half_t* z; float *e, *a1, *af, *b2, *g, *bt, *rm; void* stream;
  (void)hipMalloc(&z, P * 128 * 2); (void)hipMalloc(&e, R * 128 * 4); (void)hipMalloc(&a1, R * 384 * 4); (void)hipMalloc(&af, R * 128 * 4);
  (void)hipMalloc(&b2, 384 * 4); (void)hipMalloc(&g, 128 * 4); (void)hipMalloc(&bt, 128 * 4); (void)hipMalloc(&rm, R * 4);
  (void)hipMalloc(&stream, fd_et4_stream_bytes());
  (void)hipMemset(z, 0, P * 128 * 2); (void)hipMemset(e, 0, R * 128 * 4); (void)hipMemset(a1, 0, R * 384 * 4); (void)hipMemset(af, 0, R * 128 * 4);
  (void)hipMemset(b2, 0, 384 * 4); (void)hipMemset(g, 0, 128 * 4); (void)hipMemset(bt, 0, 128 * 4); (void)hipMemset(rm, 0, R * 4);
  (void)hipMemset(stream, 0, fd_et4_stream_bytes());
  ET2Args a; a.B = B; a.N = N; a.z_in = z; a.z_out = z; a.e = e; a.a1 = a1; a.af = af; a.stream = stream; a.b2 = b2; a.gamma = g;
  a.beta = bt; a.res_mask = rm; a.trace = nullptr;
  { char* ai; const size_t na = fd_et4_a_image_bytes(B, N), nb = fd_et4_b_image_bytes(B, N);  // one allocation: 32-bit offsets between the images
    (void)hipMalloc(&ai, na + nb); (void)hipMemset(ai, 0, na + nb); a.a1_img = ai; a.b1_img = ai + na; a.e_h16 = nullptr; }
  {  // next block's pair bias from the epilogue (argv[2] = 0 disables)
    void* wimg; float* bo; const long Np = (N + 31) / 32 * 32;
    (void)hipMalloc(&wimg, 8192); (void)hipMemset(wimg, 0, 8192); (void)hipMalloc(&bo, (size_t)B * 8 * Np * Np * 4);
    a.wb_img = (argc > 2 && atoi(argv[2]) == 0) ? nullptr : wimg; a.bb = b2; a.bias_out = bo; a.H = 8;
  }
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 3; ++i) fd_edge_transition4(a, 0);
  (void)hipEventRecord(t0, 0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) fd_edge_transition4(a, 0);
  (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
  float ms; (void)hipEventElapsedTime(&ms, t0, t1);
  const double flops = 655360.0 * P;
  printf("ET4 N=%d: %.3f ms/launch, %.1f TFLOP/s (%.1f%% of 2500)\n", N, ms / iters, flops / (ms / iters) / 1e9, flops / (ms / iters) / 1e9 / 25.0);
#ifdef E4_PROF
  {
    std::vector<unsigned> h(256 * 8);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(e4_prof), h.size() * 4);
    const char* names[6] = {"kernel prologue", "layer 1 (4 chunks, 108 MFMA/wave)", "layer 2 (12 chunks, 288)", "final (4 chunks, 132)", "request + LN epilogue + stores", "tile-end wait + barrier"};
    const int n_wt = B * ((N + 7) / 8) * (N / 4), n_tiles = (n_wt + 7) / 8;
    double tot = 0;
    for (int k = 0; k < 6; ++k) {
      double s = 0;
      for (int b = 0; b < 256; ++b) s += h[b * 8 + k];
      s /= 256.0; tot += s;
      printf("  %-36s %9.0f cyc per block-launch, %7.0f per tile\n", names[k], s, s / (n_tiles / 256.0));
    }
    printf("  %-36s %9.0f cyc (wave 0)\n", "total", tot);
    std::vector<unsigned long long> sp(1024 * 3);
    (void)hipMemcpyFromSymbol(sp.data(), HIP_SYMBOL(e4_span), sp.size() * 8);
    unsigned long long t0 = ~0ull, t1 = 0;
    int nb = 0;
    for (int b = 0; b < 1024 && sp[b * 3 + 1]; ++b) { ++nb; if (sp[b * 3] < t0) t0 = sp[b * 3]; if (sp[b * 3 + 1] > t1) t1 = sp[b * 3 + 1]; }
    printf("  blocks %d, launch span %.1f us (100 MHz ticks)\n", nb, (t1 - t0) / 100.0);
    for (int b = 0; b < nb; b += nb / 16) {
      const unsigned hw = (unsigned)sp[b * 3 + 2], xcc = (unsigned)(sp[b * 3 + 2] >> 32) & 15;
      printf("    block %4d: start %7.1f end %7.1f us  xcc %u se %u cu %u simd %u\n", b, (sp[b * 3] - t0) / 100.0, (sp[b * 3 + 1] - t0) / 100.0, xcc,
             (hw >> 13) & 7, (hw >> 8) & 15, (hw >> 4) & 3);
    }
    // blocks sharing a CU
    int shared = 0;
    for (int b = 0; b < nb; ++b)
      for (int c = b + 1; c < nb; ++c) {
        const unsigned long long m = 0xF0000FF00ull | (0x7ull << 13);
        if ((sp[b * 3 + 2] & m) == (sp[c * 3 + 2] & m) && sp[b * 3] < sp[c * 3 + 1] && sp[c * 3] < sp[b * 3 + 1]) ++shared;
      }
    printf("  overlapping block pairs on one CU: %d\n", shared);
  }
#endif
  return 0;
}
