// Stand-alone timing + in-kernel phase profile of edge_transition2_kernel (build with -DET2_PROF for the profile).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w [-DET2_PROF] tools/micro/et2_bench.hip -o et2_bench
#include "../../framedipt_amd/csrc/edge_transition3.hip"
#include <cstdio>
#include <vector>
int main(int argc, char** argv) {
  const int B = 8, N = argc > 1 ? atoi(argv[1]) : 300;
  const long P = (long)B * N * N, R = (long)B * N;
  half_t* z; float *e, *a1, *af, *b2, *g, *bt, *rm; void* stream;
  (void)hipMalloc(&z, P * 128 * 2); (void)hipMalloc(&e, R * 128 * 4); (void)hipMalloc(&a1, R * 384 * 4); (void)hipMalloc(&af, R * 128 * 4);
  (void)hipMalloc(&b2, 384 * 4); (void)hipMalloc(&g, 128 * 4); (void)hipMalloc(&bt, 128 * 4); (void)hipMalloc(&rm, R * 4);
  (void)hipMalloc(&stream, fd_et3_stream_bytes());
  (void)hipMemset(z, 0, P * 128 * 2); (void)hipMemset(e, 0, R * 128 * 4); (void)hipMemset(a1, 0, R * 384 * 4); (void)hipMemset(af, 0, R * 128 * 4);
  (void)hipMemset(b2, 0, 384 * 4); (void)hipMemset(g, 0, 128 * 4); (void)hipMemset(bt, 0, 128 * 4); (void)hipMemset(rm, 0, R * 4);
  (void)hipMemset(stream, 0, fd_et3_stream_bytes());
  ET2Args a; a.B = B; a.N = N; a.z_in = z; a.z_out = z; a.e = e; a.a1 = a1; a.af = af; a.stream = stream; a.b2 = b2; a.gamma = g;
  a.beta = bt; a.res_mask = rm; a.trace = nullptr;
  { half_t* eb; (void)hipMalloc(&eb, R * 128 * 2); (void)hipMemset(eb, 0, R * 128 * 2); a.e_h16 = eb; }
  {  // next block's pair bias from the epilogue (argv[2] = 0 disables)
    void* wimg; float* bo; const long Np = (N + 31) / 32 * 32;
    (void)hipMalloc(&wimg, 8192); (void)hipMemset(wimg, 0, 8192); (void)hipMalloc(&bo, (size_t)B * 8 * Np * Np * 4);
    a.wb_img = (argc > 2 && atoi(argv[2]) == 0) ? nullptr : wimg; a.bb = b2; a.bias_out = bo; a.H = 8;
  }
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 3; ++i) fd_edge_transition3(a, 0);
  (void)hipEventRecord(t0, 0);
  const int iters = 20;
  for (int i = 0; i < iters; ++i) fd_edge_transition3(a, 0);
  (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
  float ms; (void)hipEventElapsedTime(&ms, t0, t1);
  const double flops = 655360.0 * P;
  printf("ET3 N=%d: %.3f ms/launch, %.1f TFLOP/s (%.1f%% of 2500)\n", N, ms / iters, flops / (ms / iters) / 1e9, flops / (ms / iters) / 1e9 / 25.0);
#ifdef FD_PROF
  {
    const int nb = (int)((P + 127) / 128) < 8192 ? (int)((P + 127) / 128) : 8192;
    std::vector<unsigned long long> h((size_t)nb * 16);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(fd_prof), h.size() * 8);
    const char* names[6] = {"", "prologue", "layer 1 (3 chunks, 192 MFMA/wave)", "layer 2 (6 chunks, 288)", "final (4 chunks, 160)", "LN epilogue + stores"};
    double tot = 0;
    for (int k = 1; k < 6; ++k) {
      double s = 0;
      for (int b = 0; b < nb; ++b) s += (double)(h[(size_t)b * 16 + k] - h[(size_t)b * 16 + k - 1]);
      s /= nb; tot += s;
      printf("  %-36s %8.0f cyc\n", names[k], s);
    }
    printf("  %-36s %8.0f cyc (wave 0)\n", "total per tile", tot);
  }
#endif
  return 0;
}
