// Stand-alone timing of the fused chain kernels (includes chain.hip directly so that -DCH_ABL=... ablations are one build).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w -I framedipt_amd/csrc tools/micro/chain_bench.hip -o chain_bench
#include "../../framedipt_amd/csrc/chain.hip"
#include <cstdio>
#include <vector>
template <class F> static float timeit(F f, int iters) {
  hipEvent_t a, b; (void)hipEventCreate(&a); (void)hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  (void)hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) f();
  (void)hipEventRecord(b, 0); (void)hipEventSynchronize(b);
  float ms; (void)hipEventElapsedTime(&ms, a, b); return ms * 1000.f / iters;
}
int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 2400;
  float *in, *res, *out, *bias, *mask; void* img;
  (void)hipMalloc(&in, (size_t)M * 1024 * 4); (void)hipMalloc(&res, (size_t)M * 1024 * 4); (void)hipMalloc(&out, (size_t)M * 1024 * 4);
  (void)hipMalloc(&bias, 4096 * 4); (void)hipMalloc(&mask, (size_t)M * 4); (void)hipMalloc(&img, 4 << 20);
  (void)hipMemset(in, 0, (size_t)M * 1024 * 4); (void)hipMemset(res, 0, (size_t)M * 1024 * 4); (void)hipMemset(bias, 0, 4096 * 4);
  (void)hipMemset(mask, 0, (size_t)M * 4); (void)hipMemset(img, 0, 4 << 20);
  struct K { int kind; const char* name; int k0, nout; bool ln, resid; } kinds[] = {
      {FD_CHAIN_TRANSITION, "TRANSITION", 256, 256, true, true}, {FD_CHAIN_FFN, "FFN", 320, 320, true, true},
      {FD_CHAIN_OUTPROJ, "OUTPROJ", 320, 320, true, true},       {FD_CHAIN_POST, "POST", 320, 256, false, true},
      {FD_CHAIN_INPROJ, "INPROJ", 320, 960, false, false},       {FD_CHAIN_SKIP, "SKIP", 256, 64, false, false},
      {FD_CHAIN_ETINIT, "ETINIT", 256, 128, false, false},       {FD_CHAIN_A1, "A1", 128, 384, false, false},
      {FD_CHAIN_AF, "AF", 128, 128, false, false},               {FD_CHAIN_NODE_EMBED_72, "NE72", 72, 256, true, false},
      {FD_CHAIN_TORSION, "TORSION", 256, 256, false, true}};
  for (auto& k : kinds) {
    ChainArgs a;
    a.M = M; a.in = in; a.ld_in = k.k0; a.w[0] = a.w[1] = a.w[2] = img; a.b[0] = a.b[1] = a.b[2] = bias;
    a.residual = k.resid ? res : nullptr; a.ld_res = k.nout; a.gamma = bias; a.beta = bias; a.rowmask_pre = nullptr;
    a.rowmask_post = k.ln ? mask : nullptr; a.out = out; a.ld_out = k.nout;
    int rc = 0;
    const float us = timeit([&] { rc |= fd_chain(k.kind, a, 0); }, 200);
    printf("%-11s %7.2f us  rc=%d\n", k.name, us, rc);
  }
  return 0;
}
