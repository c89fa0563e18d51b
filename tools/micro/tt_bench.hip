// Stand-alone timing + phase profile (-DFD_PROF) of tfmr_tail_kernel (rowblock.hip), zero inputs, M = 2400 rows.
#include "../../framedipt_amd/csrc/rowblock.hip"
#include <cstdio>
#include <vector>
__global__ void tt_fill(float* p, long n, unsigned seed, float sc) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((h & 0xFFFF) / 65536.f - 0.5f) * 2.f * sc;
  }
}
__global__ void tt_fill_h(unsigned short* p, long n, unsigned seed, float sc) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = f2h(((h & 0xFFFF) / 65536.f - 0.5f) * 2.f * sc);
  }
}
int main(int argc, char** argv) {
  const int M = argc > 2 ? atoi(argv[2]) : 2400, D = 320;
  const bool split = argc > 1 && atoi(argv[1]) != 0;  // argv[1] = 1: split operands (hi + lo weight images)
  float *att, *x, *out, *vec; void* w;
  (void)hipMalloc(&att, (size_t)M * D * 4); (void)hipMalloc(&x, (size_t)M * D * 4); (void)hipMalloc(&out, (size_t)M * D * 4);
  (void)hipMalloc(&vec, 7 * D * 4); (void)hipMalloc(&w, 3 * (size_t)D * D * 2);
  (void)hipMemset(att, 0, (size_t)M * D * 4); (void)hipMemset(x, 0, (size_t)M * D * 4); (void)hipMemset(vec, 0, 7 * D * 4); (void)hipMemset(w, 0, 3 * (size_t)D * D * 2);
  TfmrTailArgs a; a.warm = L2Warm{}; a.M = M; a.ld = D; a.att = att; a.x = x; a.out = out;
  a.wo = w; a.w1 = (char*)w + (size_t)D * D * 2; a.w2 = (char*)w + 2 * (size_t)D * D * 2;
  a.wol = a.w1l = a.w2l = a.wpl = nullptr;
  if (split) { void* wl; (void)hipMalloc(&wl, 3 * (size_t)D * D * 2); (void)hipMemset(wl, 0, 3 * (size_t)D * D * 2);
               a.wol = wl; a.w1l = (char*)wl + (size_t)D * D * 2; a.w2l = (char*)wl + 2 * (size_t)D * D * 2; }
  a.rows16 = argc > 3 ? atoi(argv[3]) : 0;  // argv[3] = 1: the 16-row kernel (split only)
  a.bo = vec; a.g1 = vec + D; a.be1 = vec + 2 * D; a.b1 = vec + 3 * D; a.b2 = vec + 4 * D; a.g2 = vec + 5 * D; a.be2 = vec + 6 * D;
  if (argc > 4 && atoi(argv[4])) {  // argv[4] = 1: random operands instead of zeros
    tt_fill<<<256, 256>>>(att, (long)M * D, 1u, 1.f); tt_fill<<<256, 256>>>(x, (long)M * D, 2u, 1.f); tt_fill<<<8, 256>>>(vec, 7 * D, 3u, 0.5f);
    tt_fill_h<<<256, 256>>>((unsigned short*)w, 3L * D * D, 4u, 0.06f);
    if (split) tt_fill_h<<<256, 256>>>((unsigned short*)a.wol, 3L * D * D, 5u, 3e-5f);
  }
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  for (int i = 0; i < 3; ++i) fd_tfmr_tail(a, 0);
  (void)hipEventRecord(t0, 0);
  const int iters = 50;
  for (int i = 0; i < iters; ++i) fd_tfmr_tail(a, 0);
  (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
  float ms; (void)hipEventElapsedTime(&ms, t0, t1);
  printf("tfmr_tail M=%d split=%d rows16=%d: %.2f us/launch\n", M, (int)split, a.rows16, ms / iters * 1e3);
#ifdef FD_PROF
  {
    const int nb = a.rows16 ? (M + 15) / 16 : (M + 31) / 32;
    std::vector<unsigned long long> h((size_t)nb * 16);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(fd_prof), h.size() * 8);
    const char* names[9] = {"", "inputs + constants -> LDS", "stage 1 MFMA (out_proj)", "residual + LayerNorm1", "x_a -> LDS + fragments", "stage 2 MFMA (FFN 1)", "relu -> LDS + fragments", "stage 3 MFMA (FFN 2)", "residual + LayerNorm2"};
    for (int k = 1; k < 9; ++k) {
      double s = 0;
      for (int b = 0; b < nb; ++b) s += (double)(h[(size_t)b * 16 + k] - h[(size_t)b * 16 + k - 1]);
      printf("  %-30s %8.0f cyc\n", names[k], s / nb);
    }
    if (a.rows16) {  // per-tile stamps of the last stage (wave 0): tile u end - stage start (stamp 6)
      for (int u = 0; u < 5; ++u) { double s = 0; for (int b = 0; b < nb; ++b) s += (double)(h[(size_t)b * 16 + 9 + u] - h[(size_t)b * 16 + 6]); printf("  stage 3, tile %d done at      %8.0f cyc\n", u, s / nb); }
    }
  }
#endif
  return 0;
}
