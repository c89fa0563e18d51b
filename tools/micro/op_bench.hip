// Stand-alone timing + phase profile (-DFD_PROF) of outproj_split_kernel (IPA output projection, split operands), M = 2400, K = 2688.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w [-DFD_PROF] op_bench.hip -o op_bench ; op_bench [cold]
// cold = 1: 400 MB of unrelated traffic between the launches (the weights and activations are not L2-resident, as in the forward)
#include "../../framedipt_amd/csrc/gemm.hip"
#include <cstdio>
#include <vector>
__global__ void fill_f32(float* p, long n, unsigned seed, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((h & 0xFFFF) / 65536.f - 0.5f) * 2.f * scale;
  }
}
__global__ void chain_image_kernel2(const float* __restrict__ w, int N, int K, int lo, half_t* __restrict__ img) {
  const int NT = N / 32, KS = K / 16;
  const long n = (long)NT * KS * 64 * 8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long ts = i >> 9;
    const int s = (int)(ts % KS), T = (int)(ts / KS);
    const float v = w[(long)(32 * T + (lane & 31)) * K + 16 * s + 8 * (lane >> 5) + e];
    img[i] = lo ? f2h(v - h2f(f2h(v))) : f2h(v);
  }
}
int main(int argc, char** argv) {
  const int cold = argc > 1 ? atoi(argv[1]) : 0;
  const int M = 2400, N = 256, K = 2688;
  float *A, *W, *bias, *rm, *parts, *junk; void *wh, *wl;
  const size_t JN = (size_t)100 << 20;
  (void)hipMalloc(&A, (size_t)M * K * 4); (void)hipMalloc(&W, (size_t)N * K * 4); (void)hipMalloc(&bias, N * 4); (void)hipMalloc(&rm, M * 4);
  (void)hipMalloc(&parts, (size_t)8 * M * N * 4); (void)hipMalloc(&wh, (size_t)N * K * 2); (void)hipMalloc(&wl, (size_t)N * K * 2); (void)hipMalloc(&junk, JN * 4);
  fill_f32<<<256, 256>>>(A, (long)M * K, 1u, 1.f); fill_f32<<<256, 256>>>(W, (long)N * K, 2u, 0.05f); fill_f32<<<1, 256>>>(bias, N, 3u, 0.1f);
  fill_f32<<<16, 256>>>(rm, M, 4u, 0.f); (void)hipMemset(rm, 0, M * 4);
  { std::vector<float> o(M, 1.f); (void)hipMemcpy(rm, o.data(), M * 4, hipMemcpyHostToDevice); }
  chain_image_kernel2<<<256, 256>>>(W, N, K, 0, (half_t*)wh); chain_image_kernel2<<<256, 256>>>(W, N, K, 1, (half_t*)wl);
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  float tot = 0; const int iters = 20;
  for (int i = 0; i < iters + 3; ++i) {
    if (cold) fill_f32<<<2048, 256>>>(junk, (long)JN, 7u + i, 1.f);
    (void)hipEventRecord(t0, 0);
    if (fd_outproj_split(M, N, K, A, K, wh, wl, bias, rm, parts, (long)M * N, N, 0)) { printf("launch failed\n"); return 1; }
    (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
    float ms; (void)hipEventElapsedTime(&ms, t0, t1);
    if (i >= 3) tot += ms;
  }
  printf("outproj_split M=%d (%s): %.2f us/launch\n", M, cold ? "cold" : "warm", tot / iters * 1e3);
  {  // check against fp64 on a few entries
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hp((size_t)6 * M * N), hb(N);
    (void)hipMemcpy(hA.data(), A, hA.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hW.data(), W, hW.size() * 4, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hp.data(), parts, hp.size() * 4, hipMemcpyDeviceToHost); (void)hipMemcpy(hb.data(), bias, N * 4, hipMemcpyDeviceToHost);
    double me = 0;
    for (int m : {0, 31, 32, 63, 64, 1000, 2367, 2368, 2399})
      for (int n : {0, 1, 31, 32, 100, 255}) {
        double ref = hb[n], got = 0;
        for (int k = 0; k < K; ++k) ref += (double)hA[(size_t)m * K + k] * hW[(size_t)n * K + k];
        for (int z = 0; z < 6; ++z) got += hp[((size_t)z * M + m) * N + n];
        me = fmax(me, fabs(ref - got));
      }
    printf("  max |err| vs fp64 on 54 entries: %.3g\n", me);
  }
#ifdef FD_PROF
  {
    std::vector<unsigned long long> h((size_t)8192 * 16);
    (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(fd_prof), h.size() * 8);
    const char* names[8] = {"", "first W requests", "A requests issued", "A arrived, split, stored", "barrier", "main loop", "tile -> LDS (2 barriers)", "row stores issued"};
    double sum[8] = {0}; int nb = 38 * 6;
    for (int b = 0; b < nb; ++b) for (int k = 1; k < 8; ++k) sum[k] += (double)(h[(size_t)b * 16 + k] - h[(size_t)b * 16 + k - 1]);
    double tot2 = 0;
    for (int k = 1; k < 8; ++k) { printf("  %-28s %8.0f cyc\n", names[k], sum[k] / nb); tot2 += sum[k] / nb; }
    printf("  %-28s %8.0f cyc per block (s_memtime ticks)\n", "total", tot2);
    unsigned long long lo = ~0ull, hi = 0;
    for (int b = 0; b < nb; ++b) { lo = h[(size_t)b * 16] < lo ? h[(size_t)b * 16] : lo; hi = h[(size_t)b * 16 + 7] > hi ? h[(size_t)b * 16 + 7] : hi; }
    printf("  first start -> last end: %llu ticks\n", hi - lo);
  }
#endif
  return 0;
}
