// Stand-alone timing + in-kernel phase profile of edge_transition4_flat_kernel (build with -DE4_PROF for the profile).  Round 6: the chunk-synchronous
// predecessor is gone from the library; the two "variants" below are two runs of the flat kernel (run-to-run bit identity), variant 1 with the pair_z emission.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -w [-DFD_PROF] [-DE4_ABL=k] tools/micro/et4_bench.hip -o et4_bench
#define E4_KEEP_CHUNK
#include "../../framedipt_amd/csrc/edge_transition4.hip"
#include <cstdio>
#include <vector>
int fd_edge_transition4_variant(const ET2Args& a, hipStream_t st, int flat);
__global__ void fill_f32(float* p, long n, unsigned seed, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = ((h & 0xFFFF) / 65536.f - 0.5f) * 2.f * scale;
  }
}
__global__ void fill_h16(half_t* p, long n, unsigned seed, float scale) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    unsigned h = (unsigned)i * 2654435761u + seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = f2h(((h & 0xFFFF) / 65536.f - 0.5f) * 2.f * scale);
  }
}
__global__ void diff_count(const unsigned* x, const unsigned* y, long n, unsigned long long* out) {
  unsigned long long c = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) c += x[i] != y[i];
  if (c) atomicAdd(out, c);
}
__global__ void checksum_k(const unsigned* x, long n, unsigned long long* out) {
  unsigned long long c = 0;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) c += (unsigned long long)x[i] * (2654435761ull * (unsigned long long)i + 1ull);
  atomicAdd(out, c);
}
int main(int argc, char** argv) {
  const int B = 8, N = argc > 1 ? atoi(argv[1]) : 300;
  const int only = argc > 3 ? atoi(argv[3]) : -1;  // time only this variant (0: pair bias only, 1: + pair_z of the next block)
  const float ds = argc > 4 ? atof(argv[4]) : 1.f;  // data scale (0: all-zero operands, the low-power case)
  const long P = (long)B * N * N, R = (long)B * N;
  half_t *z, *zo[2]; float *b2, *g, *bt, *rm, *w1, *w2, *wf, *wb, *rows, *bo[2]; void* stream;
  (void)hipMalloc(&z, P * 128 * 2); (void)hipMalloc(&zo[0], P * 128 * 2); (void)hipMalloc(&zo[1], P * 128 * 2);
  (void)hipMalloc(&b2, 384 * 4); (void)hipMalloc(&g, 128 * 4); (void)hipMalloc(&bt, 128 * 4); (void)hipMalloc(&rm, R * 4);
  (void)hipMalloc(&w1, 384 * 384 * 4); (void)hipMalloc(&w2, 384 * 384 * 4); (void)hipMalloc(&wf, 128 * 384 * 4); (void)hipMalloc(&wb, 8 * 128 * 4);
  (void)hipMalloc(&rows, R * 1024 * 4);
  (void)hipMalloc(&stream, fd_et4_stream_bytes());
  fill_h16<<<1024, 256>>>(z, P * 128, 1u, 1.0f * ds);
  fill_f32<<<64, 256>>>(b2, 384, 2u, 0.1f); fill_f32<<<64, 256>>>(g, 128, 3u, 1.0f); fill_f32<<<64, 256>>>(bt, 128, 4u, 0.2f);
  fill_f32<<<64, 256>>>(w1, 384 * 384, 5u, 0.08f * ds); fill_f32<<<64, 256>>>(w2, 384 * 384, 6u, 0.08f * ds); fill_f32<<<64, 256>>>(wf, 128 * 384, 7u, 0.08f * ds);
  fill_f32<<<64, 256>>>(wb, 8 * 128, 8u, 0.1f); fill_f32<<<256, 256>>>(rows, R * 1024, 9u, 0.5f * ds);
  { std::vector<float> m(R, 1.f); for (long i = 0; i < R; i += 37) m[i] = 0.f; (void)hipMemcpy(rm, m.data(), R * 4, hipMemcpyHostToDevice); }
  fd_et4_build_stream(w1, w2, wf, stream, 0);
  ET2Args a; a.B = B; a.N = N; a.z_in = z; a.e = nullptr; a.a1 = nullptr; a.af = nullptr; a.stream = stream; a.b2 = b2; a.gamma = g;
  a.beta = bt; a.res_mask = rm; a.trace = nullptr; a.reserve_cus = 0;
  { char* ai; const size_t na = fd_et4_a_image_bytes(B, N), nb = fd_et4_b_image_bytes(B, N);  // one allocation: 32-bit offsets between the images
    (void)hipMalloc(&ai, na + nb); a.a1_img = ai; a.b1_img = ai + na; a.e_h16 = nullptr;
    fd_et4_row_images(rows, B, N, ai, ai + na, 0); }
  const long Np = (N + 31) / 32 * 32;
  {  // next block's pair bias from the epilogue (argv[2] = 0 disables)
    void* wimg;
    (void)hipMalloc(&wimg, 8192); fd_et4_build_bias_image(wb, 8, 0.57735f, wimg, 0);
    (void)hipMalloc(&bo[0], (size_t)B * 8 * Np * Np * 4); (void)hipMalloc(&bo[1], (size_t)B * 8 * Np * Np * 4);
    (void)hipMemset(bo[0], 0, (size_t)B * 8 * Np * Np * 4); (void)hipMemset(bo[1], 0, (size_t)B * 8 * Np * Np * 4);
    a.wb_img = (argc > 2 && atoi(argv[2]) == 0) ? nullptr : wimg; a.bb = b2; a.H = 8;
  }
  half_t* pz; float* bdz;
  (void)hipMalloc(&pz, fd_pz_bytes(B, N)); (void)hipMalloc(&bdz, 128); (void)hipMemset(bdz, 0, 128);
  (void)hipMemset(pz, 0, fd_pz_bytes(B, N));
  {  // down_z of the "next block": random hi / lo fragment images into the stream's last chunk (what fd_et4_set_dz copies there)
    half_t* dzi; (void)hipMalloc(&dzi, 16384);
    fill_h16<<<64, 256>>>(dzi, 4096, 11u, 0.1f * ds); fill_h16<<<64, 256>>>(dzi + 4096, 4096, 12u, 0.0001f * ds);
    (void)hipMemcpy((char*)stream + (size_t)E4_DZ_FR0 * 1024, dzi, 16384, hipMemcpyDeviceToDevice);
    fill_f32<<<1, 64>>>(bdz, 32, 13u, 0.1f);
  }
  hipEvent_t t0, t1; (void)hipEventCreate(&t0); (void)hipEventCreate(&t1);
  const double flops = 688128.0 * P;  // reference-formulation count (bench.py: ET_FLOPS_PER_PAIR)
  for (int v = 0; v < 2; ++v) {
    if (only >= 0 && v != only) continue;
    a.z_out = zo[v]; a.bias_out = bo[v];
    a.pz_out = (v == 1 && a.wb_img) ? pz : nullptr; a.bdz = bdz;  // (down_z itself: zeros in the stream's last chunk)
    for (int i = 0; i < 3; ++i) fd_edge_transition4_variant(a, 0, v);
    (void)hipEventRecord(t0, 0);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) fd_edge_transition4_variant(a, 0, v);
    (void)hipEventRecord(t1, 0); (void)hipEventSynchronize(t1);
    float ms; (void)hipEventElapsedTime(&ms, t0, t1);
    printf("ET4 %s N=%d: %.3f ms/launch, %.1f TFLOP/s (%.1f%% of 2500)\n", v ? "flat + pair_z" : "flat", N, ms / iters, flops / (ms / iters) / 1e9, flops / (ms / iters) / 1e9 / 25.0);
  }
  if (only < 0) {
    unsigned long long* d; (void)hipMalloc(&d, 16); (void)hipMemset(d, 0, 16);
    diff_count<<<1024, 256>>>((const unsigned*)zo[0], (const unsigned*)zo[1], P * 64, d);
    diff_count<<<1024, 256>>>((const unsigned*)bo[0], (const unsigned*)bo[1], (long)B * 8 * Np * Np, d + 1);
    unsigned long long h[2]; (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    std::vector<half_t> hz(4096); (void)hipMemcpy(hz.data(), zo[0] + (P / 2) * 128, 8192, hipMemcpyDeviceToHost);
    double sa = 0; for (auto x : hz) sa += fabs((double)(float)__builtin_bit_cast(_Float16, x));
    {  // checksums of every output of variant 1: equal between two builds of this file = the builds give the same bits
      unsigned long long* c3; (void)hipMalloc(&c3, 24); (void)hipMemset(c3, 0, 24);
      checksum_k<<<1024, 256>>>((const unsigned*)zo[1], P * 64, c3);
      checksum_k<<<1024, 256>>>((const unsigned*)bo[1], (long)B * 8 * Np * Np, c3 + 1);
      checksum_k<<<1024, 256>>>((const unsigned*)pz, (long)(fd_pz_bytes(B, N) / 4), c3 + 2);
      unsigned long long hc[3]; (void)hipMemcpy(hc, c3, 24, hipMemcpyDeviceToHost);
      printf("checksums z' %016llx bias %016llx pair_z %016llx\n", hc[0], hc[1], hc[2]);
    }
    printf("with vs without pair_z: %llu differing z words of %ld, %llu differing bias words; mean |z'| %.4f\n", h[0], P * 64, h[1], sa / 4096);
  }
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
