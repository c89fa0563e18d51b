// Micro-benchmark: how fast can ONE 256-thread block per 128 rows pull a [M, 320] fp32 matrix through different
// per-lane access patterns?  (decides the global I/O layout of chain.hip)   hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int K = 320;

// (a) fragment pattern: lane = row (li) , 16 B at k = 16 s + 8 hi (+4): 64 different 128 B lines per instruction
__global__ __launch_bounds__(256) void pat_rowlane(const float* __restrict__ x, float* __restrict__ out, int M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  int row = blockIdx.x * 128 + wave * 32 + li; if (row >= M) row = M - 1;
  const float* xr = x + (long)row * K + 8 * hi;
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < K / 16; ++s) { acc += *(const f32x4*)(xr + 16 * s); acc += *(const f32x4*)(xr + 16 * s + 4); }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[tid] = 1.f;
}
// (b) row-major dword: register r = row, lanes 0..31 = 32 consecutive floats, hi picks another row
__global__ __launch_bounds__(256) void pat_dword(const float* __restrict__ x, float* __restrict__ out, int M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  float acc = 0.f;
#pragma unroll
  for (int T = 0; T < K / 32; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = blockIdx.x * 128 + wave * 32 + 4 * hi + (r & 3) + 8 * (r >> 2); if (row >= M) row = M - 1;
      acc += x[(long)row * K + 32 * T + li];
    }
  if (acc == 12345.f) out[tid] = 1.f;
}
// (c) fully coalesced f32x4: the wave's 32 rows are one linear 40 KB span
__global__ __launch_bounds__(256) void pat_linear(const float* __restrict__ x, float* __restrict__ out, int M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  long base = ((long)blockIdx.x * 128 + wave * 32) * K;
  const long lim = (long)M * K - 4;
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int it = 0; it < 32 * K / 256; ++it) { long o = base + it * 256 + lane * 4; if (o > lim) o = lim; acc += *(const f32x4*)(x + o); }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[tid] = 1.f;
}
// (d) half-row per lane-pair: lane li reads 2 x f32x4 contiguous (32 B) — same as (a); (e) 16 lanes per row f32x4 (256 B segments)
__global__ __launch_bounds__(256) void pat_seg256(const float* __restrict__ x, float* __restrict__ out, int M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int c = 0; c < K / 64; ++c)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      int row = blockIdx.x * 128 + wave * 32 + it * 4 + (lane >> 4); if (row >= M) row = M - 1;
      acc += *(const f32x4*)(x + (long)row * K + 64 * c + 4 * (lane & 15));
    }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) out[tid] = 1.f;
}
// (b2) row-major dword with the 16 row pointers hoisted (immediate offsets per tile)
__global__ __launch_bounds__(256) void pat_dword2(const float* __restrict__ x, float* __restrict__ out, int M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  const float* rp[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = blockIdx.x * 128 + wave * 32 + 4 * hi + (r & 3) + 8 * (r >> 2); if (row >= M) row = M - 1;
    rp[r] = x + (long)row * K + li;
  }
  float acc = 0.f;
#pragma unroll
  for (int T = 0; T < K / 32; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc += rp[r][32 * T];
  if (acc == 12345.f) out[tid] = 1.f;
}
// (s1) stores: fragment pattern f32x4 / (s2) row-major dword hoisted / (s3) 16 lanes per row f32x4
__global__ __launch_bounds__(256) void st_rowlane(float* __restrict__ x, int M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  int row = blockIdx.x * 128 + wave * 32 + li; if (row >= M) row = M - 1;
  float* xr = x + (long)row * K + 4 * hi;
  const f32x4 v = {1.f, 2.f, 3.f, (float)tid};
#pragma unroll
  for (int s = 0; s < K / 8; ++s) *(f32x4*)(xr + 8 * s) = v;
}
__global__ __launch_bounds__(256) void st_dword2(float* __restrict__ x, int M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  float* rp[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    int row = blockIdx.x * 128 + wave * 32 + 4 * hi + (r & 3) + 8 * (r >> 2); if (row >= M) row = M - 1;
    rp[r] = x + (long)row * K + li;
  }
#pragma unroll
  for (int T = 0; T < K / 32; ++T)
#pragma unroll
    for (int r = 0; r < 16; ++r) rp[r][32 * T] = (float)tid;
}
__global__ __launch_bounds__(256) void st_seg256(float* __restrict__ x, int M) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const f32x4 v = {1.f, 2.f, 3.f, (float)tid};
#pragma unroll
  for (int c = 0; c < K / 64; ++c)
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      int row = blockIdx.x * 128 + wave * 32 + it * 4 + (lane >> 4); if (row >= M) row = M - 1;
      *(f32x4*)(x + (long)row * K + 64 * c + 4 * (lane & 15)) = v;
    }
}
__global__ void empty_kernel(float* out) { if (threadIdx.x == 9999) out[0] = 1.f; }

template <class F> static float timeit(F f, int iters) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) f();
  hipEventRecord(a, 0);
  for (int i = 0; i < iters; ++i) f();
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / iters;
}
int main() {
  const int M = 2400; float *x, *o;
  hipMalloc(&x, (size_t)M * K * 4); hipMalloc(&o, 4096); hipMemset(x, 0, (size_t)M * K * 4);
  const int G = (M + 127) / 128;
  printf("empty      %.2f us\n", timeit([&] { hipLaunchKernelGGL(empty_kernel, dim3(G), dim3(256), 0, 0, o); }, 200));
  printf("rowlane    %.2f us\n", timeit([&] { hipLaunchKernelGGL(pat_rowlane, dim3(G), dim3(256), 0, 0, x, o, M); }, 200));
  printf("dword      %.2f us\n", timeit([&] { hipLaunchKernelGGL(pat_dword, dim3(G), dim3(256), 0, 0, x, o, M); }, 200));
  printf("linear     %.2f us\n", timeit([&] { hipLaunchKernelGGL(pat_linear, dim3(G), dim3(256), 0, 0, x, o, M); }, 200));
  printf("seg256     %.2f us\n", timeit([&] { hipLaunchKernelGGL(pat_seg256, dim3(G), dim3(256), 0, 0, x, o, M); }, 200));
  printf("dword2     %.2f us\n", timeit([&] { hipLaunchKernelGGL(pat_dword2, dim3(G), dim3(256), 0, 0, x, o, M); }, 200));
  printf("st_rowlane %.2f us\n", timeit([&] { hipLaunchKernelGGL(st_rowlane, dim3(G), dim3(256), 0, 0, x, M); }, 200));
  printf("st_dword2  %.2f us\n", timeit([&] { hipLaunchKernelGGL(st_dword2, dim3(G), dim3(256), 0, 0, x, M); }, 200));
  printf("st_seg256  %.2f us\n", timeit([&] { hipLaunchKernelGGL(st_seg256, dim3(G), dim3(256), 0, 0, x, M); }, 200));
  return 0;
}
