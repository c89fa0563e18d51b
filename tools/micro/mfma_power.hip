// Micro-benchmark: what does a PURE fp16 MFMA loop sustain on all 256 CUs under the socket power cap, as a function of the operand
// bits?  No memory traffic inside the timed loop: the operand fragments sit in registers, four (or eight) independent accumulators per
// wave, two waves per SIMD.  Reports TFLOP/s, the shader clock inside the kernel (s_memtime cycles / s_memrealtime 100 MHz ticks) and
// the matrix-core utilisation at that clock.  This is the ceiling EdgeTransition's "power-limited" reading is priced against.
//   hipcc --offload-arch=gfx950 -O3 -w mfma_power.hip -o mfma_power && ./mfma_power [launch_us=300]
// operand patterns: 0 zeros | 1 random bits (finite halfs) | 2 N(0,1) activations | 3 post-ReLU activations (half of them zero) x N(0,0.05) weights
//                   | 4 N(0,1) x N(0,0.05) weights (what EdgeTransition's layers multiply)
// LDS variant (LDSOP = 1): the A fragment of every MFMA is re-read from LDS (ds_read_b128, conflict-free), the pattern of a kernel
// whose weights stream through LDS: adds the LDS read energy without adding global traffic.
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 hx8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int LDSOP>
__global__ __launch_bounds__(512, 1) void power_kernel(const u32x4* __restrict__ ops, float* __restrict__ out, int iters,
                                                       unsigned long long* __restrict__ clk) {
  __shared__ u32x4 lds[LDSOP ? 8 * 4 * 64 : 1];  // per wave 4 fragments
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  hx8 A[4], B[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    A[i] = __builtin_bit_cast(hx8, ops[((blockIdx.x * 8 + wave) % 64 * 8 + i) * 64 + lane]);
    B[i] = __builtin_bit_cast(hx8, ops[(((blockIdx.x * 8 + wave) % 64) * 8 + 4 + i) * 64 + lane]);
    if (LDSOP) lds[(wave * 4 + i) * 64 + lane] = __builtin_bit_cast(u32x4, A[i]);
  }
  __syncthreads();
  f32x16 acc[NACC];
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      hx8 a = A[u & 3];
      if (LDSOP) a = __builtin_bit_cast(hx8, lds[(wave * 4 + (u & 3)) * 64 + lane]);
      acc[u % NACC] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, B[(u >> 2) & 3], acc[u % NACC], 0, 0, 0);
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < NACC; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 1.2345e-30f) out[tid] = s;
  if (tid == 0) {
    clk[blockIdx.x * 2] = c1 - c0;
    clk[blockIdx.x * 2 + 1] = r1 - r0;
  }
}

static float gauss(unsigned& st) {
  float s = 0.f;
  for (int i = 0; i < 12; ++i) { st = st * 1664525u + 1013904223u; s += (st >> 8) * (1.0f / 16777216.0f); }
  return s - 6.0f;
}

template <int NACC, int LDSOP>
static void run(const char* name, int pattern, int nblk, float target_us, u32x4* d_ops, float* d_out, unsigned long long* d_clk) {
  // fragments: per wave slot 8 fragments (4 A, 4 B) x 64 lanes x 8 halfs, 64 distinct wave slots
  const size_t n = (size_t)64 * 8 * 64 * 8;
  std::vector<_Float16> h(n);
  unsigned st = 12345u + pattern;
  for (size_t i = 0; i < n; ++i) {
    const bool is_b = ((i / 512) % 8) >= 4;  // B fragments play the activations, A the weights
    float v = 0.f;
    switch (pattern) {
      case 0: v = 0.f; break;
      case 1: { st = st * 1664525u + 1013904223u; unsigned short bits = (unsigned short)(st >> 16); if ((bits & 0x7c00) == 0x7c00) bits &= ~0x0400; _Float16 f; __builtin_memcpy(&f, &bits, 2); h[i] = f; continue; }
      case 2: v = gauss(st); break;
      case 3: v = is_b ? fmaxf(gauss(st), 0.f) : 0.05f * gauss(st); break;
      case 4: v = is_b ? gauss(st) : 0.05f * gauss(st); break;
    }
    h[i] = (_Float16)v;
  }
  hipMemcpy(d_ops, h.data(), n * 2, hipMemcpyHostToDevice);
  // calibrate the iteration count for the target launch duration at 2.0 GHz: 16 MFMAs x 32 cycles per iteration per wave, two waves per SIMD
  int iters = (int)(target_us * 1e-6 * 2.0e9 / (16.0 * 32.0 * 2.0));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  const int reps = 40;
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((power_kernel<NACC, LDSOP>), dim3(nblk), dim3(512), 0, 0, d_ops, d_out, iters, d_clk);
  hipEventRecord(a, 0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((power_kernel<NACC, LDSOP>), dim3(nblk), dim3(512), 0, 0, d_ops, d_out, iters, d_clk);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= reps;
  std::vector<unsigned long long> c(nblk * 2);
  hipMemcpy(c.data(), d_clk, nblk * 16, hipMemcpyDeviceToHost);
  double cyc = 0, ticks = 0;
  for (int i = 0; i < nblk; ++i) { cyc += c[2 * i]; ticks += c[2 * i + 1]; }
  const double ghz = cyc / (ticks * 10.0);  // 100 MHz ticks = 10 ns
  const double flops = (double)nblk * 8 * iters * 16 * 2.0 * 32 * 32 * 16;
  const double tf = flops / (ms * 1e-3) / 1e12;
  const double mfma_cycles_per_simd = (double)iters * 16 * 32 * 2;  // two waves per SIMD
  const double util = mfma_cycles_per_simd / (cyc / nblk);
  printf("%-58s %3d blocks %7.1f us  %7.1f TF/s  %.3f of 2500  clock %.3f GHz  matrix-core busy %.3f (in-loop)\n", name, nblk, ms * 1e3, tf,
         tf / 2500.0, ghz, util);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const float us = argc > 1 ? atof(argv[1]) : 300.f;
  u32x4* d_ops; float* d_out; unsigned long long* d_clk;
  hipMalloc(&d_ops, (size_t)64 * 8 * 64 * 16); hipMalloc(&d_out, 4096); hipMalloc(&d_clk, 256 * 16);
  const char* pat[5] = {"zeros", "random bits", "N(0,1) x N(0,1)", "ReLU(N(0,1)) x N(0,0.05)", "N(0,1) x N(0,0.05)"};
  char name[128];
  for (int rep = 0; rep < 2; ++rep) {
    for (int p = 0; p < 5; ++p) {
      snprintf(name, sizeof name, "registers, 4 accumulators: %s", pat[p]);
      run<4, 0>(name, p, 256, us, d_ops, d_out, d_clk);
    }
    for (int p = 0; p < 5; ++p) {
      snprintf(name, sizeof name, "A fragment from LDS per MFMA: %s", pat[p]);
      run<4, 1>(name, p, 256, us, d_ops, d_out, d_clk);
    }
  }
  // back-to-back long run (10 x the launch length): does the sustained clock keep falling?
  for (int p : {1, 4}) {
    snprintf(name, sizeof name, "registers, 10x longer launches: %s", pat[p]);
    run<4, 0>(name, p, 256, us * 10, d_ops, d_out, d_clk);
  }
  for (int p : {1, 4}) {
    snprintf(name, sizeof name, "registers, 128 of 256 CUs: %s", pat[p]);
    run<4, 0>(name, p, 128, us, d_ops, d_out, d_clk);
  }
  return 0;
}
