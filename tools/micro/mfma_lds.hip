// Micro-benchmark: MFMA 32x32x16 bf16 issue rate of one wave per SIMD (256-thread block, 1 block/CU) when the A operand
// comes from registers vs. one ds_read_b128 per MFMA vs. one per two MFMAs.   hipcc --offload-arch=gfx950 -O3 -w
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 hx8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
constexpr int ITERS = 256, KS = 16;

template <int MODE, int NACC>
__global__ __launch_bounds__(256, 1) void k(float* out, const char* gsrc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 65536 / 4; i += 256) ((float*)smem)[i] = 0.f;
  __syncthreads();
  f32x16 acc[NACC];
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  hx8 B[KS];
  for (int s = 0; s < KS; ++s) for (int e = 0; e < 8; ++e) B[s][e] = (fd_h)(float)(tid + s + e);
  hx8 A0 = B[0];
  u16x8 stg[16];
  for (int u = 0; u < 16; ++u) stg[u] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
  for (int it = 0; it < ITERS; ++it) {
    const char* p = smem + ((it & 1) << 15) + lane * 16;
    if (MODE == 4 && (it & 1) == 0) {  // 64 KB of LDS-DMA per 64 MFMAs (two iterations), like one ET2 chunk
#pragma unroll
      for (int u = 0; u < 16; ++u) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)(smem + 65536 + (u * 256 + (tid & ~63)) * 16));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(gsrc + (size_t)(u * 256 + tid) * 16) : "memory", "m0");
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      hx8 a;
      if (MODE == 8) {  // same 64 KB per 64 MFMAs as 4 B-per-lane DMAs: 2 per MFMA pair... 64 per 64 MFMAs x 4 (256 B each)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
          const int u = (((it & 1) << 4) + s) * 4 + h;  // 0..127 per two iterations -> 128 x 256 B x 4 waves... = 128 KB? no: 256 B per wave-instr
          const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)(smem + 65536 + ((u & 63) * 1024 + (tid >> 6) * 256)));
          if (h < 2 || true) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" : : "s"(m0v), "v"(gsrc + (size_t)(u & 63) * 1024 + tid * 4) : "m0");
        }
      }
      if (MODE == 9 && (s & 1) == 0) {  // plain global_load_dwordx4 (to VGPRs, never used) at the DMA cadence: is it VMEM issue as such?
        const int u = ((it & 1) << 3) + (s >> 1);
        u16x8 t;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(gsrc + (size_t)(u * 256 + tid) * 16));
        asm volatile("" :: "v"(t));
      }
      if (MODE == 7) {  // same 64 KB per 64 MFMAs through registers: global_load_dwordx4 ... ds_write_b128 one iteration later
        if ((it & 1) == 0) stg[s] = *(const u16x8*)(gsrc + (size_t)(((it >> 1) & 7) * 65536) + (size_t)(s * 256 + tid) * 16);
        else *(u16x8*)(smem + 65536 + (s * 256 + tid) * 16) = stg[s];
      }
      if (MODE == 6 && (s & 1) == 0) {  // same 64 KB per 64 MFMAs, but ONE DMA instruction every 4 MFMAs
        const int u = ((it & 1) << 3) + (s >> 1);
        const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)(smem + 65536 + (u * 256 + (tid & ~63)) * 16));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(gsrc + (size_t)(u * 256 + tid) * 16) : "m0");
      }
      if (MODE == 0) a = A0;
      else if (MODE == 3 || MODE == 4 || MODE == 6 || MODE == 7 || MODE == 8 || MODE == 9) {  // ET2 layer-1 slab pattern: row li (512 B rows), 16 B unit (2s+hi) ^ (li & 15)
        const int li = lane & 31, hi = lane >> 5;
        a = __builtin_bit_cast(hx8, *(const u16x8*)(smem + ((it & 1) << 15) + li * 512 + ((((2 * s + hi) & 31) ^ (li & 15)) << 4)));
      } else if (MODE == 5) {  // two reads per MFMA (LDS headroom probe)
        a = __builtin_bit_cast(hx8, *(const u16x8*)(p + s * 1024));
        const hx8 a2 = __builtin_bit_cast(hx8, *(const u16x8*)(p + ((s + 7) & 15) * 1024 + 16384));
        a[0] += a2[0];
      } else a = __builtin_bit_cast(hx8, *(const u16x8*)(p + (MODE == 2 ? (s >> 1) : s) * 1024));
#pragma unroll
      for (int q = 0; q < NACC; ++q) acc[q] = fd_mfma32(a, B[(s + q) % KS], acc[q]);
    }
  }
  float t = 0.f;
  for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) t += acc[a][r];
  if (t == 12345.f) out[tid] = t;
}
static char* g;
template <int MODE, int NACC> static void run(const char* name, float* o) {
  hipFuncSetAttribute((const void*)k<MODE, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<MODE, NACC>), dim3(256), dim3(256), 131072, 0, o, g);
  hipEventRecord(a, 0);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<MODE, NACC>), dim3(256), dim3(256), 131072, 0, o, g);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  const double n = (double)ITERS * KS * NACC;  // MFMAs per wave per launch
  const double ns = ms * 1e6 / 20 / n;
  printf("%-28s %6.2f ns/MFMA  -> %6.1f TFLOP/s chip\n", name, ns, 256 * 4 * 32768.0 / ns / 1e3);
}
int main() {
  float* o; hipMalloc(&o, 4096); hipMalloc(&g, 1 << 20); hipMemset(g, 0, 1 << 20);
  run<0, 2>("regs, 2 acc", o);
  run<0, 4>("regs, 4 acc", o);
  run<1, 2>("1 ds_read / 2 MFMA (2 acc)", o);
  run<1, 1>("1 ds_read / 1 MFMA (1 acc)", o);
  run<1, 4>("1 ds_read / 4 MFMA (4 acc)", o);
  run<3, 2>("ET2 swizzle, 2 acc", o);
  run<4, 2>("ET2 swizzle + 64KB DMA/64", o);
  run<6, 2>("ET2 swizzle + DMA spread 1/4", o);
  run<9, 2>("ET2 swizzle + plain loads 1/4", o);
  run<5, 2>("2 ds_read / MFMA (2 acc)", o);
  run<5, 1>("2 ds_read / MFMA (1 acc)", o);
  return 0;
}
