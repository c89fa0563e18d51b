// Micro-benchmark: how fast can all 256 CUs pull the SAME 512 KB weight stream out of L2 (the EdgeTransition pattern), and does it
// matter whether they walk it in the same phase?   hipcc --offload-arch=gfx950 -O3 -w wstream_bench.hip -o wstream_bench
//   MODE 0: LDS-DMA (global_load_lds_dwordx4), 64 KB chunks, one chunk in flight ahead
//   MODE 1: plain global_load_dwordx4 into registers at the same cadence
//   ROT  0: every block starts at chunk 0;  1: block b starts at chunk (b >> 3) & 7 (the 32 CUs of an XCD spread over the 8 chunks);
//        2: start offset (b >> 3) * 16 KB (32 different phases);  3: each XCD reads its own 512 KB copy, same phase;
//        4: every block cycles over its OWN 512 KB (128 MB in all: misses L2, fits the 256 MB MALL);  5: every block streams its own
//        passes x 512 KB once (2.6 GB: HBM)
//   MODE 2: global_store_dwordx4 of the same footprint (ROT 5: HBM writes);  MODE 3: load + store of the same address (read-modify-write stream)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
constexpr int REGION = 512 * 1024, CHUNK = 64 * 1024, NCH = REGION / CHUNK;

template <int MODE, int ROT>
__global__ __launch_bounds__(256, 1) void k(unsigned* out, char* gsrc, int passes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, b = blockIdx.x;
  size_t phase = 0;
  char* base = gsrc;
  if (ROT == 1) phase = (size_t)((b >> 3) & 7) * CHUNK;
  if (ROT == 2) phase = (size_t)((b >> 3) & 31) * 16384;
  if (ROT == 3) base = gsrc + (size_t)(b & 7) * REGION;
  if (ROT == 4) base = gsrc + (size_t)b * REGION;
  if (ROT == 5) base = gsrc + (size_t)b * REGION * passes;
  u32x4 sink = {0, 0, 0, 0};
  const int n = passes * NCH;
  for (int c = 0; c < n; ++c) {
    const size_t off = ROT == 5 ? (size_t)c * CHUNK : (((size_t)c * CHUNK + phase) & (REGION - 1));
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const size_t a = ROT == 5 ? off + (size_t)(u * 256 + tid) * 16 : ((off + (size_t)(u * 256 + tid) * 16) & (REGION - 1));
      if (MODE == 0) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)(smem + ((c & 1) << 16) + (u * 256 + (tid & ~63)) * 16));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(base + a) : "memory", "m0");
      } else if (MODE == 2) {
        const u32x4 t = {(unsigned)c, (unsigned)u, (unsigned)tid, 7u};
        *(u32x4*)(base + a) = t;
      } else if (MODE == 3) {
        u32x4 t = *(const u32x4*)(base + a);
        t[0] += 1u;
        *(u32x4*)(base + a + (size_t)256 * 20 * REGION) = t;
      } else {
        u32x4 t;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(base + a) : "memory");
        asm volatile("" : "+v"(t));
        if (u == 15) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); sink += t; }
      }
    }
    if (MODE == 0) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink[0] == 0x12345u) out[tid] = sink[1] + ((unsigned*)smem)[tid];
}
static char* g;
template <int MODE, int ROT> static void run(const char* name, unsigned* o, int nblk) {
  hipFuncSetAttribute((const void*)k<MODE, ROT>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  const int passes = 20;
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<MODE, ROT>), dim3(nblk), dim3(256), 131072, 0, o, g, passes);
  hipEventRecord(a, 0);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((k<MODE, ROT>), dim3(nblk), dim3(256), 131072, 0, o, g, passes);
  hipEventRecord(b, 0); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
  const double bytes = (double)nblk * passes * REGION;
  printf("%-44s %3d blocks  %7.3f ms  %6.2f TB/s  %5.1f B/ns/CU\n", name, nblk, ms, bytes / ms / 1e9, bytes / nblk / ms / 1e6);
}
int main() {
  unsigned* o; hipMalloc(&o, 4096); hipMalloc(&g, (size_t)2 * 256 * 20 * REGION); hipMemset(g, 1, (size_t)2 * 256 * 20 * REGION);
  for (int nblk : {256, 64, 8}) {
    run<0, 0>("LDS-DMA, same phase", o, nblk);
    run<0, 1>("LDS-DMA, 8 chunk phases per XCD", o, nblk);
    run<0, 2>("LDS-DMA, 32 phases of 16 KB per XCD", o, nblk);
    run<0, 3>("LDS-DMA, a copy per XCD, same phase", o, nblk);
    run<1, 0>("global_load_dwordx4, same phase", o, nblk);
    run<1, 2>("global_load_dwordx4, 32 phases", o, nblk);
    run<0, 4>("LDS-DMA, own 512 KB per block (MALL)", o, nblk);
    run<1, 4>("global_load_dwordx4, own 512 KB (MALL)", o, nblk);
    run<0, 5>("LDS-DMA, own 10 MB per block once (HBM)", o, nblk);
    run<1, 5>("global_load_dwordx4, own 10 MB once (HBM)", o, nblk);
    run<2, 5>("global_store_dwordx4, own 10 MB once (HBM)", o, nblk);
    run<3, 5>("load + store to a second buffer (HBM, bytes = read)", o, nblk);
  }
  return 0;
}
