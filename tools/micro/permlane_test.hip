// v_permlane32_swap semantics check (gfx950): prints what each lane holds after swap(a = lane, b = 100 + lane)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = 100 + threadIdx.x;
  u2 r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
  float x = (float)threadIdx.x;
  const unsigned xu = __builtin_bit_cast(unsigned, x);
  const u2 q = __builtin_amdgcn_permlane32_swap(xu, 0u, false, false), w = __builtin_amdgcn_permlane32_swap(0u, xu, false, false);
  o[128 + threadIdx.x] = (unsigned)((__builtin_bit_cast(float, q[0]) + __builtin_bit_cast(float, q[1])) + (__builtin_bit_cast(float, w[0]) + __builtin_bit_cast(float, w[1])));
}
int main() {
  unsigned* d; (void)hipMalloc(&d, 192 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[192]; (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  for (int i : {0, 1, 31, 32, 33, 63}) printf("lane %2d: r0 %3u r1 %3u sum %3u\n", i, h[i], h[64 + i], h[128 + i]);
  return 0;
}
