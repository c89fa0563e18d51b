// Probe the operand / result layout of v_mfma_f32_16x16x32_bf16 on gfx950 (assumed by edge_transition3.hip):
//   A (16 x 32): lane l holds row l & 15, k = 8 (l >> 4) + e;  B (32 x 16): lane l holds column l & 15, k = 8 (l >> 4) + e
//   D (16 x 16): lane l, register r holds D[4 (l >> 4) + r][l & 15]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 hx8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(const float* A, const float* B, float* D) {  // A [16][32], B [32][16], D [16][16] row-major
  const int l = threadIdx.x;
  hx8 a, b;
  for (int e = 0; e < 8; ++e) {
    a[e] = (fd_h)A[(l & 15) * 32 + 8 * (l >> 4) + e];
    b[e] = (fd_h)B[(8 * (l >> 4) + e) * 16 + (l & 15)];
  }
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = fd_mfma16(a, b, c);
  for (int r = 0; r < 4; ++r) D[(4 * (l >> 4) + r) * 16 + (l & 15)] = c[r];
}
int main() {
  float hA[512], hB[512], hD[256], ref[256];
  for (int i = 0; i < 512; ++i) { hA[i] = (float)((i * 7) % 13 - 6); hB[i] = (float)((i * 5) % 11 - 5); }
  for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int kk = 0; kk < 32; ++kk) s += hA[m * 32 + kk] * hB[kk * 16 + n]; ref[m * 16 + n] = s; }
  float *dA, *dB, *dD; hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dD, 1024);
  hipMemcpy(dA, hA, 2048, hipMemcpyHostToDevice); hipMemcpy(dB, hB, 2048, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
  hipMemcpy(hD, dD, 1024, hipMemcpyDeviceToHost);
  double err = 0; for (int i = 0; i < 256; ++i) err = fmax(err, fabs(hD[i] - ref[i]));
  printf("mfma_f32_16x16x32_bf16 layout check: max |err| = %g  (%s)\n", err, err == 0 ? "OK" : "MISMATCH");
  return err == 0 ? 0 : 1;
}
