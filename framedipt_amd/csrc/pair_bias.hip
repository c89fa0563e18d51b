// pair_bias.hip — pair bias of the IPA attention as its own pass over the half-precision pair representation:
// linear_b(z)/sqrt(3) (framedipt/model/ipa_pytorch.py:247,256-257).  The default path gets the bias from the epilogue of
// the kernel that produces z (edge embedder, EdgeTransition); this pass serves the layouts those epilogues do not cover
// (N > 512, N % 4 != 0 first block, non-default widths of the attention).
#include "common.hpp"
#include "kernels.hpp"

// ------------------------------------------------------------------ pair bias
// bias[b,h,i,j] = sqrt(1/3) (Wb z[b,i,j,:] + bb)_h : lane = pair, heads = MFMA rows (8 of 32 used), z fragments loaded
// straight from HBM in B-operand layout (no LDS), Wb fragments in registers.  HBM-bound: one pass over z.
__global__ __launch_bounds__(FD_THREADS) void pair_bias2_kernel(int B, int N, int H, const half_t* __restrict__ z,
                                                                const half_t* __restrict__ wb /* [H,128] pre-scaled */,
                                                                const float* __restrict__ bb, float* __restrict__ out,
                                                                int frag /* 1: fd_bias_frag_off order (attention3) */) {
  const int lane = threadIdx.x & 63, hi = lane >> 5, li = lane & 31;
  const long NN = (long)N * N, n_pairs = (long)B * NN;
  hx8 Wf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    u16x8 w = {0, 0, 0, 0, 0, 0, 0, 0};
    if (li < H) w = *(const u16x8*)(wb + li * 128 + 16 * s + 8 * hi);
    Wf[s] = __builtin_bit_cast(hx8, w);
  }
  const long n_tiles = (n_pairs + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += (long)gridDim.x * 4) {
    const long p_raw = tile * 32 + li;
    const long p = p_raw < n_pairs ? p_raw : n_pairs - 1;
    const half_t* zr = z + p * 128 + 8 * hi;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    hx8 zf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) zf[s] = __builtin_bit_cast(hx8, *(const u16x8*)(zr + 16 * s));
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = fd_mfma32(Wf[s], zf[s], acc);
    if (p_raw < n_pairs) {
      const long bidx = p / NN, ij = p - bidx * NN;
      const int i = (int)(ij / N), j = (int)(ij - (long)i * N), nt = (N + 31) >> 5;
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // rows 4hi + q (r = q): heads 0..7
        const int hh = 4 * hi + q;
        if (hh < H) {
          if (frag) out[fd_bias_frag_off(bidx * H + hh, nt, i, j)] = acc[q] + bb[hh];
          else out[(bidx * H + hh) * NN + ij] = acc[q] + bb[hh];
        }
      }
    }
  }
}

int fd_pair_bias2(int B, int N, int H, const void* z, const void* wb, const float* bb, float* out, int frag, hipStream_t st) {
  if (H > 8) return FDIPT_ESIZE;
  const long n_tiles = ((long)B * N * N + 31) / 32;
  const int grid = (int)(n_tiles / 4 + 1 < 2048 ? n_tiles / 4 + 1 : 2048);
  hipLaunchKernelGGL(pair_bias2_kernel, dim3(grid), dim3(FD_THREADS), 0, st, B, N, H, (const half_t*)z, (const half_t*)wb, bb,
                     out, frag);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// ------------------------------------------------------------------ fp32 mode: linear_b over the fp32 pair representation
// out[p, h] = sum_c z[p, c] Wb[h, c] + bb[h]   ([B, N, N, H] = the layout attn_kernel<PrecF32> reads), H = 8, c_z = 128.
// The tiled GEMM spent 1.04 ms per call at N = 1000, B = 4 on it (8 of 64 tile columns used, 2 GB of z at 2 TB/s); this is the
// one pass over z it has to be: 16 lanes share a pair row (32 B per lane: a wave reads 2 KB of consecutive pairs per step),
// 8 heads x 8 channels of Wb live in registers, the 16 partial sums of a head meet in a 4-step exchange in which every lane
// gives away half of its heads per step (8 cross-lane moves instead of 32), lanes 0, 2, .. of a group store the row's 8 values.
__device__ __forceinline__ float pb_xor(float v, int m) { return __shfl_xor(v, m, 64); }
__global__ __launch_bounds__(FD_THREADS) void pair_bias_f32_kernel(long n_pairs, const float* __restrict__ z, const float* __restrict__ wb,
                                                                   const float* __restrict__ bb, float* __restrict__ out) {
  const int lane = threadIdx.x & 63, l = lane & 15, g = lane >> 4;
  float w[8][8];
#pragma unroll
  for (int h = 0; h < 8; ++h) {
    const f32x4 a = *(const f32x4*)(wb + h * 128 + 8 * l), b = *(const f32x4*)(wb + h * 128 + 8 * l + 4);
    w[h][0] = a[0]; w[h][1] = a[1]; w[h][2] = a[2]; w[h][3] = a[3]; w[h][4] = b[0]; w[h][5] = b[1]; w[h][6] = b[2]; w[h][7] = b[3];
  }
  // after the exchange lane l holds head hsel: bit 2 from l & 8, bit 1 from l & 4, bit 0 from l & 2
  const int hsel = ((l >> 3) & 1) * 4 + ((l >> 2) & 1) * 2 + ((l >> 1) & 1);
  const float bias = bb[hsel];
  const long n_quads = (n_pairs + 3) >> 2;
  const long wave_id = (long)blockIdx.x * (FD_THREADS / 64) + (threadIdx.x >> 6), n_waves = (long)gridDim.x * (FD_THREADS / 64);
  constexpr int U = 4;  // quads in flight per wave
  for (long q0 = wave_id * U; q0 < n_quads; q0 += n_waves * U) {
    f32x4 x[U][2];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long p = (q0 + u) * 4 + g;
      if (p > n_pairs - 1) p = n_pairs - 1;
      x[u][0] = *(const f32x4*)(z + p * 128 + 8 * l);
      x[u][1] = *(const f32x4*)(z + p * 128 + 8 * l + 4);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float s[8];
#pragma unroll
      for (int h = 0; h < 8; ++h) {
        float a = x[u][0][0] * w[h][0];
        a = fmaf(x[u][0][1], w[h][1], a); a = fmaf(x[u][0][2], w[h][2], a); a = fmaf(x[u][0][3], w[h][3], a);
        a = fmaf(x[u][1][0], w[h][4], a); a = fmaf(x[u][1][1], w[h][5], a); a = fmaf(x[u][1][2], w[h][6], a); a = fmaf(x[u][1][3], w[h][7], a);
        s[h] = a;
      }
      // step 1 (lanes 8 apart): keep heads 4..7 if l & 8 else 0..3
      float t4[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float keep = (l & 8) ? s[4 + k] : s[k], give = (l & 8) ? s[k] : s[4 + k];
        t4[k] = keep + pb_xor(give, 8);
      }
      float t2[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float keep = (l & 4) ? t4[2 + k] : t4[k], give = (l & 4) ? t4[k] : t4[2 + k];
        t2[k] = keep + pb_xor(give, 4);
      }
      const float keep1 = (l & 2) ? t2[1] : t2[0], give1 = (l & 2) ? t2[0] : t2[1];
      float t1 = keep1 + pb_xor(give1, 2);
      t1 += pb_xor(t1, 1);
      const long p = (q0 + u) * 4 + g;
      if (!(l & 1) && p < n_pairs) out[p * 8 + hsel] = t1 + bias;
    }
  }
}
int fd_pair_bias_f32(long n_pairs, int H, int CZ, const float* z, const float* wb, const float* bb, float* out, hipStream_t st) {
  if (H != 8 || CZ != 128 || n_pairs <= 0) return FDIPT_EINVAL;
  const long n_quads = (n_pairs + 3) / 4, want = (n_quads + 15) / 16;
  const int grid = (int)(want < 4096 ? want : 4096);
  hipLaunchKernelGGL(pair_bias_f32_kernel, dim3(grid), dim3(FD_THREADS), 0, st, n_pairs, z, wb, bb, out);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
