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
