// rowblock.hip — "row-complete" fused layers of the sequence transformer (nn.TransformerEncoderLayer, post-norm,
// framedipt/model/ipa_pytorch.py:433-443): x_a = LayerNorm(x + out_proj(att)) and x_b = LayerNorm(x_a + W2 relu(W1 x_a)),
// d_model = 320, bf16 operands, fp32 accumulate / LayerNorm.
//
// M = B*N is a few thousand rows: as tiled GEMM + LayerNorm launches these layers are latency-bound (≈11 + 7 us each),
// as 128-row chains (chain.hip) they occupy 19 CUs.  Here a block owns only 32 rows but ALL 320 output columns, and the
// block's 4 waves split the OUTPUT TILES (10 tiles of 32 features -> 3,3,2,2):
//   * x rows: one coalesced pass fp32 -> bf16 into an LDS tile; every wave keeps all 20 B fragments in registers;
//   * weights: fragment images (fd_chain_build_image, natural k order) read STRAIGHT from L2, one linear 1 KB load per
//     MFMA, the 20 fragments of a tile requested at once and the next tile's under the current tile's MFMAs;
//   * transposed MFMA (D^T[feature, row]): a lane owns one row, so LayerNorm statistics are lane-local sums + one
//     lane^32 shuffle + a 4-wave exchange through LDS;
//   * hidden activations of the feed-forward go through a second LDS tile (one barrier);
//   * residual rows and results cross a wave-private LDS tile so that global accesses are whole 128 B row segments.
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

#define RB_K 320
#define RB_KS (RB_K / 16)
#define RB_NT (RB_K / 32)
#define RB_XROW 656       // bytes per activation row in LDS (320 bf16 + 16: conflict-free b128 fragment reads)
#define RB_SROW 144       // bytes per row of a wave's 32 x 32 fp32 exchange tile

typedef __bf16 rb_bf16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ bf16x8 rb_ld(const char* p) { return __builtin_bit_cast(bf16x8, *(const u16x8*)p); }

// FFN = true: two layers (W0 relu, W1) ; false: one layer (W0).  Always + residual, LayerNorm.
template <bool FFN>
__global__ __launch_bounds__(FD_THREADS, 1) void rowblock_kernel(RowBlockArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;                                                  // x rows, bf16            [32][RB_XROW]
  char* hs = xs + 32 * RB_XROW;                                     // hidden rows, bf16       [32][RB_XROW] (FFN)
  char* st_all = hs + (FFN ? 32 * RB_XROW : 0);                     // per-wave exchange tiles [4][32][RB_SROW]
  float* cst = (float*)(st_all + 4 * 32 * RB_SROW);                 // b0 | b1 | gamma | beta  [4][RB_K]
  float (*red)[4][32] = (float (*)[4][32])(cst + 4 * RB_K);         // [2][4][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  const int row0 = blockIdx.x * 32;
  char* stg = st_all + wave * 32 * RB_SROW;
  const char* w0 = (const char*)a.w0;
  const char* w1 = (const char*)a.w1;
  // ---- first weight tile of this wave in flight before anything else
  bf16x8 Wf[2][RB_KS];
  auto w_load = [&](auto BUF, const char* img, int T) {
    constexpr int bf = decltype(BUF)::value;
#pragma unroll
    for (int s = 0; s < RB_KS; ++s) Wf[bf][s] = rb_ld(img + ((size_t)(T * RB_KS + s) * 64 + lane) * 16);
  };
  w_load(std::integral_constant<int, 0>{}, w0, wave);
  // ---- x rows -> LDS (bf16), constants -> LDS
  {
    f32x4 xv[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / 80, c4 = idx % 80;
      const int gr = row0 + r < a.M ? row0 + r : a.M - 1;
      xv[k] = *(const f32x4*)(a.in + (long)gr * a.ld_in + 4 * c4);
    }
    for (int v = tid; v < 4 * RB_K; v += FD_THREADS) {
      const int which = v / RB_K, c = v % RB_K;
      cst[v] = which == 0 ? a.b0[c] : (which == 1 ? (FFN ? a.b1[c] : 0.f) : (which == 2 ? a.gamma[c] : a.beta[c]));
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / 80, c4 = idx % 80;
      rb_bf16x4 pk;
#pragma unroll
      for (int q = 0; q < 4; ++q) pk[q] = (__bf16)xv[k][q];
      *(rb_bf16x4*)(xs + r * RB_XROW + 8 * c4) = pk;
    }
  }
  // residual row segments of this wave's output tiles: requested now, consumed after the last MFMA
  f32x4 rv[3][4];
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int T = wave + 4 * u < RB_NT ? wave + 4 * u : wave;
      const int r = 8 * it + (lane >> 3), gr = row0 + r < a.M ? row0 + r : a.M - 1;
      rv[u][it] = *(const f32x4*)(a.residual + (long)gr * a.ld_res + 32 * T + 4 * (lane & 7));
    }
  __syncthreads();
  bf16x8 X[RB_KS];
#pragma unroll
  for (int s = 0; s < RB_KS; ++s) X[s] = rb_ld(xs + li * RB_XROW + 32 * s + 16 * hi);

  f32x16 acc[3];
  // one layer: tiles wave, wave+4, wave+8 of `img` against the B fragments Bf; the first tile's fragments are in Wf[0]
  auto layer = [&](const char* img, const bf16x8* Bf) {
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int T = wave + 4 * u;
      if (u + 1 < 3 && T + 4 < RB_NT) {
        if (u & 1) w_load(std::integral_constant<int, 0>{}, img, T + 4);
        else w_load(std::integral_constant<int, 1>{}, img, T + 4);
      }
      if (T < RB_NT) {
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
        for (int s = 0; s < RB_KS; ++s) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[u & 1][s], Bf[s], c, 0, 0, 0);
        acc[u] = c;
      }
    }
  };
  layer(w0, X);
  if constexpr (FFN) {
    // hidden = relu(acc + b0) -> bf16 -> LDS rows (natural feature order), then every wave re-reads all of it as B fragments
    w_load(std::integral_constant<int, 0>{}, w1, wave);  // second layer's first tile: in flight across the barrier
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int T = wave + 4 * u;
      if (T < RB_NT) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f0 = 32 * T + 8 * g + 4 * hi;
          const f32x4 bv = *(const f32x4*)(cst + f0);
          rb_bf16x4 pk;
#pragma unroll
          for (int q = 0; q < 4; ++q) pk[q] = (__bf16)fmaxf(acc[u][4 * g + q] + bv[q], 0.f);
          *(rb_bf16x4*)(hs + li * RB_XROW + 2 * f0) = pk;
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int s = 0; s < RB_KS; ++s) X[s] = rb_ld(hs + li * RB_XROW + 32 * s + 16 * hi);
    layer(w1, X);
  }
  // ---- + bias + residual (fetched as 128 B row segments, turned into fragment layout through the wave's tile)
  const float* bo = cst + (FFN ? RB_K : 0);
  float s1 = 0.f;
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int T = wave + 4 * u;
    if (T < RB_NT) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *(f32x4*)(stg + (8 * it + (lane >> 3)) * RB_SROW + 16 * (lane & 7)) = rv[u][it];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *(const f32x4*)(bo + 32 * T + 8 * g + 4 * hi);
        const f32x4 rr = *(const f32x4*)(stg + li * RB_SROW + (8 * g + 4 * hi) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float v = acc[u][4 * g + q] + bv[q] + rr[q];
          acc[u][4 * g + q] = v;
          s1 += v;
        }
      }
    }
  }
  // ---- LayerNorm over the row: lane-local sums, lane^32, then the 4 waves through LDS (two passes)
  s1 += __shfl_xor(s1, 32, 64);
  if (hi == 0) red[0][wave][li] = s1;
  __syncthreads();
  const float mu = (red[0][0][li] + red[0][1][li] + red[0][2][li] + red[0][3][li]) * (1.0f / RB_K);
  float s2 = 0.f;
#pragma unroll
  for (int u = 0; u < 3; ++u)
    if (wave + 4 * u < RB_NT)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float d = acc[u][r] - mu;
        s2 += d * d;
      }
  s2 += __shfl_xor(s2, 32, 64);
  if (hi == 0) red[1][wave][li] = s2;
  __syncthreads();
  const float rstd = 1.0f / sqrtf((red[1][0][li] + red[1][1][li] + red[1][2][li] + red[1][3][li]) * (1.0f / RB_K) + 1e-5f);
  // ---- normalise, back through the wave's tile, store 128 B row segments
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int T = wave + 4 * u;
    if (T < RB_NT) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f0 = 32 * T + 8 * g + 4 * hi;
        const f32x4 gm = *(const f32x4*)(cst + 2 * RB_K + f0), bt = *(const f32x4*)(cst + 3 * RB_K + f0);
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = (acc[u][4 * g + q] - mu) * rstd * gm[q] + bt[q];
        *(f32x4*)(stg + li * RB_SROW + (8 * g + 4 * hi) * 4) = o;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = 8 * it + (lane >> 3);
        const f32x4 o = *(const f32x4*)(stg + r * RB_SROW + 16 * (lane & 7));
        if (row0 + r < a.M) *(f32x4*)(a.out + (long)(row0 + r) * a.ld_out + 32 * T + 4 * (lane & 7)) = o;
      }
    }
  }
}

int fd_rowblock_supported(int d_model) { return d_model == RB_K; }

int fd_rowblock(int ffn, const RowBlockArgs& a, hipStream_t st) {
  if (a.M <= 0 || (a.ld_in & 3) || (a.ld_res & 3) || (a.ld_out & 3) || !a.residual || a.residual == a.out) return FDIPT_EINVAL;
  const dim3 grid(cdiv(a.M, 32));
  const size_t smem = (size_t)(ffn ? 2 : 1) * 32 * RB_XROW + 4 * 32 * RB_SROW + 4 * RB_K * 4 + 2 * 4 * 32 * 4;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)rowblock_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)rowblock_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_set = true;
  }
  if (ffn) hipLaunchKernelGGL((rowblock_kernel<true>), grid, dim3(FD_THREADS), smem, st, a);
  else hipLaunchKernelGGL((rowblock_kernel<false>), grid, dim3(FD_THREADS), smem, st, a);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
