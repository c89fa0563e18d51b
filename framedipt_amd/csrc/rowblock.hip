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

#pragma clang fp contract(off)  // the fused frame update keeps the float32 evaluation order of the reference expressions

#define RB_SROW 144       // bytes per row of a wave's 32 x 32 fp32 exchange tile

typedef fd_h rb_hx4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ hx8 rb_ld(const char* p) { return __builtin_bit_cast(hx8, *(const u16x8*)p); }
__host__ __device__ constexpr int rb_max(int a, int b) { return a > b ? a : b; }

// K0: input width (zero-padded to a multiple of 16); N1 / N2: hidden widths (0 = absent); NOUT: output width;
// FLAGS bit0 / bit1: ReLU after hidden 1 / 2, bit2: LayerNorm.  Residual and the final row mask are optional (pointers).
// FLAGS bit5 (SPLIT): every product runs on split operands — activations x = hi + lo and weights W = hi + lo, each part one
// half-precision value (22 significant bits together), as Whi.xhi + Whi.xlo + Wlo.xhi with fp32 accumulation: the accuracy of
// an fp32 product at a third of the half-precision MFMA rate (the fp32 MFMA runs at a sixteenth).  The layers of the node
// path whose operand rounding dominates the error of the predicted frames / psi use it (tests/err_budget.py).
// Activation rows in LDS are [32][width bf16 + 16 B]: with widths 80..320 the 16 lanes of a b128 read hit 16 distinct
// 16 B slots, no swizzle needed.
template <int N, class F>
__device__ __forceinline__ void ch_rb_for(F&& f) {  // f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>)
  if constexpr (N > 0) {
    ch_rb_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

template <int K0, int N1, int N2, int NOUT, int FLAGS>
struct RBShape {
  static constexpr int KS0 = (K0 + 15) / 16;
  static constexpr int NL = 1 + (N1 > 0) + (N2 > 0);
  static constexpr int KSMAX = rb_max(KS0, rb_max(N1 / 16, N2 / 16));
  static constexpr int WMAX = rb_max(KS0 * 16, rb_max(N1, N2));          // widest activation tile
  static constexpr int XROW = WMAX * 2 + 16;                              // bytes per LDS activation row
  static constexpr bool LN = (FLAGS & 4) != 0;
  static constexpr int CP = (FLAGS & 64) ? 2 : 1;                          // column parts of the LAST layer: blockIdx.y takes NOUT / CP of its
                                                                           // columns (the hidden layers are recomputed by every part)
  static constexpr int NTO_BLK = NOUT / 32 / CP;                           // output tiles of a block
  static constexpr int NTMAX = rb_max(NTO_BLK, rb_max(N1 / 32, N2 / 32));
  static constexpr int NTW = (NTMAX + 3) / 4;                              // output tiles per wave
  static constexpr bool BB = (FLAGS & 8) != 0;                            // fused BackboneUpdate + compose_q_update_vec
  static constexpr bool IMG = (FLAGS & 16) != 0;                          // output = edge_transition4 fold-fragment images
  static constexpr bool SPLIT = (FLAGS & 32) != 0;                        // split operands: x = hi + lo, W = hi + lo, 3 MFMAs per k-step
  static constexpr int XBUF = 32 * XROW * (SPLIT ? 2 : 1);                // one activation buffer: hi rows (then lo rows)
  static constexpr int NCONST = N1 + N2 + NOUT + (LN ? 2 * NOUT : 0) + (BB ? 6 * NOUT : 0);  // b0 | b1 | b_out | gamma | beta | Wbb
  static constexpr size_t SMEM = (size_t)(NL > 1 ? 2 : 1) * XBUF + 4 * 32 * RB_SROW + (size_t)NCONST * 4 + 2 * 4 * 32 * 4 + 128 + (BB ? 4 * 32 * 8 * 4 : 0) + 16;
  static_assert(N1 % 32 == 0 && N2 % 32 == 0 && NOUT % 32 == 0 && NOUT <= 1024, "tile shapes");
  static_assert(CP == 1 || ((FLAGS & 16) && N1 > 0), "column parts: image outputs (no row-wide epilogue) behind a hidden layer");
};

template <int K0, int N1, int N2, int NOUT, int FLAGS>
__global__ __launch_bounds__(FD_THREADS, 1) void rowblock_kernel(RowBlockArgs a) {
  using S = RBShape<K0, N1, N2, NOUT, FLAGS>;
  constexpr int KS0 = S::KS0, NL = S::NL, XROW = S::XROW, KSMAX = S::KSMAX, NTW = S::NTW;
  constexpr bool LN = S::LN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;                                                  // activation rows (ping)  [32][XROW]
  char* hs = xs + S::XBUF;                                          // activation rows (pong)  [32][XROW]  (NL > 1)
  char* st_all = hs + (NL > 1 ? S::XBUF : 0);                       // per-wave exchange tiles [4][32][RB_SROW]
  constexpr bool SPLIT = S::SPLIT;
  constexpr int XLO = 32 * XROW;                                    // SPLIT: the lo rows follow the hi rows of a buffer
  float* cst = (float*)(st_all + 4 * 32 * RB_SROW);                 // b0 | b1 | b_out | gamma | beta
  float (*red)[4][32] = (float (*)[4][32])(cst + S::NCONST);        // [2][4][32]
  float* pmask = (float*)(red + 2);                                 // [32] final row mask
  float* red3 = pmask + 32;                                         // [4][32][8] partial BackboneUpdate outputs (BB)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  const int row0 = blockIdx.x * 32;
  char* stg = st_all + wave * 32 * RB_SROW;
  const char* wimg[3] = {(const char*)a.w0, (const char*)a.w1, (const char*)a.w2};
  const char* wlo[3] = {(const char*)a.w0l, (const char*)a.w1l, (const char*)a.w2l};  // SPLIT: images of W - half(W)
  // ---- first weight tile of this wave in flight before anything else
  // weight-fragment buffers: tile u of a wave lives in buffer u % NB; wide outputs with short K (8 tiles of 8 fragments per wave)
  // keep three tiles in flight instead of one (each tile is a dependent L2 round trip otherwise)
  constexpr int NB = ((NTW >= 6 || S::IMG) && KSMAX <= 16) ? 4 : 2;  // (image outputs: the transposed products need the two-tile form)
  const int t0 = S::CP > 1 ? (int)blockIdx.y * S::NTO_BLK : 0;        // first output tile of this block's column part
  // split operands: a tile's hi fragments live in an even buffer, its lo fragments in the odd one behind it; NB == 4 = two
  // tiles in flight (wide outputs with short K), NB == 2 = one
  constexpr int PAIRS = NB / 2;
  hx8 Wf[NB][KSMAX];
  auto w_load = [&](auto BUF, auto KSC, const char* img, int T) {
    constexpr int bf = decltype(BUF)::value, KS = decltype(KSC)::value;
#pragma unroll
    for (int s = 0; s < KS; ++s) Wf[bf][s] = rb_ld(img + ((size_t)(T * KS + s) * 64 + lane) * 16);
  };
  w_load(std::integral_constant<int, 0>{}, std::integral_constant<int, KS0>{}, wimg[0], wave);
  if constexpr (SPLIT) w_load(std::integral_constant<int, 1>{}, std::integral_constant<int, KS0>{}, wlo[0], wave);
  // ---- input rows -> LDS (bf16, zero-padded to 16 KS0 columns), constants -> LDS
  {
    constexpr int C4 = KS0 * 4;                       // float4 columns per row (padded)
    constexpr int NV = (32 * C4 + FD_THREADS - 1) / FD_THREADS;
    f32x4 xv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / C4, c4 = idx % C4;
      const int gr = row0 + r < a.M ? row0 + r : a.M - 1;
      xv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < 32 * C4 && 4 * c4 < K0) xv[k] = *(const f32x4*)(a.in + (long)gr * a.ld_in + 4 * c4);
    }
    {  // constants: all loads of a thread in flight together (a rolled loop is one dependent L2 round trip per iteration)
      constexpr int NCV = (S::NCONST + FD_THREADS - 1) / FD_THREADS;
      float cv[NCV];
#pragma unroll
      for (int k = 0; k < NCV; ++k) {
        const int v = tid + k * FD_THREADS;
        float x = 0.f;
        if (v < N1) x = a.b0[v];
        else if (v < N1 + N2) x = a.b1[v - N1];
        else if (v < N1 + N2 + NOUT) x = (NL == 1 ? a.b0 : (NL == 2 ? a.b1 : a.b2))[v - N1 - N2];
        else if (v < N1 + N2 + 2 * NOUT) { if (S::LN) x = a.gamma[v - N1 - N2 - NOUT]; }
        else if (v < N1 + N2 + 3 * NOUT) { if (S::LN) x = a.beta[v - N1 - N2 - 2 * NOUT]; }
        else if (v < S::NCONST) { if (S::BB) x = a.bb_w[v - N1 - N2 - 3 * NOUT]; }  // [6][NOUT]
        cv[k] = x;
      }
#pragma unroll
      for (int k = 0; k < NCV; ++k)
        if (tid + k * FD_THREADS < S::NCONST) cst[tid + k * FD_THREADS] = cv[k];
    }
    if (tid < 32) pmask[tid] = a.rowmask_post ? a.rowmask_post[row0 + tid < a.M ? row0 + tid : a.M - 1] : 1.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / C4, c4 = idx % C4;
      rb_hx4 pk, pl;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pk[q] = (fd_h)xv[k][q];
        pl[q] = (fd_h)(xv[k][q] - (float)pk[q]);
      }
      if (idx < 32 * C4) {
        *(rb_hx4*)(xs + r * XROW + 8 * c4) = pk;
        if constexpr (SPLIT) *(rb_hx4*)(xs + XLO + r * XROW + 8 * c4) = pl;
      }
    }
  }
  // residual row segments of this wave's output tiles: requested now, consumed after the last MFMA
  constexpr int NTO = NOUT / 32;
  f32x4 rv[NTW][4];
  const bool has_res = a.residual != nullptr;
#pragma unroll
  for (int u = 0; u < NTW; ++u)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int T = wave + 4 * u < NTO ? wave + 4 * u : wave;
      const int r = 8 * it + (lane >> 3), gr = row0 + r < a.M ? row0 + r : a.M - 1;
      rv[u][it] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (has_res) rv[u][it] = *(const f32x4*)(a.residual + (long)gr * a.ld_res + 32 * T + 4 * (lane & 7));
    }
  const unsigned warm_tok = fd_l2_warm(a.warm, blockIdx.x, gridDim.x, tid, FD_THREADS);
  __syncthreads();

  hx8 X[KSMAX];
  hx8 Xl[SPLIT ? KSMAX : 1];  // SPLIT: lo parts of the B fragments
  f32x16 acc[NTW];
  // one layer: tiles wave, wave+4, wave+8 (< NT) of `img` against the B fragments X[0..KS); the first tile's fragments
  // are already in Wf[0] (SPLIT: and its lo fragments in Wf[1])
  auto layer = [&](auto KSC, auto NTC, const char* img, const char* img_lo, auto SWAPC, int toff) {
    constexpr int KS = decltype(KSC)::value, NT = decltype(NTC)::value;
    constexpr bool SWAP = decltype(SWAPC)::value;  // operands exchanged: lane = output feature, registers = rows
    if constexpr (SPLIT && PAIRS == 1) {
      // per tile: Whi.xhi + Whi.xlo out of buffer 0, then Wlo.xhi out of buffer 1; the next tile's hi fragments are requested
      // when buffer 0 is free (under the lo pass), its lo fragments when buffer 1 is (under the next tile's hi passes)
      static_assert(!SWAP, "one tile in flight: untransposed products only");
      ch_rb_for<NTW>([&](auto U) {
        constexpr int u = decltype(U)::value;
        const int T = wave + 4 * u;
        if (T < NT) {
          f32x16 c;
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
          for (int s = 0; s < KS; ++s) c = fd_mfma32(Wf[0][s], X[s], c);
#pragma unroll
          for (int s = 0; s < KS; ++s) c = fd_mfma32(Wf[0][s], Xl[s], c);
          if (u + 1 < NTW && T + 4 < NT) w_load(std::integral_constant<int, 0>{}, KSC, img, toff + T + 4);
#pragma unroll
          for (int s = 0; s < KS; ++s) c = fd_mfma32(Wf[1][s], X[s], c);
          if (u + 1 < NTW && T + 4 < NT) w_load(std::integral_constant<int, 1>{}, KSC, img_lo, toff + T + 4);
          acc[u] = c;
        }
      });
      return;
    }
    if constexpr (SPLIT && PAIRS == 2) {
      // two tiles in flight: tile u lives in buffers 2 (u & 1) (hi) and 2 (u & 1) + 1 (lo); tile u + 1 is requested before the
      // products of tile u start (tile 0 was requested by the caller)
      ch_rb_for<NTW>([&](auto U) {
        constexpr int u = decltype(U)::value, bh = 2 * (u & 1), bn = 2 * ((u + 1) & 1);
        const int T = wave + 4 * u;
        if (u + 1 < NTW && T + 4 < NT) {
          w_load(std::integral_constant<int, bn>{}, KSC, img, toff + T + 4);
          w_load(std::integral_constant<int, bn + 1>{}, KSC, img_lo, toff + T + 4);
        }
        if (T < NT) {
          f32x16 c;
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
          for (int s = 0; s < KS; ++s) c = SWAP ? fd_mfma32(X[s], Wf[bh + 1][s], c) : fd_mfma32(Wf[bh + 1][s], X[s], c);
#pragma unroll
          for (int s = 0; s < KS; ++s) c = SWAP ? fd_mfma32(Xl[s], Wf[bh][s], c) : fd_mfma32(Wf[bh][s], Xl[s], c);
#pragma unroll
          for (int s = 0; s < KS; ++s) c = SWAP ? fd_mfma32(X[s], Wf[bh][s], c) : fd_mfma32(Wf[bh][s], X[s], c);
          acc[u] = c;
        }
      });
      return;
    }
    auto load_tile = [&](auto V) {  // tile v of this wave -> buffer v % NB
      constexpr int v = decltype(V)::value;
      if (v < NTW && wave + 4 * v < NT) w_load(std::integral_constant<int, v % NB>{}, KSC, img, toff + wave + 4 * v);
    };
    if constexpr (NB > 2) {  // tiles 1 .. NB-2 up front (tile 0 was requested by the caller, tile u + NB - 1 follows at tile u)
      load_tile(std::integral_constant<int, 1>{});
      load_tile(std::integral_constant<int, 2>{});
    }
    ch_rb_for<NTW>([&](auto U) {
      constexpr int u = decltype(U)::value;
      const int T = wave + 4 * u;
      load_tile(std::integral_constant<int, u + NB - 1>{});
      if (T < NT) {
        f32x16 c;
#pragma unroll
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
          c = SWAP ? fd_mfma32(X[s], Wf[u % NB][s], c)
                   : fd_mfma32(Wf[u % NB][s], X[s], c);
        acc[u] = c;
      }
    });
  };
  // hidden = act(acc + bias) -> bf16 -> LDS rows (natural feature order); every wave then re-reads all of it
  auto to_hidden = [&](auto NTC, auto RELU, const float* bias, char* dst, unsigned short* hid_h16, int hid_ld) {
    constexpr int NT = decltype(NTC)::value;
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      const int T = wave + 4 * u;
      if (T < NT) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f0 = 32 * T + 8 * g + 4 * hi;
          const f32x4 bv = *(const f32x4*)(bias + f0);
          rb_hx4 pk, pl;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            float v = acc[u][4 * g + q] + bv[q];
            if (decltype(RELU)::value) v = fmaxf(v, 0.f);
            pk[q] = (fd_h)v;
            pl[q] = (fd_h)(v - (float)pk[q]);
          }
          *(rb_hx4*)(dst + li * XROW + 2 * f0) = pk;
          if constexpr (SPLIT) *(rb_hx4*)(dst + XLO + li * XROW + 2 * f0) = pl;
          if (hid_h16 && row0 + li < a.M) *(rb_hx4*)(hid_h16 + (long)(row0 + li) * hid_ld + f0) = pk;  // optional bf16 copy of the rows
        }
      }
    }
  };
  auto x_load = [&](auto KSC, const char* buf) {  // this lane's B fragments of the activation rows in `buf`
#pragma unroll
    for (int s = 0; s < decltype(KSC)::value; ++s) {
      X[s] = rb_ld(buf + li * XROW + 32 * s + 16 * hi);
      if constexpr (SPLIT) Xl[s] = rb_ld(buf + XLO + li * XROW + 32 * s + 16 * hi);
    }
  };
  auto w_first = [&](auto KSC, int l, int toff) {  // first tile of layer l: in flight across the barrier
    w_load(std::integral_constant<int, 0>{}, KSC, wimg[l], toff + wave);
    if constexpr (SPLIT) w_load(std::integral_constant<int, 1>{}, KSC, wlo[l], toff + wave);
  };
  x_load(std::integral_constant<int, KS0>{}, xs);
  constexpr int NTO_B = S::NTO_BLK;
  if constexpr (NL == 1) {
    layer(std::integral_constant<int, KS0>{}, std::integral_constant<int, NTO_B>{}, wimg[0], wlo[0], std::false_type{}, 0);
  } else {
    layer(std::integral_constant<int, KS0>{}, std::integral_constant<int, N1 / 32>{}, wimg[0], wlo[0], std::false_type{}, 0);
    w_first(std::integral_constant<int, N1 / 16>{}, 1, NL == 2 ? t0 : 0);
    to_hidden(std::integral_constant<int, N1 / 32>{}, std::integral_constant<bool, (FLAGS & 1) != 0>{}, cst, hs, a.hid_h16, N1);
    __syncthreads();
    x_load(std::integral_constant<int, N1 / 16>{}, hs);
    if constexpr (NL == 2) {
      layer(std::integral_constant<int, N1 / 16>{}, std::integral_constant<int, NTO_B>{}, wimg[1], wlo[1], std::integral_constant<bool, S::IMG>{}, t0);
    } else {
      layer(std::integral_constant<int, N1 / 16>{}, std::integral_constant<int, N2 / 32>{}, wimg[1], wlo[1], std::false_type{}, 0);
      w_first(std::integral_constant<int, N2 / 16>{}, 2, 0);
      to_hidden(std::integral_constant<int, N2 / 32>{}, std::integral_constant<bool, (FLAGS & 2) != 0>{}, cst + N1, xs, nullptr, 0);  // xs is free again
      __syncthreads();
      x_load(std::integral_constant<int, N2 / 16>{}, xs);
      layer(std::integral_constant<int, N2 / 16>{}, std::integral_constant<int, NTO_B>{}, wimg[2], wlo[2], std::false_type{}, 0);
    }
  }
  if constexpr (S::IMG) {
    // ---- edge_transition4 fold-fragment images straight from the accumulators (lane = output column, registers = 4-runs of
    // rows): columns 0..511 = [A1 | Af] of 8 consecutive flattened rows -> a_img[row / 8][column / 32][column % 32][row % 8];
    // columns 512..1023 = [B1 | Bf] of 4 consecutive residues j of sample b -> b_img[b][j / 4][..][0..3], and the same 8 bytes
    // as elements 4..7 of sample b - 1 (the rows of a patch that straddles two samples); elements 4..7 of the last sample are
    // zeros (finite: they meet zeros of the selection matrix).  Needs img_N % 4 == 0.
    static_assert(NL == 2 && NOUT == 1024, "ET4 image kind");
    const float* bo = cst + N1 + N2;
    const int NJ4 = a.img_N >> 2;
    half_t* ia = (half_t*)a.img_a;
    half_t* ib = (half_t*)a.img_b;
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      if (wave + 4 * u >= NTO_B) continue;
      const int T = t0 + wave + 4 * u;
      const float bv = bo[32 * T + li];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int r0 = row0 + 8 * g + 4 * hi;  // rows r0 .. r0 + 3 (M % 4 == 0: all four valid or none)
        u16x4 o = {0, 0, 0, 0};
        if (r0 < a.M) o = u16x4{f2h(acc[u][4 * g] + bv), f2h(acc[u][4 * g + 1] + bv), f2h(acc[u][4 * g + 2] + bv), f2h(acc[u][4 * g + 3] + bv)};
        if (T < 16) {
          if ((r0 >> 3) < ((a.M + 7) >> 3)) *(u16x4*)(ia + ((((long)(r0 >> 3) * 16 + T) * 32 + li) << 3) + (r0 & 4)) = o;  // (rows beyond M: zeros; the image is padded to 8 rows)
        } else if (r0 < a.M) {
          const int b = r0 / a.img_N, jt = (r0 - b * a.img_N) >> 2, ft = T - 16;
          half_t* dst = ib + ((((long)b * NJ4 + jt) * 16 + ft) * 32 + li) * 8;
          *(u16x4*)dst = o;
          if (b > 0) *(u16x4*)(dst - (long)NJ4 * 16 * 32 * 8 + 4) = o;
          if (b == a.img_B - 1) *(u16x4*)(dst + 4) = u16x4{0, 0, 0, 0};
        }
      }
    }
    fd_l2_warm_done(warm_tok);
    return;
  }
  // ---- + bias + residual (row segments -> fragment layout through the wave's tile)
  const float* bo = cst + N1 + N2;
  float s1 = 0.f;
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int T = wave + 4 * u;
    if (T < NTO) {
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
  float mu = 0.f, rstd = 1.f;
  if constexpr (LN) {  // lane-local sums, lane^32, then the 4 waves through LDS (two passes)
    s1 += __shfl_xor(s1, 32, 64);
    if (hi == 0) red[0][wave][li] = s1;
    __syncthreads();
    mu = (red[0][0][li] + red[0][1][li] + red[0][2][li] + red[0][3][li]) * (1.0f / NOUT);
    float s2 = 0.f;
#pragma unroll
    for (int u = 0; u < NTW; ++u)
      if (wave + 4 * u < NTO)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float d = acc[u][r] - mu;
          s2 += d * d;
        }
    s2 += __shfl_xor(s2, 32, 64);
    if (hi == 0) red[1][wave][li] = s2;
    __syncthreads();
    rstd = 1.0f / sqrtf((red[1][0][li] + red[1][1][li] + red[1][2][li] + red[1][3][li]) * (1.0f / NOUT) + 1e-5f);
  }
  const float pm = pmask[li];
  float pd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // ---- (normalise,) mask, back through the wave's tile, store 128 B row segments
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int T = wave + 4 * u;
    if (T < NTO) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f0 = 32 * T + 8 * g + 4 * hi;
        f32x4 o;
        if constexpr (LN) {
          const f32x4 gm = *(const f32x4*)(bo + NOUT + f0), bt = *(const f32x4*)(bo + 2 * NOUT + f0);
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = ((acc[u][4 * g + q] - mu) * rstd * gm[q] + bt[q]) * pm;
        } else {
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = acc[u][4 * g + q] * pm;
        }
        *(f32x4*)(stg + li * RB_SROW + (8 * g + 4 * hi) * 4) = o;
        if constexpr (S::BB) {
#pragma unroll
          for (int k = 0; k < 6; ++k) {
            const f32x4 wv = *(const f32x4*)(bo + 3 * NOUT + k * NOUT + f0);
            pd[k] += (o[0] * wv[0] + o[1] * wv[1]) + (o[2] * wv[2] + o[3] * wv[3]);
          }
        }
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = 8 * it + (lane >> 3);
        const f32x4 o = *(const f32x4*)(stg + r * RB_SROW + 16 * (lane & 7));
        if (row0 + r < a.M) {
          if (a.out2 && 32 * T >= a.split) *(f32x4*)(a.out2 + (long)(row0 + r) * a.ld_out2 + 32 * T - a.split + 4 * (lane & 7)) = o;
          else *(f32x4*)(a.out + (long)(row0 + r) * a.ld_out + 32 * T + 4 * (lane & 7)) = o;
        }
      }
    }
  }
  if constexpr (S::BB) {
    // BackboneUpdate (ipa_pytorch.py:542-545): 6 outputs per row = this lane's partial dots + lane^32 + the other waves
    // through LDS, then Rigid.compose_q_update_vec with the update mask (rigid_utils.py:587-616,1039-1063) in place
#pragma unroll
    for (int k = 0; k < 6; ++k) pd[k] += __shfl_xor(pd[k], 32, 64);
    if (hi == 0) {
#pragma unroll
      for (int k = 0; k < 6; ++k) red3[(wave * 32 + li) * 8 + k] = pd[k];
    }
    __syncthreads();
    if (tid < 32 && row0 + tid < a.M) {
      const long r = row0 + tid;
      float upd[6];
#pragma unroll
      for (int k = 0; k < 6; ++k)
        upd[k] = ((red3[tid * 8 + k] + red3[(32 + tid) * 8 + k]) + (red3[(64 + tid) * 8 + k] + red3[(96 + tid) * 8 + k])) + a.bb_b[k];
      const float m = a.upd_mask ? a.upd_mask[r] : 1.f;
      const float q0 = a.quat[r * 4], q1 = a.quat[r * 4 + 1], q2 = a.quat[r * 4 + 2], q3 = a.quat[r * 4 + 3];
      // quat_multiply_by_vec (rigid_utils.py:266-279), quat_to_rot (:173-205), rot_vec_mul (:82-106)
      const float dq0 = -q1 * upd[0] - q2 * upd[1] - q3 * upd[2];
      const float dq1 = q0 * upd[0] + q2 * upd[2] - q3 * upd[1];
      const float dq2 = q0 * upd[1] - q1 * upd[2] + q3 * upd[0];
      const float dq3 = q0 * upd[2] + q1 * upd[1] - q2 * upd[0];
      const float R0 = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, R1 = 2 * q1 * q2 - 2 * q0 * q3, R2 = 2 * q1 * q3 + 2 * q0 * q2;
      const float R3 = 2 * q1 * q2 + 2 * q0 * q3, R4 = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3, R5 = 2 * q2 * q3 - 2 * q0 * q1;
      const float R6 = 2 * q1 * q3 - 2 * q0 * q2, R7 = 2 * q2 * q3 + 2 * q0 * q1, R8 = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
      const float d0 = R0 * upd[3] + R1 * upd[4] + R2 * upd[5];
      const float d1 = R3 * upd[3] + R4 * upd[4] + R5 * upd[5];
      const float d2 = R6 * upd[3] + R7 * upd[4] + R8 * upd[5];
      const float n0 = q0 + dq0 * m, n1 = q1 + dq1 * m, n2 = q2 + dq2 * m, n3 = q3 + dq3 * m;
      const float nrm = sqrtf(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
      a.quat[r * 4] = n0 / nrm; a.quat[r * 4 + 1] = n1 / nrm; a.quat[r * 4 + 2] = n2 / nrm; a.quat[r * 4 + 3] = n3 / nrm;
      fd_store3(a.trans + r * 3, a.trans[r * 3] + d0 * m, a.trans[r * 3 + 1] + d1 * m, a.trans[r * 3 + 2] + d2 * m);
    }
  }
  fd_l2_warm_done(warm_tok);
}

// ------------------------------------------------------------------ whole post-attention half of an encoder layer
// x_a = LayerNorm1(x + W_o att + b_o);  x_b = LayerNorm2(x_a + W2 relu(W1 x_a + b1) + b2)  in ONE launch (d_model 320): x_a
// never leaves the block — its bf16 rows go to LDS as the feed-forward input, its fp32 values stay in the accumulator
// registers of the wave that owns the tile (the same tile ownership as the feed-forward output) and are the residual.
#define TL_D 320
#define TL_KS (TL_D / 16)
#define TL_NT (TL_D / 32)
#define TL_XROW (TL_D * 2 + 16)
#define TL_NP 256  // post_tfmr output width (POST variant)
#define TL_SMEM(SPLIT) (((SPLIT) ? 4 : 2) * 32 * TL_XROW + 4 * 32 * RB_SROW + (7 * TL_D + TL_NP) * 4 + 2 * 4 * 32 * 4 + 16)
// POST: the last layer of the stack also applies post_tfmr (Linear d_model -> c_s) + the node residual (ipa:539) to its own
// output rows, which then never go to memory.
// SPLIT: every product on split operands (activations and weights as hi + lo half-precision parts, 3 MFMAs per k-step; see
// rowblock_kernel): wol / w1l / w2l / wpl are the lo images.
template <bool POST, bool SPLIT>
__global__ __launch_bounds__(FD_THREADS, 1) void tfmr_tail_kernel(TfmrTailArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;                                            // att rows, then hidden rows   [32][TL_XROW]
  constexpr int XLO = 32 * TL_XROW;                           // SPLIT: the lo rows follow the hi rows of a buffer
  constexpr int XBUF = SPLIT ? 2 * XLO : XLO;
  char* hs = xs + XBUF;                                       // x_a rows                     [32][TL_XROW]
  char* st_all = hs + XBUF;                                   // per-wave exchange tiles      [4][32][RB_SROW]
  float* cst = (float*)(st_all + 4 * 32 * RB_SROW);           // b_o | g1 | be1 | b1 | b2 | g2 | be2 | b_post
  float (*red)[4][32] = (float (*)[4][32])(cst + 7 * TL_D + TL_NP);   // [2][4][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  const int row0 = blockIdx.x * 32;
  char* stg = st_all + wave * 32 * RB_SROW;
  FD_STAMP(0);
  // SPLIT: THREE fragment buffers.  Tile u of a wave has its hi fragments in buffer (2 u) % 3 and its lo fragments in (2 u + 1) % 3; the
  // third buffer is free while tile u runs and takes tile u + 1's hi fragments at the START of tile u, the lo fragments follow into the hi
  // buffer once tile u's hi products are done: every fragment load has a whole tile (60 matrix instructions) of lead instead of the 20 lo
  // products.  The 80 registers come from the lo parts of the activation fragments, which the hi x lo pass now reads from LDS (a 3-deep
  // ring, one read per product).  Measured (tools/micro/tt_bench.hip 1): the three stages 9.9 / 9.5 / 8.7 k -> 9.0 / 9.0 / 8.5 k cycles,
  // 24.3 -> 23.0 us stand-alone — the lead was not what bounds a stage: 400 KB of hi + lo fragments per block and stage through the
  // CU's 64 B/clk L2 path are 6.4 k cycles next to 5.8 k of matrix work.
  hx8 Wf[SPLIT ? 3 : 2][TL_KS];
  auto w_load = [&](auto BUF, const char* img, int T) {
    constexpr int bf = decltype(BUF)::value;
#pragma unroll
    for (int s = 0; s < TL_KS; ++s) Wf[bf][s] = rb_ld(img + ((size_t)(T * TL_KS + s) * 64 + lane) * 16);
  };
  // first tile of a stage: hi fragments -> buffer 0 (SPLIT: lo fragments -> buffer 1)
  auto w_first = [&](const void* img, const void* img_lo) {
    w_load(std::integral_constant<int, 0>{}, (const char*)img, wave);
    if constexpr (SPLIT) w_load(std::integral_constant<int, 1>{}, (const char*)img_lo, wave);
  };
  // 4 values -> half-precision row pieces at byte offset `off` of an activation buffer (SPLIT: hi and lo parts)
  auto put4 = [&](char* buf, int off, float v0, float v1, float v2, float v3) {
    const float v[4] = {v0, v1, v2, v3};
    rb_hx4 pk, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pk[q] = (fd_h)v[q];
      pl[q] = (fd_h)(v[q] - (float)pk[q]);
    }
    *(rb_hx4*)(buf + off) = pk;
    if constexpr (SPLIT) *(rb_hx4*)(buf + XLO + off) = pl;
  };
  w_first(a.wo, a.wol);
  {
    f32x4 xv[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / 80, c4 = idx % 80;
      const int gr = row0 + r < a.M ? row0 + r : a.M - 1;
      xv[k] = *(const f32x4*)(a.att + (long)gr * a.ld + 4 * c4);
    }
    {  // the 7 x 320 constants: all 9 loads of a thread in flight together (a rolled loop is 9 dependent L2 round trips)
      constexpr int NC = 7 * TL_D + (POST ? TL_NP : 0), NCV = (NC + FD_THREADS - 1) / FD_THREADS;
      float cv[NCV];
#pragma unroll
      for (int k = 0; k < NCV; ++k) {
        const int v = tid + k * FD_THREADS, which = v / TL_D, c = v % TL_D;
        const float* src = which == 0 ? a.bo : which == 1 ? a.g1 : which == 2 ? a.be1 : which == 3 ? a.b1 : which == 4 ? a.b2 : which == 5 ? a.g2 : a.be2;
        cv[k] = v < 7 * TL_D ? src[c] : (POST && v < NC ? a.bp[v - 7 * TL_D] : 0.f);
      }
#pragma unroll
      for (int k = 0; k < NCV; ++k)
        if (tid + k * FD_THREADS < NC) cst[tid + k * FD_THREADS] = cv[k];
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / 80, c4 = idx % 80;
      put4(xs, r * TL_XROW + 8 * c4, xv[k][0], xv[k][1], xv[k][2], xv[k][3]);
    }
  }
  f32x4 rv[3][4];  // residual x: row segments of this wave's tiles
#pragma unroll
  for (int u = 0; u < 3; ++u)
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int T = wave + 4 * u < TL_NT ? wave + 4 * u : wave;
      const int r = 8 * it + (lane >> 3), gr = row0 + r < a.M ? row0 + r : a.M - 1;
      rv[u][it] = *(const f32x4*)(a.x + (long)gr * a.ld + 32 * T + 4 * (lane & 7));
    }
  f32x4 rvp[POST ? 2 : 1][4];  // POST: node rows (residual of post_tfmr), tiles wave and wave + 4 of TL_NP / 32
  auto load_rvp = [&]() {
#pragma unroll
    for (int u = 0; u < (POST ? 2 : 0); ++u)
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = 8 * it + (lane >> 3), gr = row0 + r < a.M ? row0 + r : a.M - 1;
        rvp[u][it] = *(const f32x4*)(a.pres + (long)gr * a.ld_pres + 32 * (wave + 4 * u) + 4 * (lane & 7));
      }
  };
  if constexpr (POST && !SPLIT) load_rvp();  // (SPLIT: requested after the feed-forward, the registers are needed until then)
  const unsigned warm_tok = fd_l2_warm(a.warm, blockIdx.x, gridDim.x, tid, FD_THREADS);
  __syncthreads();
  FD_STAMP(1);
  hx8 X[TL_KS];
  f32x16 acc[3], xa[3];
  const char* xl_base = nullptr;  // SPLIT: this lane's lo fragments of the current stage input (LDS)
  auto x_load = [&](const char* buf) {
#pragma unroll
    for (int s = 0; s < TL_KS; ++s) X[s] = rb_ld(buf + li * TL_XROW + 32 * s + 16 * hi);
    xl_base = buf + XLO + li * TL_XROW + 16 * hi;
  };
  auto layer = [&](const void* img_, const void* img_lo_, auto NTC) {
    constexpr int NT = decltype(NTC)::value, NU = (NT + 3) / 4;
    const char* img = (const char*)img_;
    const char* img_lo = (const char*)img_lo_;
    ch_rb_for<NU>([&](auto U) {
      constexpr int u = decltype(U)::value;
      const int T = wave + 4 * u;
      const bool more = u + 1 < NU && T + 4 < NT;
      if constexpr (SPLIT) {  // Whi.xhi + Whi.xlo out of buffer HB, Wlo.xhi out of buffer LB; the free buffer takes the next hi fragments
        constexpr int HB = (2 * u) % 3, LB = (2 * u + 1) % 3, FB = (2 * u + 2) % 3;
        if (more) w_load(std::integral_constant<int, FB>{}, img, T + 4);
        if (T < NT) {
          f32x16 c;
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
          for (int s = 0; s < TL_KS; ++s) c = fd_mfma32(Wf[HB][s], X[s], c);
          {  // hi x lo: the lo fragments of the activations from LDS, three reads in flight
            hx8 xr[3];
            xr[0] = rb_ld(xl_base);
            xr[1] = rb_ld(xl_base + 32);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < TL_KS; ++s) {
              if (s + 2 < TL_KS) xr[(s + 2) % 3] = rb_ld(xl_base + 32 * (s + 2));
              c = fd_mfma32(Wf[HB][s], xr[s % 3], c);
              __builtin_amdgcn_sched_barrier(0);  // pin: one read, one product per k-step (hipcc otherwise sinks the reads to their uses)
            }
          }
          if (more) w_load(std::integral_constant<int, HB>{}, img_lo, T + 4);
#pragma unroll
          for (int s = 0; s < TL_KS; ++s) c = fd_mfma32(Wf[LB][s], X[s], c);
          acc[u] = c;
        }
      } else {
        if (more) {
          if (u & 1) w_load(std::integral_constant<int, 0>{}, img, T + 4);
          else w_load(std::integral_constant<int, 1>{}, img, T + 4);
        }
        if (T < NT) {
          f32x16 c;
#pragma unroll
          for (int r = 0; r < 16; ++r) c[r] = 0.f;
#pragma unroll
          for (int s = 0; s < TL_KS; ++s) c = fd_mfma32(Wf[u & 1][s], X[s], c);
          acc[u] = c;
        }
      }
    });
  };
  constexpr std::integral_constant<int, TL_NT> NT_D{};
  // LayerNorm of the rows held as acc[] (lane = row, this wave's tiles) -> normalised values back in acc[]
  auto layernorm = [&](const float* gam, const float* bet) {
    // one pass: sum and sum of squares together -> ONE exchange between the lane halves and the four waves
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < 3; ++u)
      if (wave + 4 * u < TL_NT)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          s1 += acc[u][r];
          s2 += acc[u][r] * acc[u][r];
        }
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    __syncthreads();  // red[] may still be read by a slower wave from the previous LayerNorm
    if (hi == 0) {
      red[0][wave][li] = s1;
      red[1][wave][li] = s2;
    }
    __syncthreads();
    const float mu = (red[0][0][li] + red[0][1][li] + red[0][2][li] + red[0][3][li]) * (1.0f / TL_D);
    const float ex2 = (red[1][0][li] + red[1][1][li] + red[1][2][li] + red[1][3][li]) * (1.0f / TL_D);
    const float rstd = 1.0f / sqrtf(fmaxf(ex2 - mu * mu, 0.f) + 1e-5f);
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int T = wave + 4 * u;
      if (T < TL_NT)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f0 = 32 * T + 8 * g + 4 * hi;
          const f32x4 gm = *(const f32x4*)(gam + f0), bt = *(const f32x4*)(bet + f0);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[u][4 * g + q] = (acc[u][4 * g + q] - mu) * rstd * gm[q] + bt[q];
        }
    }
  };
  x_load(xs);
  // ---- stage 1: out_proj + x, LayerNorm1
  layer(a.wo, a.wol, NT_D);
  FD_STAMP(2);
  w_first(a.w1, a.w1l);  // first feed-forward tile: in flight across the barriers
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int T = wave + 4 * u;
    if (T < TL_NT) {
#pragma unroll
      for (int it = 0; it < 4; ++it) *(f32x4*)(stg + (8 * it + (lane >> 3)) * RB_SROW + 16 * (lane & 7)) = rv[u][it];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *(const f32x4*)(cst + 32 * T + 8 * g + 4 * hi);
        const f32x4 rr = *(const f32x4*)(stg + li * RB_SROW + (8 * g + 4 * hi) * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[u][4 * g + q] += bv[q] + rr[q];
      }
    }
  }
  layernorm(cst + TL_D, cst + 2 * TL_D);
  FD_STAMP(3);
  // x_a: fp32 in registers (residual of stage 2), bf16 rows -> LDS (input of stage 2)
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int T = wave + 4 * u;
    xa[u] = acc[u];
    if (T < TL_NT)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f0 = 32 * T + 8 * g + 4 * hi;
        put4(hs, li * TL_XROW + 2 * f0, acc[u][4 * g], acc[u][4 * g + 1], acc[u][4 * g + 2], acc[u][4 * g + 3]);
      }
  }
  __syncthreads();
  x_load(hs);
  // ---- stage 2: feed-forward
  FD_STAMP(4);
  layer(a.w1, a.w1l, NT_D);
  FD_STAMP(5);
  w_first(a.w2, a.w2l);
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int T = wave + 4 * u;
    if (T < TL_NT)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int f0 = 32 * T + 8 * g + 4 * hi;
        const f32x4 bv = *(const f32x4*)(cst + 3 * TL_D + f0);
        // (the att rows are dead: every wave read them before LayerNorm1's barriers)
        put4(xs, li * TL_XROW + 2 * f0, fmaxf(acc[u][4 * g] + bv[0], 0.f), fmaxf(acc[u][4 * g + 1] + bv[1], 0.f),
             fmaxf(acc[u][4 * g + 2] + bv[2], 0.f), fmaxf(acc[u][4 * g + 3] + bv[3], 0.f));
      }
  }
  __syncthreads();
  x_load(xs);
  FD_STAMP(6);
  layer(a.w2, a.w2l, NT_D);
  FD_STAMP(7);
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int T = wave + 4 * u;
    if (T < TL_NT)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *(const f32x4*)(cst + 4 * TL_D + 32 * T + 8 * g + 4 * hi);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[u][4 * g + q] += bv[q] + xa[u][4 * g + q];
      }
  }
  if constexpr (POST && SPLIT) load_rvp();
  if constexpr (POST) w_first(a.wp, a.wpl);  // first post_tfmr tile: in flight across the LayerNorm
  layernorm(cst + 5 * TL_D, cst + 6 * TL_D);
  FD_STAMP(8);
  if constexpr (POST) {
    // ---- stage 4: post_tfmr on the layer's output rows (bf16 through LDS, as every stage input) + bias + node rows
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int T = wave + 4 * u;
      if (T < TL_NT)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f0 = 32 * T + 8 * g + 4 * hi;
          // (the x_a rows are dead since stage 2 read them)
          put4(hs, li * TL_XROW + 2 * f0, acc[u][4 * g], acc[u][4 * g + 1], acc[u][4 * g + 2], acc[u][4 * g + 3]);
        }
    }
    __syncthreads();
    x_load(hs);
    layer(a.wp, a.wpl, std::integral_constant<int, TL_NP / 32>{});
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int T = wave + 4 * u;
#pragma unroll
      for (int it = 0; it < 4; ++it) *(f32x4*)(stg + (8 * it + (lane >> 3)) * RB_SROW + 16 * (lane & 7)) = rvp[u][it];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = *(const f32x4*)(cst + 7 * TL_D + 32 * T + 8 * g + 4 * hi);
        const f32x4 rr = *(const f32x4*)(stg + li * RB_SROW + (8 * g + 4 * hi) * 4);
        f32x4 o;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = acc[u][4 * g + q] + bv[q] + rr[q];
        *(f32x4*)(stg + li * RB_SROW + (8 * g + 4 * hi) * 4) = o;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = 8 * it + (lane >> 3);
        const f32x4 o = *(const f32x4*)(stg + r * RB_SROW + 16 * (lane & 7));
        if (row0 + r < a.M) *(f32x4*)(a.pout + (long)(row0 + r) * a.ld_pout + 32 * T + 4 * (lane & 7)) = o;
      }
    }
    fd_l2_warm_done(warm_tok);
    return;
  }
  // ---- x_b rows out through the wave's tile (128 B row segments)
#pragma unroll
  for (int u = 0; u < 3; ++u) {
    const int T = wave + 4 * u;
    if (T < TL_NT) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 o = {acc[u][4 * g], acc[u][4 * g + 1], acc[u][4 * g + 2], acc[u][4 * g + 3]};
        *(f32x4*)(stg + li * RB_SROW + (8 * g + 4 * hi) * 4) = o;
      }
#pragma unroll
      for (int it = 0; it < 4; ++it) {
        const int r = 8 * it + (lane >> 3);
        const f32x4 o = *(const f32x4*)(stg + r * RB_SROW + 16 * (lane & 7));
        if (row0 + r < a.M) *(f32x4*)(a.out + (long)(row0 + r) * a.ld + 32 * T + 4 * (lane & 7)) = o;
      }
    }
  }
  fd_l2_warm_done(warm_tok);
}

// ------------------------------------------------------------------ the same on 16-row blocks (round 3, split operands only)
// A stage of the 32-row kernel is bound by the CU's L2 path (400 KB of hi + lo fragments at 64 B/clk = 6.4 k cycles) next to 5.8 k cycles of
// matrix work, 75 blocks on 256 CUs.  With 16 rows per block the same bytes feed half the matrix work on twice the CUs
// (v_mfma_f32_16x16x32_f16: tiles of 16 features, k-steps of 32; prototype tools/micro/stage16_bench.hip: a stage 12.3 k -> 7.6 k cycles,
// the input staging 7.3 k -> 4.2 k).  Lane = (row lane % 16, feature group lane / 16): D[feature 16 T + 4 fg + i, row]; a lane's four
// consecutive features make its residual / output accesses 16 B pieces (64 B per row and instruction), so no exchange tile is needed;
// LayerNorm statistics are lane-local sums + the three other feature groups (lane ^ 16, ^ 32) + the four waves through LDS.
#ifndef RB16_BLOCKS
#define RB16_BLOCKS 1  // blocks per CU the 16-row kernels are compiled for.  2 (256 registers; tfmr_tail16 then spills 230 - 300 dwords, mlp16<..ETR> 58): 64 samples
#endif                 // 15.08 against 14.01 ms per step, eight samples 2.45 against 2.09 (round 6, gpurun_out/r6k) - A/B builds only
#define T16_KS (TL_D / 32)
#define T16_NT (TL_D / 16)
#define T16_XROW (TL_D * 2 + 32)  // 42 chunks of 16 B: the b128 fragment reads of 16 rows x 4 feature groups are conflict-free (41: 2-way, SQ_LDS_BANK_CONFLICT 44 %)
#define T16_XLO (16 * T16_XROW)
#define T16_SMEM (4 * T16_XLO + (7 * TL_D + TL_NP) * 4 + 2 * 4 * 16 * 4 + 16)
template <bool POST>
__global__ __launch_bounds__(FD_THREADS, RB16_BLOCKS) void tfmr_tail16_kernel(TfmrTailArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;                       // att rows, then hidden rows: hi rows | lo rows   [2][16][T16_XROW]
  char* hs = xs + 2 * T16_XLO;           // x_a rows
  float* cst = (float*)(hs + 2 * T16_XLO);  // b_o | g1 | be1 | b1 | b2 | g2 | be2 | b_post
  float (*red)[4][16] = (float (*)[4][16])(cst + 7 * TL_D + TL_NP);  // [2][4][16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, fg = lane >> 4;
  const int row0 = blockIdx.x * 16;
  const long grow = row0 + lr < a.M ? row0 + lr : a.M - 1;
  FD_STAMP(0);
  hx8 Wh[2][T16_KS], Wl[2][T16_KS];
  auto w_load = [&](auto BUF, const void* img, const void* img_lo, int T) {
    constexpr int b = decltype(BUF)::value;
#pragma unroll
    for (int s = 0; s < T16_KS; ++s) {
      Wh[b][s] = rb_ld((const char*)img + ((size_t)(T * T16_KS + s) * 64 + lane) * 16);
      Wl[b][s] = rb_ld((const char*)img_lo + ((size_t)(T * T16_KS + s) * 64 + lane) * 16);
    }
  };
  auto put4 = [&](char* buf, int off, float v0, float v1, float v2, float v3) {
    const float v[4] = {v0, v1, v2, v3};
    rb_hx4 pk, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pk[q] = (fd_h)v[q];
      pl[q] = (fd_h)(v[q] - (float)pk[q]);
    }
    *(rb_hx4*)(buf + off) = pk;
    *(rb_hx4*)(buf + T16_XLO + off) = pl;
  };
  constexpr std::integral_constant<int, 0> B0{};
  constexpr std::integral_constant<int, 1> B1{};
  w_load(B0, a.wo, a.wol, wave);
  {
    f32x4 xv[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / 80, c4 = idx % 80;
      const int gr = row0 + r < a.M ? row0 + r : a.M - 1;
      xv[k] = *(const f32x4*)(a.att + (long)gr * a.ld + 4 * c4);
    }
    {
      constexpr int NC = 7 * TL_D + (POST ? TL_NP : 0), NCV = (NC + FD_THREADS - 1) / FD_THREADS;
      float cv[NCV];
#pragma unroll
      for (int k = 0; k < NCV; ++k) {
        const int v = tid + k * FD_THREADS, which = v / TL_D, c = v % TL_D;
        const float* src = which == 0 ? a.bo : which == 1 ? a.g1 : which == 2 ? a.be1 : which == 3 ? a.b1 : which == 4 ? a.b2 : which == 5 ? a.g2 : a.be2;
        cv[k] = v < 7 * TL_D ? src[c] : (POST && v < NC ? a.bp[v - 7 * TL_D] : 0.f);
      }
#pragma unroll
      for (int k = 0; k < NCV; ++k)
        if (tid + k * FD_THREADS < NC) cst[tid + k * FD_THREADS] = cv[k];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / 80, c4 = idx % 80;
      put4(xs, r * T16_XROW + 8 * c4, xv[k][0], xv[k][1], xv[k][2], xv[k][3]);
    }
  }
  // residual x: this lane's four features of its five tiles (tile T = wave + 4 u)
  f32x4 rv[5];
#pragma unroll
  for (int u = 0; u < 5; ++u) rv[u] = *(const f32x4*)(a.x + grow * a.ld + 16 * (wave + 4 * u) + 4 * fg);
  const unsigned warm_tok = fd_l2_warm(a.warm, blockIdx.x, gridDim.x, tid, FD_THREADS);
  __syncthreads();
  FD_STAMP(1);
  hx8 X[T16_KS];
  const char* xl_base = nullptr;  // this lane's lo fragments of the current stage input (read from LDS by the hi x lo pass: 40 registers less)
  auto x_load = [&](const char* buf) {
#pragma unroll
    for (int s = 0; s < T16_KS; ++s) X[s] = rb_ld(buf + lr * T16_XROW + (32 * s + 8 * fg) * 2);
    xl_base = buf + T16_XLO + lr * T16_XROW + 16 * fg;
  };
  f32x4 acc[5], xa[5];
  auto mma = [](hx8 w, hx8 x, f32x4 c) {
    return fd_mfma16(w, x, c);
  };
  // one stage: tiles wave, wave + 4, ... (< NT); the first tile's fragments are in buffer 0, tile u + 1 is requested when tile u starts
  auto layer = [&](const void* img, const void* img_lo, auto NTC) {
    constexpr int NT = decltype(NTC)::value, NU = NT / 4;
    ch_rb_for<NU>([&](auto U) {
      constexpr int u = decltype(U)::value, b = u & 1;
      const int T = wave + 4 * u;
      if constexpr (u + 1 < NU) {
        if constexpr (b == 0) w_load(B1, img, img_lo, T + 4);
        else w_load(B0, img, img_lo, T + 4);
      }
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < T16_KS; ++s) c = mma(Wh[b][s], X[s], c);
      {
        hx8 xr[3];
        xr[0] = rb_ld(xl_base);
        xr[1] = rb_ld(xl_base + 64);
#pragma unroll
        for (int s = 0; s < T16_KS; ++s) {
          if (s + 2 < T16_KS) xr[(s + 2) % 3] = rb_ld(xl_base + 64 * (s + 2));
          c = mma(Wh[b][s], xr[s % 3], c);
        }
      }
#pragma unroll
      for (int s = 0; s < T16_KS; ++s) c = mma(Wl[b][s], X[s], c);
      acc[u] = c;
      // tile boundary.  hipcc's schedule of this loop is fickle: left alone it interleaves the tiles' products and loads and a stage takes
      // 10.8 k cycles; with a compiler-level fence here (an LDS-counter wait that nothing is waiting for + a memory clobber) 6.5 k: 23.0 / 26.6
      // -> 19.4 / 23.6 us per call in the forward.  (`sched_barrier(0)` alone: 21.2 / 37.6; the same fence in mlp16_kernel: no gain.)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    });
  };
  constexpr std::integral_constant<int, T16_NT> NT_D{};
  auto layernorm = [&](const float* gam, const float* bet) {
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int u = 0; u < 5; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        s1 += acc[u][i];
        s2 += acc[u][i] * acc[u][i];
      }
    s1 += __shfl_xor(s1, 16, 64); s2 += __shfl_xor(s2, 16, 64);
    s1 += __shfl_xor(s1, 32, 64); s2 += __shfl_xor(s2, 32, 64);
    __syncthreads();  // red[] may still be read by a slower wave from the previous LayerNorm
    if (fg == 0) {
      red[0][wave][lr] = s1;
      red[1][wave][lr] = s2;
    }
    __syncthreads();
    const float mu = (red[0][0][lr] + red[0][1][lr] + red[0][2][lr] + red[0][3][lr]) * (1.0f / TL_D);
    const float ex2 = (red[1][0][lr] + red[1][1][lr] + red[1][2][lr] + red[1][3][lr]) * (1.0f / TL_D);
    const float rstd = 1.0f / sqrtf(fmaxf(ex2 - mu * mu, 0.f) + 1e-5f);
#pragma unroll
    for (int u = 0; u < 5; ++u) {
      const int f0 = 16 * (wave + 4 * u) + 4 * fg;
      const f32x4 gm = *(const f32x4*)(gam + f0), bt = *(const f32x4*)(bet + f0);
#pragma unroll
      for (int i = 0; i < 4; ++i) acc[u][i] = (acc[u][i] - mu) * rstd * gm[i] + bt[i];
    }
  };
  x_load(xs);
  // ---- stage 1: out_proj + x, LayerNorm1
  layer(a.wo, a.wol, NT_D);
  FD_STAMP(2);
  w_load(B0, a.w1, a.w1l, wave);  // first feed-forward tile: in flight across the barriers
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const f32x4 bv = *(const f32x4*)(cst + 16 * (wave + 4 * u) + 4 * fg);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[u][i] += bv[i] + rv[u][i];
  }
  layernorm(cst + TL_D, cst + 2 * TL_D);
  FD_STAMP(3);
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    xa[u] = acc[u];
    put4(hs, lr * T16_XROW + 2 * (16 * (wave + 4 * u) + 4 * fg), acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
  }
  __syncthreads();
  x_load(hs);
  FD_STAMP(4);
  // ---- stage 2: feed-forward
  layer(a.w1, a.w1l, NT_D);
  FD_STAMP(5);
  w_load(B0, a.w2, a.w2l, wave);
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const int f0 = 16 * (wave + 4 * u) + 4 * fg;
    const f32x4 bv = *(const f32x4*)(cst + 3 * TL_D + f0);
    // (the att rows are dead: every wave read them before LayerNorm1's barriers)
    put4(xs, lr * T16_XROW + 2 * f0, fmaxf(acc[u][0] + bv[0], 0.f), fmaxf(acc[u][1] + bv[1], 0.f), fmaxf(acc[u][2] + bv[2], 0.f),
         fmaxf(acc[u][3] + bv[3], 0.f));
  }
  __syncthreads();
  x_load(xs);
  FD_STAMP(6);
  layer(a.w2, a.w2l, NT_D);
  FD_STAMP(7);
#pragma unroll
  for (int u = 0; u < 5; ++u) {
    const f32x4 bv = *(const f32x4*)(cst + 4 * TL_D + 16 * (wave + 4 * u) + 4 * fg);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[u][i] += bv[i] + xa[u][i];
  }
  f32x4 rvp[4];
  if constexpr (POST) {
    w_load(B0, a.wp, a.wpl, wave);  // first post_tfmr tile: in flight across the LayerNorm
  }
  layernorm(cst + 5 * TL_D, cst + 6 * TL_D);
  FD_STAMP(8);
  if constexpr (POST) {
    // ---- stage 4: post_tfmr on the layer's output rows + bias + node rows
#pragma unroll
    for (int u = 0; u < 5; ++u)  // (the x_a rows are dead since stage 2 read them)
      put4(hs, lr * T16_XROW + 2 * (16 * (wave + 4 * u) + 4 * fg), acc[u][0], acc[u][1], acc[u][2], acc[u][3]);
    __syncthreads();
    x_load(hs);
#pragma unroll
    for (int u = 0; u < 4; ++u) rvp[u] = *(const f32x4*)(a.pres + grow * a.ld_pres + 16 * (wave + 4 * u) + 4 * fg);  // (under the stage's products)
    layer(a.wp, a.wpl, std::integral_constant<int, TL_NP / 16>{});
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f0 = 16 * (wave + 4 * u) + 4 * fg;
      const f32x4 bv = *(const f32x4*)(cst + 7 * TL_D + f0);
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = acc[u][i] + bv[i] + rvp[u][i];
      if (row0 + lr < a.M) *(f32x4*)(a.pout + (long)(row0 + lr) * a.ld_pout + f0) = o;
    }
    fd_l2_warm_done(warm_tok);
    return;
  }
#pragma unroll
  for (int u = 0; u < 5; ++u)
    if (row0 + lr < a.M) *(f32x4*)(a.out + (long)(row0 + lr) * a.ld + 16 * (wave + 4 * u) + 4 * fg) = acc[u];
  fd_l2_warm_done(warm_tok);
}

// ------------------------------------------------------------------ StructureModuleTransition + LayerNorm + mask + BackboneUpdate on 16-row blocks
// rowblock_kernel<256,256,256,256, relu | relu | LN | BB | SPLIT> (FD_RB_TRANSITION_BB_SPLIT) in the shape of tfmr_tail16_kernel: three stages
// 256 -> 256 (ReLU, ReLU, -), + residual, LayerNorm, row mask, then BackboneUpdate (Linear 256 -> 6) and compose_q_update_vec in place
// (ipa_pytorch.py:365-413,542-545; rigid_utils.py:587-616,1039-1063).  Images: fd_chain_build_image16 (hi, lo) of the three matrices.
#define TR_D 256
#define TR_KS (TR_D / 32)
#define TR_NT (TR_D / 16)
#define TR_XROW (TR_D * 2 + 32)  // 34 chunks of 16 B: conflict-free b128 fragment reads (see T16_XROW)
#define TR_XLO (16 * TR_XROW)
#define TR_NC (11 * TR_D)  // b0 | b1 | b2 | gamma | beta | Wbb [6][256]
#define TR_SMEM (4 * TR_XLO + TR_NC * 4 + 2 * 4 * 16 * 4 + 4 * 16 * 8 * 4 + 64 + 16)
// KS0: 32-wide k-steps of the first layer (input width a.k0 <= 32 KS0, zero-padded in LDS); NL: 2 or 3 layers (256 wide; ReLU behind every layer
// but the last); LN: LayerNorm on the output; BB: BackboneUpdate + compose.  <8, 3, true, true> = the transition, <3, 3, true, false> = the node
// embedder (72 / 88 input features), <8, 2, false, false> = the torsion head's residual block.
// SKIP: one more 256 -> 256 layer (w3 / w3l / b3) on the output rows -> out2.
// ETR: the EdgeTransition row launch folded in (RowBlockArgs.we0 ...): two more stages on the output rows — e = initial_embed(row)
// (256 -> 128, transposed products like the stages above) and its 1024 fold columns (128 -> 1024) with the MFMA operands exchanged
// (lane = output column, registers = four consecutive rows: 8 B pieces of edge_transition4's fold-fragment images, the layout
// rowblock_kernel's IMG epilogue writes from 32-row tiles).  Saves the launch of rowblock_kernel<256,128,0,1024,IMG> (16 us at 2400
// rows, three times per forward) for ~0.65 MB more weight fragments per block.
template <int KS0, int NL, bool LN, bool BB, bool SKIP = false, bool ETR = false>
__global__ __launch_bounds__(FD_THREADS, RB16_BLOCKS) void mlp16_kernel(RowBlockArgs a, int k0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* xs = smem;
  char* hs = xs + 2 * TR_XLO;
  float* cst = (float*)(hs + 2 * TR_XLO);
  float (*red)[4][16] = (float (*)[4][16])(cst + TR_NC);  // [2][4][16]
  float* red3 = (float*)(red + 2);                         // [4][16][8] partial BackboneUpdate outputs
  float* pmask = red3 + 4 * 16 * 8;                        // [16]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, lr = lane & 15, fg = lane >> 4;
  const int row0 = blockIdx.x * 16;
  const long grow = row0 + lr < a.M ? row0 + lr : a.M - 1;
  hx8 Wh[2][TR_KS], Wl[2][TR_KS];
  auto w_load = [&](auto BUF, auto KSC, const void* img, const void* img_lo, int T) {
    constexpr int b = decltype(BUF)::value, KS = decltype(KSC)::value;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      Wh[b][s] = rb_ld((const char*)img + ((size_t)(T * KS + s) * 64 + lane) * 16);
      Wl[b][s] = rb_ld((const char*)img_lo + ((size_t)(T * KS + s) * 64 + lane) * 16);
    }
  };
  constexpr std::integral_constant<int, KS0> K0C{};
  constexpr std::integral_constant<int, TR_KS> K8C{};
  auto put4 = [&](char* buf, int off, float v0, float v1, float v2, float v3) {
    const float v[4] = {v0, v1, v2, v3};
    rb_hx4 pk, pl;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      pk[q] = (fd_h)v[q];
      pl[q] = (fd_h)(v[q] - (float)pk[q]);
    }
    *(rb_hx4*)(buf + off) = pk;
    *(rb_hx4*)(buf + TR_XLO + off) = pl;
  };
  constexpr std::integral_constant<int, 0> B0{};
  constexpr std::integral_constant<int, 1> B1{};
  w_load(B0, K0C, a.w0, a.w0l, wave);
  {
    // input rows: k0 / 4 float4 per row (k0 % 4 == 0), columns beyond k0 up to 32 KS0 are zeros
    constexpr int C4 = KS0 * 8, NV = (16 * C4 + FD_THREADS - 1) / FD_THREADS;
    f32x4 xv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / C4, c4 = idx % C4;
      const int gr = row0 + r < a.M ? row0 + r : a.M - 1;
      xv[k] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (idx < 16 * C4 && 4 * c4 < k0) xv[k] = *(const f32x4*)(a.in + (long)gr * a.ld_in + 4 * c4);
    }
    {
      constexpr int NCV = TR_NC / FD_THREADS;
      float cv[NCV];
#pragma unroll
      for (int k = 0; k < NCV; ++k) {
        const int v = tid + k * FD_THREADS, which = v / TR_D, c = v % TR_D;
        cv[k] = which == 0 ? a.b0[c] : which == 1 ? a.b1[c] : which == 2 ? (NL == 3 ? a.b2[c] : 0.f) : which == 3 ? (LN ? a.gamma[c] : 1.f)
                : which == 4 ? (LN ? a.beta[c] : 0.f) : (BB ? a.bb_w[v - 5 * TR_D] : 0.f);
      }
#pragma unroll
      for (int k = 0; k < NCV; ++k) cst[tid + k * FD_THREADS] = cv[k];
    }
    if (tid < 16) pmask[tid] = a.rowmask_post ? a.rowmask_post[row0 + tid < a.M ? row0 + tid : a.M - 1] : 1.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / C4, c4 = idx % C4;
      if (idx < 16 * C4) put4(xs, r * TR_XROW + 8 * c4, xv[k][0], xv[k][1], xv[k][2], xv[k][3]);
    }
  }
  f32x4 rv[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    rv[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (a.residual) rv[u] = *(const f32x4*)(a.residual + grow * a.ld_res + 16 * (wave + 4 * u) + 4 * fg);
  }
  const unsigned warm_tok = fd_l2_warm(a.warm, blockIdx.x, gridDim.x, tid, FD_THREADS);
  __syncthreads();
  hx8 X[TR_KS];
  const char* xl_base = nullptr;
  auto x_load = [&](auto KSC, const char* buf) {
#pragma unroll
    for (int s = 0; s < decltype(KSC)::value; ++s) X[s] = rb_ld(buf + lr * TR_XROW + (32 * s + 8 * fg) * 2);
    xl_base = buf + TR_XLO + lr * TR_XROW + 16 * fg;
  };
  f32x4 acc[4];
  auto mma = [](hx8 w, hx8 x, f32x4 c) {
    return fd_mfma16(w, x, c);
  };
  auto layer = [&](auto KSC, const void* img, const void* img_lo) {
    constexpr int KS = decltype(KSC)::value;
    ch_rb_for<4>([&](auto U) {
      constexpr int u = decltype(U)::value, b = u & 1;
      const int T = wave + 4 * u;
      if constexpr (u + 1 < 4) {
        if constexpr (b == 0) w_load(B1, KSC, img, img_lo, T + 4);
        else w_load(B0, KSC, img, img_lo, T + 4);
      }
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < KS; ++s) c = mma(Wh[b][s], X[s], c);
      {
        hx8 xr[3];
        xr[0] = rb_ld(xl_base);
        if (KS > 1) xr[1] = rb_ld(xl_base + 64);
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          if (s + 2 < KS) xr[(s + 2) % 3] = rb_ld(xl_base + 64 * (s + 2));
          c = mma(Wh[b][s], xr[s % 3], c);
        }
      }
#pragma unroll
      for (int s = 0; s < KS; ++s) c = mma(Wl[b][s], X[s], c);
      acc[u] = c;
    });
  };
  auto relu_to = [&](char* dst, const float* bias) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f0 = 16 * (wave + 4 * u) + 4 * fg;
      const f32x4 bv = *(const f32x4*)(bias + f0);
      put4(dst, lr * TR_XROW + 2 * f0, fmaxf(acc[u][0] + bv[0], 0.f), fmaxf(acc[u][1] + bv[1], 0.f), fmaxf(acc[u][2] + bv[2], 0.f),
           fmaxf(acc[u][3] + bv[3], 0.f));
    }
  };
  x_load(K0C, xs);
  layer(K0C, a.w0, a.w0l);
  w_load(B0, K8C, a.w1, a.w1l, wave);
  relu_to(hs, cst);
  __syncthreads();
  x_load(K8C, hs);
  layer(K8C, a.w1, a.w1l);
  if constexpr (NL == 3) {
    w_load(B0, K8C, a.w2, a.w2l, wave);
    relu_to(xs, cst + TR_D);  // (the input rows are dead: every wave read them before the barrier above)
    __syncthreads();
    x_load(K8C, xs);
    layer(K8C, a.w2, a.w2l);
  }
  // ---- + bias + residual, LayerNorm, mask
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const f32x4 bv = *(const f32x4*)(cst + (NL - 1) * TR_D + 16 * (wave + 4 * u) + 4 * fg);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float v = acc[u][i] + bv[i] + rv[u][i];
      acc[u][i] = v;
      s1 += v;
    }
  }
  float mu = 0.f, rstd = 1.f;
  if constexpr (LN) {  // (two passes like the 32-row kernel: mean first, then the centred sum of squares)
    s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
    if (fg == 0) red[0][wave][lr] = s1;
    __syncthreads();
    mu = (red[0][0][lr] + red[0][1][lr] + red[0][2][lr] + red[0][3][lr]) * (1.0f / TR_D);
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float dlt = acc[u][i] - mu;
        s2 += dlt * dlt;
      }
    s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
    if (fg == 0) red[1][wave][lr] = s2;
    __syncthreads();
    rstd = 1.0f / sqrtf((red[1][0][lr] + red[1][1][lr] + red[1][2][lr] + red[1][3][lr]) * (1.0f / TR_D) + 1e-5f);
  }
  const float pm = pmask[lr];
  float pd[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int f0 = 16 * (wave + 4 * u) + 4 * fg;
    const f32x4 gm = *(const f32x4*)(cst + 3 * TR_D + f0), bt = *(const f32x4*)(cst + 4 * TR_D + f0);
    f32x4 o;
#pragma unroll
    for (int i = 0; i < 4; ++i) o[i] = LN ? ((acc[u][i] - mu) * rstd * gm[i] + bt[i]) * pm : acc[u][i] * pm;
    if (row0 + lr < a.M) *(f32x4*)(a.out + (long)(row0 + lr) * a.ld_out + f0) = o;
    if constexpr (BB) {
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const f32x4 wv = *(const f32x4*)(cst + 5 * TR_D + k * TR_D + f0);
        pd[k] += (o[0] * wv[0] + o[1] * wv[1]) + (o[2] * wv[2] + o[3] * wv[3]);
      }
    }
  }
  if constexpr (SKIP || ETR) {  // skip_embed of every trunk block (SKIP) / initial_embed of EdgeTransition (ETR) on the rows just produced (they never leave the CU for it)
    w_load(B0, K8C, SKIP ? a.w3 : a.we0, SKIP ? a.w3l : a.we0l, wave);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f0 = 16 * (wave + 4 * u) + 4 * fg;
      const f32x4 gm = *(const f32x4*)(cst + 3 * TR_D + f0), bt = *(const f32x4*)(cst + 4 * TR_D + f0);
      put4(hs, lr * TR_XROW + 2 * f0, LN ? ((acc[u][0] - mu) * rstd * gm[0] + bt[0]) * pm : acc[u][0] * pm,
           LN ? ((acc[u][1] - mu) * rstd * gm[1] + bt[1]) * pm : acc[u][1] * pm, LN ? ((acc[u][2] - mu) * rstd * gm[2] + bt[2]) * pm : acc[u][2] * pm,
           LN ? ((acc[u][3] - mu) * rstd * gm[3] + bt[3]) * pm : acc[u][3] * pm);  // (hs: dead since the second layer read it)
    }
    __syncthreads();
    x_load(K8C, hs);
    if constexpr (SKIP) {
    layer(K8C, a.w3, a.w3l);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f0 = 16 * (wave + 4 * u) + 4 * fg;
      const f32x4 bv = *(const f32x4*)(a.b3 + f0);
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = acc[u][i] + bv[i];
      if (row0 + lr < a.M) *(f32x4*)(a.out2 + (long)(row0 + lr) * a.ld_out2 + f0) = o;
    }
    }
  }
  if constexpr (ETR) {
    // ---- stage E1: e = initial_embed(out row), 128 features = 8 tiles of 16: tiles wave and wave + 4 (first fragments requested above)
    f32x4 ea[2];
    ch_rb_for<2>([&](auto U) {
      constexpr int u = decltype(U)::value;
      const int T = wave + 4 * u;
      if constexpr (u == 0) w_load(B1, K8C, a.we0, a.we0l, T + 4);
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < TR_KS; ++s) c = mma(Wh[u][s], X[s], c);
      {
        hx8 xr[3];
        xr[0] = rb_ld(xl_base);
        xr[1] = rb_ld(xl_base + 64);
#pragma unroll
        for (int s = 0; s < TR_KS; ++s) {
          if (s + 2 < TR_KS) xr[(s + 2) % 3] = rb_ld(xl_base + 64 * (s + 2));
          c = mma(Wh[u][s], xr[s % 3], c);
        }
      }
#pragma unroll
      for (int s = 0; s < TR_KS; ++s) c = mma(Wl[u][s], X[s], c);
      ea[u] = c;
    });
    constexpr std::integral_constant<int, 4> K4C{};
    // fold-tile fragments (4 hi + 4 lo per tile) travel through a ring of FOUR slots — slot j = half (j >> 1) of buffer (j & 1) —
    // three tiles ahead of their use: with one tile ahead (8 KB per wave in flight) the stage ran at 40 B/clk of the CU's L2 path
    auto e2_load = [&](auto J, int T) {
      constexpr int j = decltype(J)::value, b = j & 1, h = j >> 1;
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        Wh[b][4 * h + s] = rb_ld((const char*)a.we1 + ((size_t)(T * 4 + s) * 64 + lane) * 16);
        Wl[b][4 * h + s] = rb_ld((const char*)a.we1l + ((size_t)(T * 4 + s) * 64 + lane) * 16);
      }
    };
    e2_load(std::integral_constant<int, 2>{}, wave);       // (buffer halves 1: free since E1 used fragments 0..7 of both buffers — loaded
    e2_load(std::integral_constant<int, 3>{}, wave + 4);   //  only after E1's products in program order)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f0 = 16 * (wave + 4 * u) + 4 * fg;
      const f32x4 bv = *(const f32x4*)(a.be0 + f0);
      put4(xs, lr * TR_XROW + 2 * f0, ea[u][0] + bv[0], ea[u][1] + bv[1], ea[u][2] + bv[2], ea[u][3] + bv[3]);  // (xs: dead since the last transition layer read it)
    }
    e2_load(std::integral_constant<int, 0>{}, wave + 8);
    __syncthreads();
    // ---- stage E2: fold columns, 64 tiles of 16: tiles wave + 4 u, operands exchanged: D[row 4 fg + i, column 16 T + lr]
    x_load(K4C, xs);
    const int NJ4 = a.img_N >> 2, r0 = row0 + 4 * fg;  // rows r0 .. r0 + 3 (M % 4 == 0: all four valid or none)
    half_t* ia = (half_t*)a.img_a;
    half_t* ib = (half_t*)a.img_b;
    const int rb_ = r0 < a.M ? r0 / a.img_N : 0, jt_ = r0 < a.M ? (r0 - rb_ * a.img_N) >> 2 : 0;
    float bvs[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) bvs[u] = a.be1[16 * (wave + 4 * u) + lr];
    ch_rb_for<16>([&](auto U) {
      constexpr int u = decltype(U)::value, j = (u + 2) & 3, b = j & 1, h = j >> 1;  // tile u sits in slot (u + 2) % 4
      const int T = wave + 4 * u;
      if constexpr (u + 3 < 16) e2_load(std::integral_constant<int, (u + 5) & 3>{}, T + 12);  // the slot tile u - 1 has just left
      f32x4 c = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int s = 0; s < 4; ++s) c = mma(X[s], Wh[b][4 * h + s], c);
#pragma unroll
      for (int s = 0; s < 4; ++s) c = mma(rb_ld(xl_base + 64 * s), Wh[b][4 * h + s], c);
#pragma unroll
      for (int s = 0; s < 4; ++s) c = mma(X[s], Wl[b][4 * h + s], c);
      const float bv = bvs[u];
      const int T32 = T >> 1, f = 16 * (T & 1) + lr;
      u16x4 o = {0, 0, 0, 0};
      if (r0 < a.M) o = u16x4{f2h(c[0] + bv), f2h(c[1] + bv), f2h(c[2] + bv), f2h(c[3] + bv)};
      if (T32 < 16) {
        if ((r0 >> 3) < ((a.M + 7) >> 3)) *(u16x4*)(ia + ((((long)(r0 >> 3) * 16 + T32) * 32 + f) << 3) + (r0 & 4)) = o;  // (rows beyond M: zeros; the image is padded to 8 rows)
      } else if (r0 < a.M) {
        half_t* dst = ib + ((((long)rb_ * NJ4 + jt_) * 16 + (T32 - 16)) * 32 + f) * 8;
        *(u16x4*)dst = o;
        if (rb_ > 0) *(u16x4*)(dst - (long)NJ4 * 16 * 32 * 8 + 4) = o;
        if (rb_ == a.img_B - 1) *(u16x4*)(dst + 4) = u16x4{0, 0, 0, 0};
      }
    });
  }
  if constexpr (!BB) {
    fd_l2_warm_done(warm_tok);
    return;
  }
  // ---- BackboneUpdate: this lane's partial dots + the other feature groups + the other waves, then compose_q_update_vec in place
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    pd[k] += __shfl_xor(pd[k], 16, 64);
    pd[k] += __shfl_xor(pd[k], 32, 64);
  }
  if (fg == 0) {
#pragma unroll
    for (int k = 0; k < 6; ++k) red3[(wave * 16 + lr) * 8 + k] = pd[k];
  }
  __syncthreads();
  if (tid < 16 && row0 + tid < a.M) {
    const long r = row0 + tid;
    float upd[6];
#pragma unroll
    for (int k = 0; k < 6; ++k)
      upd[k] = ((red3[tid * 8 + k] + red3[(16 + tid) * 8 + k]) + (red3[(32 + tid) * 8 + k] + red3[(48 + tid) * 8 + k])) + a.bb_b[k];
    const float m = a.upd_mask ? a.upd_mask[r] : 1.f;
    const float q0 = a.quat[r * 4], q1 = a.quat[r * 4 + 1], q2 = a.quat[r * 4 + 2], q3 = a.quat[r * 4 + 3];
    const float dq0 = -q1 * upd[0] - q2 * upd[1] - q3 * upd[2];
    const float dq1 = q0 * upd[0] + q2 * upd[2] - q3 * upd[1];
    const float dq2 = q0 * upd[1] - q1 * upd[2] + q3 * upd[0];
    const float dq3 = q0 * upd[2] + q1 * upd[1] - q2 * upd[0];
    const float R0 = q0 * q0 + q1 * q1 - q2 * q2 - q3 * q3, R1 = 2 * q1 * q2 - 2 * q0 * q3, R2 = 2 * q1 * q3 + 2 * q0 * q2;
    const float R3 = 2 * q1 * q2 + 2 * q0 * q3, R4 = q0 * q0 - q1 * q1 + q2 * q2 - q3 * q3, R5 = 2 * q2 * q3 - 2 * q0 * q1;
    const float R6 = 2 * q1 * q3 - 2 * q0 * q2, R7 = 2 * q2 * q3 + 2 * q0 * q1, R8 = q0 * q0 - q1 * q1 - q2 * q2 + q3 * q3;
    const float d0 = R0 * upd[3] + R1 * upd[4] + R2 * upd[5];
    const float d1 = R3 * upd[3] + R4 * upd[4] + R5 * upd[5];
    const float d2 = R6 * upd[3] + R7 * upd[4] + R8 * upd[5];
    const float n0 = q0 + dq0 * m, n1 = q1 + dq1 * m, n2 = q2 + dq2 * m, n3 = q3 + dq3 * m;
    const float nrm = sqrtf(n0 * n0 + n1 * n1 + n2 * n2 + n3 * n3);
    a.quat[r * 4] = n0 / nrm; a.quat[r * 4 + 1] = n1 / nrm; a.quat[r * 4 + 2] = n2 / nrm; a.quat[r * 4 + 3] = n3 / nrm;
    fd_store3(a.trans + r * 3, a.trans[r * 3] + d0 * m, a.trans[r * 3 + 1] + d1 * m, a.trans[r * 3 + 2] + d2 * m);
  }
  fd_l2_warm_done(warm_tok);
}
// images w0 / w1 / w2 (+ lo): fd_chain_build_image16 (the first one with K padded to 32 KS0)
template <int KS0, int NL, bool LN, bool BB, bool SKIP = false, bool ETR = false>
static int mlp16_launch(const RowBlockArgs& a, int k0, hipStream_t st) {
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)mlp16_kernel<KS0, NL, LN, BB, SKIP, ETR>, hipFuncAttributeMaxDynamicSharedMemorySize, TR_SMEM) != hipSuccess) return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  hipLaunchKernelGGL((mlp16_kernel<KS0, NL, LN, BB, SKIP, ETR>), dim3(cdiv(a.M, 16)), dim3(FD_THREADS), TR_SMEM, st, a, k0);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
// FD_RB_TRANSITION_BB_SPLIT on 16-row blocks
int fd_transition16(const RowBlockArgs& a, hipStream_t st) {
  if (a.M <= 0 || (a.ld_in & 3) || (a.ld_res & 3) || (a.ld_out & 3) || !a.w0l || !a.w1l || !a.w2l || !a.residual || !a.gamma || !a.beta || !a.bb_w ||
      !a.bb_b || !a.quat || !a.trans)
    return FDIPT_EINVAL;
  if (a.we0) {  // EdgeTransition rows folded in
    if (!a.we0l || !a.we1 || !a.we1l || !a.be0 || !a.be1 || !a.img_a || !a.img_b || (a.img_N & 3) || a.M != a.img_B * a.img_N) return FDIPT_EINVAL;
    return mlp16_launch<8, 3, true, true, false, true>(a, TR_D, st);
  }
  return mlp16_launch<8, 3, true, true>(a, TR_D, st);
}
// FD_RB_NODE_EMBED_72 / 88_SPLIT (k0 input features, k0 % 4 == 0, k0 <= 96) and FD_RB_TORSION_SPLIT on 16-row blocks
int fd_node_embed16(const RowBlockArgs& a, int k0, hipStream_t st) {
  if (a.M <= 0 || (a.ld_in & 3) || (a.ld_out & 3) || (k0 & 3) || k0 > 96 || !a.w0l || !a.w1l || !a.w2l || !a.gamma || !a.beta) return FDIPT_EINVAL;
  if (a.w3) {
    if (!a.w3l || !a.b3 || !a.out2 || (a.ld_out2 & 3)) return FDIPT_EINVAL;
    return mlp16_launch<3, 3, true, false, true>(a, k0, st);
  }
  return mlp16_launch<3, 3, true, false>(a, k0, st);
}
int fd_torsion16(const RowBlockArgs& a, hipStream_t st) {
  if (a.M <= 0 || (a.ld_in & 3) || (a.ld_res & 3) || (a.ld_out & 3) || !a.w0l || !a.w1l) return FDIPT_EINVAL;
  return mlp16_launch<8, 2, false, false>(a, TR_D, st);
}

int fd_tfmr_tail(const TfmrTailArgs& a, hipStream_t st) {
  if (a.M <= 0 || (a.ld & 3) || a.x == a.out) return FDIPT_EINVAL;
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)tfmr_tail_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TL_SMEM(0)) != hipSuccess ||
        hipFuncSetAttribute((const void*)tfmr_tail_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, TL_SMEM(0)) != hipSuccess ||
        hipFuncSetAttribute((const void*)tfmr_tail_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TL_SMEM(1)) != hipSuccess ||
        hipFuncSetAttribute((const void*)tfmr_tail_kernel<true, true>, hipFuncAttributeMaxDynamicSharedMemorySize, TL_SMEM(1)) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  const bool split = a.wol != nullptr;  // split operands: every lo image must be there
  if (split && (!a.w1l || !a.w2l || (a.wp && !a.wpl))) return FDIPT_EINVAL;
  if (a.rows16) {  // 16-row blocks (split operands, fd_chain_build_image16 images)
    if (!split) return FDIPT_EINVAL;
    if (a.wp && (!a.bp || !a.pres || !a.pout || (a.ld_pres & 3) || (a.ld_pout & 3))) return FDIPT_EINVAL;
    static FdPerDevice attr16;
    if (!attr16.get(dev_)) {
      if (hipFuncSetAttribute((const void*)tfmr_tail16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, T16_SMEM) != hipSuccess ||
          hipFuncSetAttribute((const void*)tfmr_tail16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, T16_SMEM) != hipSuccess)
        return FDIPT_ELAUNCH;
      attr16.set(dev_, 1);
    }
    if (a.wp) hipLaunchKernelGGL((tfmr_tail16_kernel<true>), dim3(cdiv(a.M, 16)), dim3(FD_THREADS), T16_SMEM, st, a);
    else hipLaunchKernelGGL((tfmr_tail16_kernel<false>), dim3(cdiv(a.M, 16)), dim3(FD_THREADS), T16_SMEM, st, a);
    FD_CHECK_LAUNCH();
    return FDIPT_OK;
  }
  if (a.wp && (!a.bp || !a.pres || !a.pout || (a.ld_pres & 3) || (a.ld_pout & 3))) return FDIPT_EINVAL;
  const dim3 grid(cdiv(a.M, 32)), block(FD_THREADS);
  if (a.wp && split) hipLaunchKernelGGL((tfmr_tail_kernel<true, true>), grid, block, TL_SMEM(1), st, a);
  else if (a.wp) hipLaunchKernelGGL((tfmr_tail_kernel<true, false>), grid, block, TL_SMEM(0), st, a);
  else if (split) hipLaunchKernelGGL((tfmr_tail_kernel<false, true>), grid, block, TL_SMEM(1), st, a);
  else hipLaunchKernelGGL((tfmr_tail_kernel<false, false>), grid, block, TL_SMEM(0), st, a);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

template <int K0, int N1, int N2, int NOUT, int FLAGS>
static int rb_launch(const RowBlockArgs& a, hipStream_t st) {
  using S = RBShape<K0, N1, N2, NOUT, FLAGS>;
  if (a.M <= 0 || (a.ld_in & 3) || (a.residual && (a.ld_res & 3)) || (a.ld_out & 3) || (a.residual && a.residual == a.out && false))
    return FDIPT_EINVAL;
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)rowblock_kernel<K0, N1, N2, NOUT, FLAGS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)S::SMEM) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  hipLaunchKernelGGL((rowblock_kernel<K0, N1, N2, NOUT, FLAGS>), dim3(cdiv(a.M, 32), S::CP), dim3(FD_THREADS), S::SMEM, st, a);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// shapes of the reference network (c_s 256, d_model 320); FDIPT_EINVAL for anything else
int fd_rowblock(int kind, const RowBlockArgs& a, hipStream_t st) {
  switch (kind) {
    case FD_RB_OUTPROJ: return rb_launch<320, 0, 0, 320, 4>(a, st);              // out_proj + residual, LN
    case FD_RB_FFN: return rb_launch<320, 320, 0, 320, 1 | 4>(a, st);            // l1 relu l2 + residual, LN
    case FD_RB_TRANSITION: return rb_launch<256, 256, 256, 256, 1 | 2 | 4>(a, st);  // t1 relu t2 relu t3 + residual, LN, mask
    case FD_RB_TRANSITION_BB: return rb_launch<256, 256, 256, 256, 1 | 2 | 4 | 8>(a, st);  // ... + BackboneUpdate + compose
    // split-operand forms (RowBlockArgs.w0l / w1l / w2l): the node embedder, the transition and the torsion head
    case FD_RB_TRANSITION_BB_SPLIT: return a.w0l && a.w1l && a.w2l ? rb_launch<256, 256, 256, 256, 1 | 2 | 4 | 8 | 32>(a, st) : FDIPT_EINVAL;
    case FD_RB_NODE_EMBED_72_SPLIT: return a.w0l && a.w1l && a.w2l ? rb_launch<72, 256, 256, 256, 1 | 2 | 4 | 32>(a, st) : FDIPT_EINVAL;
    case FD_RB_NODE_EMBED_88_SPLIT: return a.w0l && a.w1l && a.w2l ? rb_launch<88, 256, 256, 256, 1 | 2 | 4 | 32>(a, st) : FDIPT_EINVAL;
    case FD_RB_TORSION_SPLIT: return a.w0l && a.w1l ? rb_launch<256, 256, 0, 256, 1 | 32>(a, st) : FDIPT_EINVAL;
    case FD_RB_NODE_EMBED_72: return rb_launch<72, 256, 256, 256, 1 | 2 | 4>(a, st);
    case FD_RB_NODE_EMBED_88: return rb_launch<88, 256, 256, 256, 1 | 2 | 4>(a, st);
    case FD_RB_TORSION: return rb_launch<256, 256, 0, 256, 1>(a, st);            // l1 relu l2 + residual
    case FD_RB_ET_ROWS: return rb_launch<256, 128, 0, 512, 0>(a, st);            // e = init(node); [A1 | Af] = [W1e; Wfe] e + b
    case FD_RB_ET4_ROWS: return rb_launch<256, 128, 0, 1024, 0>(a, st);          // ... [A1 | Af | B1 | Bf]: e_i and e_j columns
    case FD_RB_ET4_IMAGES:                                                        // ... written as edge_transition4's fold fragments
      if (!a.img_a || !a.img_b || (a.img_N & 3) || a.M != a.img_B * a.img_N) return FDIPT_EINVAL;
      // split operands (w0l / w1l): e = initial_embed(node) and the four per-residue products to fp32 accuracy before the rows are
      // rounded to the fold fragments — these rows are shared by all pairs of a residue, their errors do not average out over keys
      // ... in two column parts ([A1 | Af] and [B1 | Bf]: 150 blocks, four instead of eight output tiles per wave)
      if (a.w0l && a.w1l) return rb_launch<256, 128, 0, 1024, 16 | 32 | 64>(a, st);
      return rb_launch<256, 128, 0, 1024, 16>(a, st);
    default: return FDIPT_EINVAL;
  }
}
