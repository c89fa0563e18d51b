// attention_seq.hip — self-attention of the sequence transformer (nn.MultiheadAttention inside
// nn.TransformerEncoderLayer, framedipt/model/ipa_pytorch.py:433-443,536-538): d_model 320, 4 heads of 80, bf16.
//
// Two launches per layer, no LDS, no barrier:
//   seq_images_kernel  re-lays the fp32 in_proj output [B*N, 3*d] as bf16 MFMA FRAGMENT images (one fragment = one
//                      linear 1 KB load; a row-per-lane gather costs 8x the TA cycles, tools/micro/io_pattern.hip):
//                        Qi, Ki [B,h,Np/32,KS,64,8]  row 32t + (lane & 31), channel 16s + 8(lane >> 5) + e   (Q pre-scaled)
//                        Vi     [B,h,DT,Np/16,64,8]  V transposed: row = channel 32dt + (lane & 31), key position
//                                                    16s + 8(lane >> 5) + e, keys permuted inside every 16-group
//                                                    (perm16: C/D fragment -> B fragment order); pads are zero
//                      channel 80 carries the key padding mask through the matrix cores: Qi[.., 80] = 1,
//                      Ki[key, 80] = -1e30 for masked / padded keys (channels 81..95 are zero)
//   seq_attn_kernel    one block = 32 queries of one (sample, head); the KEYS are dealt round-robin to the 4 waves in
//                      tiles of 32 (S^T[key, query] = Ki_tile * Q^T), softmax = registers + lane^32 shuffle + 2 x 128
//                      floats of LDS across waves, P fragments are exchanged through LDS once and wave w < 3 then owns
//                      channel tile w of O^T[d, query] = Vi * P^T over ALL keys.  Every global operand of a wave is
//                      requested in its first instructions (V included), so the whole block is ONE memory round trip.
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

#define SA_HD 80
#define SA_KS 6         // k-steps of 16 channels: 80 + the mask channel, padded to 96
#define SA_DT 3         // 32-channel output tiles (80 -> 96, padded rows are zero)
#define SA_NTW_MAX 8    // key tiles per wave -> N <= 8 * 4 * 32 = 1024

__host__ __device__ __forceinline__ int sa_perm16(int pos) {  // involution
  const int hi = pos >> 3, e = pos & 7;
  return 4 * hi + (e & 3) + 8 * (e >> 2);
}

__global__ void seq_images_kernel(int B, int N, int Np, int H, const float* __restrict__ qkv, int ld, float qscale,
                                  const float* __restrict__ res_mask,
                                  half_t* __restrict__ Qi, half_t* __restrict__ Ki, half_t* __restrict__ Vi) {
  const int nt = Np >> 5, dm = H * SA_HD;
  const long nqk = (long)B * H * nt * SA_KS * 64, nv = (long)B * H * SA_DT * (2 * nt) * 64;
  for (long u = blockIdx.x * (long)blockDim.x + threadIdx.x; u < 2 * nqk + nv; u += (long)gridDim.x * blockDim.x) {
    u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (u < 2 * nqk) {
      const bool isk = u >= nqk;
      const long v = isk ? u - nqk : u;
      const int lane = (int)(v & 63), s = (int)((v >> 6) % SA_KS);
      const long r2 = (v >> 6) / SA_KS;
      const int t = (int)(r2 % nt);
      const long bh = r2 / nt;
      const int h = (int)(bh % H);
      const long b = bh / H;
      const int row = 32 * t + (lane & 31), c0 = 16 * s + 8 * (lane >> 5);
      if (c0 < SA_HD) {
        if (row < N) {
          const float* src = qkv + (b * N + row) * ld + (isk ? dm : 0) + h * SA_HD + c0;
          const f32x4 x0 = *(const f32x4*)src, x1 = *(const f32x4*)(src + 4);
          const float sc = isk ? 1.f : qscale;
          o = u16x8{f2h(x0[0] * sc), f2h(x0[1] * sc), f2h(x0[2] * sc), f2h(x0[3] * sc),
                    f2h(x1[0] * sc), f2h(x1[1] * sc), f2h(x1[2] * sc), f2h(x1[3] * sc)};
        }
      } else if (c0 == SA_HD) {  // mask channel
        if (!isk) o[0] = f2h(1.0f);
        else if (row >= N || res_mask[b * N + row] == 0.f) o[0] = f2h(FD_H_NEG_BIG);
      }
      *(u16x8*)((isk ? Ki : Qi) + v * 8) = o;
    } else {
      const long v = u - 2 * nqk;
      const int lane = (int)(v & 63), s = (int)((v >> 6) % (2 * nt));
      const long r2 = (v >> 6) / (2 * nt);
      const int dt = (int)(r2 % SA_DT);
      const long bh = r2 / SA_DT;
      const int h = (int)(bh % H);
      const long b = bh / H;
      const int d = 32 * dt + (lane & 31);
      if (d < SA_HD) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int pos = 16 * s + 8 * (lane >> 5) + e, key = (pos & ~15) + sa_perm16(pos & 15);
          if (key < N) o[e] = f2h(qkv[(b * N + key) * ld + 2 * dm + h * SA_HD + d]);
        }
      }
      *(u16x8*)(Vi + v * 8) = o;
    }
  }
}

// ------------------------------------------------------------------ fused in_proj -> images
// seq_images_init_kernel: everything in the images that does not depend on the layer — zeros (padded rows / keys /
// channels 81..95), Q's mask channel = 1, K's mask channel = -1e30 for masked or padded keys.  Once per forward.
// The other once-per-forward fills of the trunk ride on this launch (SeqInitExtra): a plain zero fill (attention3's value-point
// image) and the padded keys [N, Np) of attention3's Kb / Vt images (layouts: gemm.hip, kv_zero_pad_kernel).
__global__ void seq_images_init_kernel(int B, int N, int Np, int H, const float* __restrict__ res_mask, half_t* __restrict__ Qi,
                                       half_t* __restrict__ Ki, half_t* __restrict__ Vi, SeqInitExtra x) {
  const long gtid = blockIdx.x * (long)blockDim.x + threadIdx.x, gsz = (long)gridDim.x * blockDim.x;
  for (long i = gtid; i < x.fill_n16; i += gsz) ((uint4*)x.fill)[i] = make_uint4(0, 0, 0, 0);
  if (x.Kb) {
    const int pad = Np - N, ntl = Np >> 5, cg = x.C >> 3;
    const long n = x.BH * pad * cg;
    const u16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
    half_t* Kb = (half_t*)x.Kb;
    half_t* Vt = (half_t*)x.Vt;
    half_t* Vt2 = (half_t*)x.Vt2;
    for (long i = gtid; i < n; i += gsz) {
      const int g = (int)(i % cg);
      const long r2 = i / cg;
      const int key = N + (int)(r2 % pad);
      const long bh = r2 / pad;
      *(u16x8*)(Kb + ((((bh * ntl + (key >> 5)) * (x.C >> 4) + (g >> 1)) * 64 + (g & 1) * 32 + (key & 31)) << 3)) = z8;
      const int p16 = key & 15, pp = (key & ~15) + 4 * (p16 >> 3) + (p16 & 3) + 8 * ((p16 & 7) >> 2);
      for (int c = 8 * g; c < 8 * g + 8; ++c) {
        const long o = ((((bh * (x.C >> 5) + (c >> 5)) * (2 * ntl) + (pp >> 4)) * 64 + ((pp >> 3) & 1) * 32 + (c & 31)) << 3) + (pp & 7);
        Vt[o] = 0;
        if (Vt2) Vt2[o] = 0;
      }
    }
  }
  const int nt = Np >> 5;
  const long nqk = (long)B * H * nt * SA_KS * 64, nv = (long)B * H * SA_DT * (2 * nt) * 64;
  for (long u = blockIdx.x * (long)blockDim.x + threadIdx.x; u < 2 * nqk + nv; u += (long)gridDim.x * blockDim.x) {
    u16x8 o = {0, 0, 0, 0, 0, 0, 0, 0};
    if (u < 2 * nqk) {
      const bool isk = u >= nqk;
      const long v = isk ? u - nqk : u;
      const int lane = (int)(v & 63), s = (int)((v >> 6) % SA_KS);
      const long r2 = (v >> 6) / SA_KS;
      const int t = (int)(r2 % nt);
      const long b = (r2 / nt) / H;
      const int row = 32 * t + (lane & 31);
      if (s == SA_KS - 1 && (lane >> 5) == 0) {  // channels 80..87: the mask channel is the first of them
        if (!isk) o[0] = f2h(1.0f);
        else if (row >= N || res_mask[b * N + row] == 0.f) o[0] = f2h(FD_H_NEG_BIG);
      }
      *(u16x8*)((isk ? Ki : Qi) + v * 8) = o;
    } else {
      *(u16x8*)(Vi + (u - 2 * nqk) * 8) = o;
    }
  }
}

// seq_qkv_kernel: in_proj (Linear 320 -> 960, fp32 rows in, bf16 MFMA) whose epilogue writes the attention images
// directly.  A block owns 32 rows and all 30 output tiles (dealt to the 4 waves); weights as a natural-order fragment image
// (fd_chain_build_image), read straight from L2.  Q / K tiles run transposed (lane = row: 4 consecutive channels -> 8 B
// pieces of a fragment unit); V tiles run with the operands exchanged (lane = channel, registers = rows: 4 consecutive
// keys -> 8 B pieces of V^T).  Needs N % 4 == 0.
#define SQ_K 320
#define SQ_KS (SQ_K / 16)
#define SQ_XROW (SQ_K * 2 + 16)
typedef fd_h sa_hx4 __attribute__((ext_vector_type(4)));
// SPLIT: the product runs on split operands (x = hi + lo, W = hi + lo: Whi.xhi + Whi.xlo + Wlo.xhi, see rowblock.hip); the
// images still receive half-precision values (their rounding is averaged over the keys by the attention, tests/err_budget.py).
template <bool SPLIT>
__global__ __launch_bounds__(FD_THREADS, 1) void seq_qkv_kernel(int B, int N, int Np, int H, const float* __restrict__ x, int ld_x,
                                                                const char* __restrict__ wimg, const char* __restrict__ wimg_lo,
                                                                const float* __restrict__ bias,
                                                                float qscale, half_t* __restrict__ Qi, half_t* __restrict__ Ki,
                                                                half_t* __restrict__ Vi) {
  __shared__ __attribute__((aligned(16))) char xs[(SPLIT ? 2 : 1) * 32 * SQ_XROW];
  constexpr int XLO = 32 * SQ_XROW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  // 1-D grid, id = (x / 8) * 24 + y * 8 + (x % 8): the three column parts of a row block run on the row block's XCD (x % 8, the
  // XCD whose L2 the producer of the rows — a 32-row-block kernel with the same mapping — wrote them through)
  const int bx = (blockIdx.x / 24) * 8 + (blockIdx.x & 7), by = (blockIdx.x % 24) >> 3;
  const int M = B * N, row0 = bx * 32, nt = Np >> 5, dm = H * SA_HD;
  if (row0 >= M) return;
  hx8 Wf[2][SQ_KS];
  auto w_load = [&](auto BUF, const char* img, int T) {
    constexpr int bf = decltype(BUF)::value;
#pragma unroll
    for (int s = 0; s < SQ_KS; ++s) Wf[bf][s] = __builtin_bit_cast(hx8, *(const u16x8*)(img + ((size_t)(T * SQ_KS + s) * 64 + lane) * 16));
  };
  w_load(std::integral_constant<int, 0>{}, wimg, by * (SQ_K / 32) + wave);
  if constexpr (SPLIT) w_load(std::integral_constant<int, 1>{}, wimg_lo, by * (SQ_K / 32) + wave);
  {
    f32x4 xv[10];
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / 80, c4 = idx % 80;
      const int gr = row0 + r < M ? row0 + r : M - 1;
      xv[k] = *(const f32x4*)(x + (long)gr * ld_x + 4 * c4);
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
      const int idx = tid + k * FD_THREADS, r = idx / 80, c4 = idx % 80;
      sa_hx4 pk, pl;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pk[q] = (fd_h)xv[k][q];
        pl[q] = (fd_h)(xv[k][q] - (float)pk[q]);
      }
      *(sa_hx4*)(xs + r * SQ_XROW + 8 * c4) = pk;
      if constexpr (SPLIT) *(sa_hx4*)(xs + XLO + r * SQ_XROW + 8 * c4) = pl;
    }
  }
  __syncthreads();
  hx8 X[SQ_KS];
  hx8 Xl[SPLIT ? SQ_KS : 1];
#pragma unroll
  for (int s = 0; s < SQ_KS; ++s) {
    X[s] = __builtin_bit_cast(hx8, *(const u16x8*)(xs + li * SQ_XROW + 32 * s + 16 * hi));
    if constexpr (SPLIT) Xl[s] = __builtin_bit_cast(hx8, *(const u16x8*)(xs + XLO + li * SQ_XROW + 32 * s + 16 * hi));
  }
  // this lane's row (transposed tiles) -> sample / key
  const int m = row0 + li, mb = m < M ? m / N : 0, mr = m - mb * N;
  // by = 0 / 1 / 2 takes the Q / K / V third of the 30 output tiles (3x the blocks: the kernel is latency-bound)
  constexpr int NTP = SQ_K / 32;  // 10 tiles per part
  const int T0 = by * NTP;
  // bias of this wave's tiles, requested now (before the MFMAs), in the layout of the part's epilogue: a load issued after a
  // tile's MFMAs is one exposed L2 round trip per tile
  f32x4 bq[(NTP + 3) / 4][4];
#pragma unroll
  for (int u = 0; u < (NTP + 3) / 4; ++u) {
    const int Tb = T0 + (wave + 4 * u < NTP ? wave + 4 * u : wave);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      if (by < 2) bq[u][g] = *(const f32x4*)(bias + 32 * Tb + 8 * g + 4 * hi);
      else if (g == 0) bq[u][0][0] = bias[32 * Tb + li];
    }
  }
#pragma unroll
  for (int u = 0; u < (NTP + 3) / 4; ++u) {
    const int T = T0 + wave + 4 * u;
    const bool more = u + 1 < (NTP + 3) / 4 && wave + 4 * (u + 1) < NTP;
    if (!SPLIT && more) {
      if (u & 1) w_load(std::integral_constant<int, 0>{}, wimg, T + 4);
      else w_load(std::integral_constant<int, 1>{}, wimg, T + 4);
    }
    if (wave + 4 * u >= NTP) continue;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const bool qk = T < 2 * SQ_K / 32;  // Q or K: D^T[feature, row]; V: D[row, feature], lane = feature
    if constexpr (SPLIT) {  // hi fragments in buffer 0, lo fragments in buffer 1; the next tile's follow as the buffers free up
      if (qk) {
#pragma unroll
        for (int s = 0; s < SQ_KS; ++s) acc = fd_mfma32(Wf[0][s], X[s], acc);
#pragma unroll
        for (int s = 0; s < SQ_KS; ++s) acc = fd_mfma32(Wf[0][s], Xl[s], acc);
      } else {
#pragma unroll
        for (int s = 0; s < SQ_KS; ++s) acc = fd_mfma32(X[s], Wf[0][s], acc);
#pragma unroll
        for (int s = 0; s < SQ_KS; ++s) acc = fd_mfma32(Xl[s], Wf[0][s], acc);
      }
      if (more) w_load(std::integral_constant<int, 0>{}, wimg, T + 4);
      if (qk) {
#pragma unroll
        for (int s = 0; s < SQ_KS; ++s) acc = fd_mfma32(Wf[1][s], X[s], acc);
      } else {
#pragma unroll
        for (int s = 0; s < SQ_KS; ++s) acc = fd_mfma32(X[s], Wf[1][s], acc);
      }
      if (more) w_load(std::integral_constant<int, 1>{}, wimg_lo, T + 4);
    }
    if (qk) {
      if constexpr (!SPLIT) {
#pragma unroll
        for (int s = 0; s < SQ_KS; ++s) acc = fd_mfma32(Wf[u & 1][s], X[s], acc);
      }
      if (m < M) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int f = 32 * T + 8 * g + 4 * hi;  // 4 consecutive output features
          const bool isk = f >= SQ_K;
          const int c = isk ? f - SQ_K : f, h = c / SA_HD, cc = c - h * SA_HD;
          const f32x4 bv = bq[u][g];
          const float sc = isk ? 1.f : qscale;
          sa_hx4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (fd_h)((acc[4 * g + q] + bv[q]) * sc);
          half_t* dst = (isk ? Ki : Qi) +
                        ((((((long)mb * H + h) * nt + (mr >> 5)) * SA_KS + (cc >> 4)) * 64 + ((cc >> 3) & 1) * 32 + (mr & 31)) << 3) + (cc & 7);
          *(sa_hx4*)dst = o;
        }
      }
    } else {
      if constexpr (!SPLIT) {
#pragma unroll
        for (int s = 0; s < SQ_KS; ++s) acc = fd_mfma32(X[s], Wf[u & 1][s], acc);
      }
      const int f = 32 * T + li, c = f - 2 * SQ_K, h = c / SA_HD, d = c - h * SA_HD;
      const float bv = bq[u][0][0];
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int m0 = row0 + 8 * g + 4 * hi;  // 4 consecutive rows = 4 consecutive keys of one sample (N % 4 == 0)
        if (m0 < M) {
          const int b0 = m0 / N, key = m0 - b0 * N;
          const int pp = (key & ~15) + sa_perm16(key & 15);
          sa_hx4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (fd_h)(acc[4 * g + q] + bv);
          half_t* dst = Vi + ((((((long)b0 * H + h) * SA_DT + (d >> 5)) * (2 * nt) + (pp >> 4)) * 64 + ((pp >> 3) & 1) * 32 + (d & 31)) << 3) + (pp & 7);
          *(sa_hx4*)dst = o;
        }
      }
    }
  }
  (void)dm;
}

__device__ __forceinline__ hx8 sa_ld(const half_t* p) { return __builtin_bit_cast(hx8, *(const u16x8*)p); }
__device__ __forceinline__ hx8 sa_pack8(const float* v) {
  hx8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (fd_h)v[e];
  return o;
}

// SA_NTW: key tiles per wave (N <= 128 SA_NTW); with 3 the kernel fits 256 registers and two blocks share a CU (LB = 2)
template <int SA_NTW, int LB>
__global__ __launch_bounds__(FD_THREADS, LB) void seq_attn_kernel(int B, int N, int Np, int H, const half_t* __restrict__ Qi,
                                                                 const half_t* __restrict__ Ki, const half_t* __restrict__ Vi,
                                                                 float* __restrict__ out, int out_ld, L2Warm warm) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = Np >> 5, ks = 2 * nt;
  unsigned warm_tok = 0;
  float* mxs = (float*)smem;              // [4][32]
  float* sms = mxs + 128;                 // [4][32]
  u16x8* Pfs = (u16x8*)(sms + 128);       // [2 nt][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  // XCD-aware: all query tiles of one (sample, head) on the same XCD (see attention3.hip)
  const int BH = B * H;
  int bhq, qt;
  {
    const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
    const int per = (BH + 7) >> 3;
    bhq = xcd * per + local / nt;
    qt = local % nt;
    if (local >= per * nt || bhq >= BH) return;
  }
  const int h = bhq % H, b = bhq / H;
  const long bh = bhq, rb = (long)b * N;
  const int i = 32 * qt + li;
  // ---- every global operand of this wave, requested up front (more than 4 key tiles per wave: the V fragments would not fit
  // beside the K fragments and the scores; they are requested once the scores are final)
  constexpr bool EARLY_V = SA_NTW <= 4;
  hx8 Va[8 * SA_NTW];
  auto v_request = [&]() {
    if (wave < SA_DT) {
      const half_t* vr = Vi + (((bh * SA_DT + wave) * ks) * 64 + lane) * 8;
#pragma unroll
      for (int s = 0; s < 8 * SA_NTW; ++s)
        if (s < ks) Va[s] = sa_ld(vr + s * 512);
    }
  };
  if constexpr (EARLY_V) v_request();
  hx8 Qf[SA_KS];
#pragma unroll
  for (int s = 0; s < SA_KS; ++s) Qf[s] = sa_ld(Qi + (((bh * nt + qt) * SA_KS + s) * 64 + lane) * 8);
  hx8 Kf[SA_NTW][SA_KS];
#pragma unroll
  for (int u = 0; u < SA_NTW; ++u) {
    const int t = wave + 4 * u;
    if (t < nt)
#pragma unroll
      for (int s = 0; s < SA_KS; ++s) Kf[u][s] = sa_ld(Ki + (((bh * nt + t) * SA_KS + s) * 64 + lane) * 8);
  }
  // ---- scores of this wave's key tiles (mask included: channel 80)
  f32x16 S[SA_NTW];
  float mx = -3.0e38f;
#pragma unroll
  for (int u = 0; u < SA_NTW; ++u) {
    if (wave + 4 * u < nt) {
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < SA_KS; ++s) acc = fd_mfma32(Kf[u][s], Qf[s], acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, acc[r]);
      S[u] = acc;
    }
  }
  // wave 3 (no V operands, nothing in flight from here on) touches the weights of the kernel launched next
  if (wave == 3) warm_tok = fd_l2_warm(warm, blockIdx.x, gridDim.x, lane, 64);
  // ---- softmax over all keys: own registers -> lane^32 -> the other waves through LDS
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (hi == 0) mxs[wave * 32 + li] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(mxs[li], mxs[32 + li]), fmaxf(mxs[64 + li], mxs[96 + li]));
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < SA_NTW; ++u)
    if (wave + 4 * u < nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = __builtin_amdgcn_exp2f((S[u][r] - mx) * 1.4426950408889634f);  // exp(x): one multiply + v_exp_f32 (expf adds range fix-ups)
        S[u][r] = e;
        sum += e;
      }
  sum += __shfl_xor(sum, 32, 64);
  if (hi == 0) sms[wave * 32 + li] = sum;
  __syncthreads();
  const float inv = 1.0f / (sms[li] + sms[32 + li] + sms[64 + li] + sms[96 + li]);
#pragma unroll
  for (int u = 0; u < SA_NTW; ++u) {
    const int t = wave + 4 * u;
    if (t < nt) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = S[u][r] * inv;
      Pfs[(2 * t) * 64 + lane] = __builtin_bit_cast(u16x8, sa_pack8(v));
      Pfs[(2 * t + 1) * 64 + lane] = __builtin_bit_cast(u16x8, sa_pack8(v + 8));
    }
  }
  if constexpr (!EARLY_V) v_request();
  __syncthreads();
  // ---- O^T[d, query] for channel tile `wave` over all keys
  if (wave < SA_DT) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8 * SA_NTW; ++s)
      if (s < ks) acc = fd_mfma32(Va[s], __builtin_bit_cast(hx8, Pfs[s * 64 + lane]), acc);
    if (i < N) {
      float* orow = out + (rb + i) * out_ld + (long)h * SA_HD + 32 * wave + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (32 * wave + 8 * g + 4 * hi < SA_HD) {
          f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
          *(f32x4*)(orow + 8 * g) = o;
        }
    }
  }
  fd_l2_warm_done(warm_tok);
}

size_t fd_seq_attention_image_bytes(int B, int N, int H) {
  const size_t Np = ((size_t)N + 31) / 32 * 32;
  return (size_t)B * H * Np * (2 * SA_KS * 16 + SA_DT * 32) * 2;  // Qi + Ki + Vi
}
int fd_seq_attention_supported(int N, int H, int hd) { return hd == SA_HD && N >= 1 && N <= 4 * SA_NTW_MAX * 32 && H >= 1; }

int fd_seq_attention(int B, int N, int H, const float* qkv, int ld, float scale, const float* res_mask, void* images,
                     float* out, int out_ld, hipStream_t st) {
  if (!fd_seq_attention_supported(N, H, SA_HD) || (ld & 3) || (out_ld & 3)) return FDIPT_EINVAL;
  const int Np = (N + 31) / 32 * 32, nt = Np / 32;
  half_t* Qi = (half_t*)images;
  half_t* Ki = Qi + (size_t)B * H * Np * SA_KS * 16;
  half_t* Vi = Ki + (size_t)B * H * Np * SA_KS * 16;
  const long units = 2L * B * H * nt * SA_KS * 64 + (long)B * H * SA_DT * 2 * nt * 64;
  hipLaunchKernelGGL(seq_images_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, B, N, Np, H, qkv, ld, scale,
                     res_mask, Qi, Ki, Vi);
  FD_CHECK_LAUNCH();
  return fd_seq_attention_run(B, N, H, images, out, out_ld, nullptr, st);
}

// Fused path: fd_seq_images_init once per forward, then per layer fd_seq_qkv (in_proj + images) and fd_seq_attention_run.
int fd_seq_images_init(int B, int N, int H, const float* res_mask, void* images, const SeqInitExtra& x, hipStream_t st) {
  if (x.Kb && (((N + 31) / 32 * 32) == N || (x.C & 31))) return FDIPT_EINVAL;
  if (!fd_seq_attention_supported(N, H, SA_HD)) return FDIPT_EINVAL;
  const int Np = (N + 31) / 32 * 32, nt = Np / 32;
  half_t* Qi = (half_t*)images;
  half_t* Ki = Qi + (size_t)B * H * Np * SA_KS * 16;
  half_t* Vi = Ki + (size_t)B * H * Np * SA_KS * 16;
  const long units = 2L * B * H * nt * SA_KS * 64 + (long)B * H * SA_DT * 2 * nt * 64;
  hipLaunchKernelGGL(seq_images_init_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, B, N, Np, H, res_mask, Qi, Ki, Vi, x);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
int fd_seq_qkv_supported(int N, int H, int d_model) { return d_model == SQ_K && H * SA_HD == SQ_K && (N & 3) == 0; }
int fd_seq_qkv(int B, int N, int H, const float* x, int ld_x, const void* wimg, const void* wimg_lo, const float* bias, float scale,
               void* images, hipStream_t st) {
  if (!fd_seq_qkv_supported(N, H, SQ_K) || (ld_x & 3)) return FDIPT_EINVAL;
  const int Np = (N + 31) / 32 * 32;
  half_t* Qi = (half_t*)images;
  half_t* Ki = Qi + (size_t)B * H * Np * SA_KS * 16;
  half_t* Vi = Ki + (size_t)B * H * Np * SA_KS * 16;
  const dim3 grid(24 * cdiv(cdiv(B * N, 32), 8));
  if (wimg_lo)  // split operands (lo image given)
    hipLaunchKernelGGL(seq_qkv_kernel<true>, grid, dim3(FD_THREADS), 0, st, B, N, Np, H, x, ld_x, (const char*)wimg, (const char*)wimg_lo,
                       bias, scale, Qi, Ki, Vi);
  else
    hipLaunchKernelGGL(seq_qkv_kernel<false>, grid, dim3(FD_THREADS), 0, st, B, N, Np, H, x, ld_x, (const char*)wimg, (const char*)nullptr,
                       bias, scale, Qi, Ki, Vi);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
int fd_seq_attention_run(int B, int N, int H, const void* images, float* out, int out_ld, const L2Warm* warm, hipStream_t st) {
  const L2Warm wm = warm ? *warm : L2Warm{};
  if (!fd_seq_attention_supported(N, H, SA_HD) || (out_ld & 3)) return FDIPT_EINVAL;
  const int Np = (N + 31) / 32 * 32, nt = Np / 32;
  const half_t* Qi = (const half_t*)images;
  const half_t* Ki = Qi + (size_t)B * H * Np * SA_KS * 16;
  const half_t* Vi = Ki + (size_t)B * H * Np * SA_KS * 16;
  const int per = (B * H + 7) / 8;
  const size_t smem = 2 * 128 * 4 + (size_t)2 * nt * 64 * 16;
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {  // N > 1000: more than the default 64 KB of dynamic LDS
    if (hipFuncSetAttribute((const void*)seq_attn_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  const dim3 grid(8 * per * nt), block(FD_THREADS);
  if (N <= 3 * 4 * 32) hipLaunchKernelGGL((seq_attn_kernel<3, 2>), grid, block, smem, st, B, N, Np, H, Qi, Ki, Vi, out, out_ld, wm);
  else if (N <= 4 * 4 * 32) hipLaunchKernelGGL((seq_attn_kernel<4, 1>), grid, block, smem, st, B, N, Np, H, Qi, Ki, Vi, out, out_ld, wm);
  else if (N <= 6 * 4 * 32) hipLaunchKernelGGL((seq_attn_kernel<6, 1>), grid, block, smem, st, B, N, Np, H, Qi, Ki, Vi, out, out_ld, wm);
  else hipLaunchKernelGGL((seq_attn_kernel<8, 1>), grid, block, smem, st, B, N, Np, H, Qi, Ki, Vi, out, out_ld, wm);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// ------------------------------------------------------------------ fp32 mode (round 5): the same block structure on v_mfma_f32_32x32x2_f32
// The fp32 mode ran the sequence attention on the generic LDS-score kernel of attention.hip (136 us per call at N = 300, B = 8; 680 us at
// N = 1000, B = 4: 6 % of a step).  Here: one block = 32 queries of one (sample, head), the key tiles of 32 dealt round-robin to the four
// waves, scores in registers, operands straight from the fp32 in_proj rows [B N, 3 d] (no images: an fp32 MFMA takes ONE k value per lane,
// and which channel a (lane half, step) pair stands for is free as long as A and B agree — lane half hi walks channels 40 hi .. 40 hi + 39,
// so a lane reads 40 consecutive floats of its row: ten 16 B loads per operand tile).  Softmax as in seq_attn_kernel; P crosses LDS once
// (fp32, [key][32 queries]); waves 0..2 then own one 32-channel tile of O^T[d, query] = V^T P^T over all keys (A = two 128 B rows of V per
// MFMA, coalesced; B = two LDS rows of P).  Masked / padded keys get a score of -1e30 (weight exactly 0, as key_padding_mask does).
template <int NTW, int LB>
__global__ __launch_bounds__(FD_THREADS, LB) void seq_attn_f32_kernel(int B, int N, int H, const float* __restrict__ qkv, int ld, float scale,
                                                                     const float* __restrict__ res_mask, float* __restrict__ out, int out_ld) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int nt = (N + 31) >> 5;
  float* mxs = (float*)smem;   // [4][32]
  float* sms = mxs + 128;      // [4][32]
  float* Ps = sms + 128;       // [32 nt][32]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  const int BH = B * H;
  int bhq, qt;
  {  // all query tiles of one (sample, head) on the same XCD (as seq_attn_kernel)
    const int id = blockIdx.x, xcd = id & 7, local = id >> 3, per = (BH + 7) >> 3;
    bhq = xcd * per + local / nt;
    qt = local % nt;
    if (local >= per * nt || bhq >= BH) return;
  }
  const int h = bhq % H, b = bhq / H, dm = H * SA_HD;
  const long rb = (long)b * N;
  const int qi = 32 * qt + li, qrow = qi < N ? qi : N - 1;
  constexpr int HC = SA_HD / 2;  // channels per lane half
  float Qf[HC];
  {
    const float* qp = qkv + (rb + qrow) * ld + h * SA_HD + HC * hi;
#pragma unroll
    for (int s = 0; s < HC; s += 4) {
      const f32x4 v = *(const f32x4*)(qp + s);
      Qf[s] = v[0] * scale; Qf[s + 1] = v[1] * scale; Qf[s + 2] = v[2] * scale; Qf[s + 3] = v[3] * scale;
    }
  }
  f32x16 S[NTW];
  float mx = -3.0e38f;
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int t = wave + 4 * u;
    if (t < nt) {
      const int key = 32 * t + li, krow = key < N ? key : N - 1;
      float Kf[HC];
      const float* kp = qkv + (rb + krow) * ld + dm + h * SA_HD + HC * hi;
#pragma unroll
      for (int s = 0; s < HC; s += 4) {
        const f32x4 v = *(const f32x4*)(kp + s);
        Kf[s] = v[0]; Kf[s + 1] = v[1]; Kf[s + 2] = v[2]; Kf[s + 3] = v[3];
      }
      const unsigned long long live = __ballot(key < N && (res_mask ? res_mask[rb + krow] != 0.f : true));
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < HC; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kf[s], Qf[s], acc, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        if (!((live >> c_row(r, lane)) & 1ull)) acc[r] = -1.0e30f;
        mx = fmaxf(mx, acc[r]);
      }
      S[u] = acc;
    }
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (hi == 0) mxs[wave * 32 + li] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(mxs[li], mxs[32 + li]), fmaxf(mxs[64 + li], mxs[96 + li]));
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < NTW; ++u)
    if (wave + 4 * u < nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = expf(S[u][r] - mx);
        S[u][r] = e;
        sum += e;
      }
  sum += __shfl_xor(sum, 32, 64);
  if (hi == 0) sms[wave * 32 + li] = sum;
  __syncthreads();
  const float inv = 1.0f / (sms[li] + sms[32 + li] + sms[64 + li] + sms[96 + li]);
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int t = wave + 4 * u;
    if (t < nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) Ps[(32 * t + c_row(r, lane)) * 32 + li] = S[u][r] * inv;
  }
  __syncthreads();
  // ---- O^T[d, query] for channel tile `wave` over all keys: MFMA m takes keys 2 m + hi
  if (wave < SA_DT) {
    const int d = 32 * wave + li;
    const bool dlive = d < SA_HD;
    const float* vp = qkv + rb * ld + 2 * dm + h * SA_HD + (dlive ? d : SA_HD - 1);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int nm = 16 * nt;  // key pairs
    constexpr int DEPTH = 16;
    float Vr[DEPTH];
#pragma unroll
    for (int m = 0; m < DEPTH; ++m) {
      const int key = 2 * m + hi;
      Vr[m] = vp[(long)(key < N ? key : N - 1) * ld];
    }
    for (int m0 = 0; m0 < nm; m0 += DEPTH) {
      float Vc[DEPTH];
#pragma unroll
      for (int m = 0; m < DEPTH; ++m) Vc[m] = Vr[m];
      if (m0 + DEPTH < nm) {
#pragma unroll
        for (int m = 0; m < DEPTH; ++m) {
          const int key = 2 * (m0 + DEPTH + m) + hi;
          Vr[m] = vp[(long)(key < N ? key : N - 1) * ld];
        }
      }
#pragma unroll
      for (int m = 0; m < DEPTH; ++m) {
        const float p = Ps[(2 * (m0 + m) + hi) * 32 + li];  // (padded keys: weight 0)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dlive ? Vc[m] : 0.f, p, acc, 0, 0, 0);
      }
    }
    if (qi < N) {
      float* orow = out + (rb + qi) * out_ld + (long)h * SA_HD + 32 * wave + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; ++g)
        if (32 * wave + 8 * g + 4 * hi < SA_HD) {
          f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
          *(f32x4*)(orow + 8 * g) = o;
        }
    }
  }
}

int fd_seq_attention_f32_supported(int N, int H, int hd, int ld) { return hd == SA_HD && N >= 1 && N <= 4 * SA_NTW_MAX * 32 && H >= 1 && !(ld & 3); }
// softmax(Q K^T * scale + key mask) V on the fp32 in_proj rows qkv [B N, ld] = (q | k | v), d_model = H * 80; out [B N, out_ld]
int fd_seq_attention_f32(int B, int N, int H, const float* qkv, int ld, float scale, const float* res_mask, float* out, int out_ld,
                         hipStream_t st) {
  if (!fd_seq_attention_f32_supported(N, H, SA_HD, ld) || (out_ld & 3)) return FDIPT_EINVAL;
  const int nt = (N + 31) / 32, per = (B * H + 7) / 8;
  const size_t smem = 2 * 128 * 4 + (size_t)32 * nt * 32 * 4;
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)seq_attn_f32_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 128 * 4 + 32 * 32 * 32 * 4) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  const dim3 grid(8 * per * nt), block(FD_THREADS);
  if (N <= 3 * 4 * 32) hipLaunchKernelGGL((seq_attn_f32_kernel<3, 2>), grid, block, smem, st, B, N, H, qkv, ld, scale, res_mask, out, out_ld);
  else hipLaunchKernelGGL((seq_attn_f32_kernel<8, 1>), grid, block, smem, st, B, N, H, qkv, ld, scale, res_mask, out, out_ld);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
