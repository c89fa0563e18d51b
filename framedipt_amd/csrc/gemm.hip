// gemm.hip — generic Linear (+bias/ReLU/mask/residual), LayerNorm and the MFMA self-test.
// Replaces the per-residue torch.nn.Linear / LayerNorm calls of the score network
// (framedipt/model/ipa_pytorch.py:36-58,202-239,325,386-413,531-541; score_network.py:86-96).
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

// C[M,N] = epi(A[M,K] * W[N,K]^T).  Block tile BM x BN (64x64 or 128x128), 4 waves as 2x2, BK = 64, LDS double
// buffer, next k-tile prefetched into registers while the current one feeds the MFMAs (one barrier per k-tile).
// LDS row padding of the tile loop.  fp32 (round 6): 4 words - rows stay 16 B aligned and, with a stride of 68 = 4 (mod 32) words, the
// ds_read_b128 / ds_write_b128 of the loop below are conflict-free (the layout of pair_mlp.hip's fp32 EdgeTransition); rounds 1 - 5 padded
// by one word and moved every fp32 operand through LDS 4 B at a time (two ds_read_b32 per MFMA, four ds_write_b32 per 16 B loaded).
template <class P> struct GemmPad { static constexpr int value = P::PAD; };
template <> struct GemmPad<PrecF32> { static constexpr int value = 4; };
template <class P, class SrcT, int ROWS>
struct TilePrefetch {  // ROWS x 64 tile of SrcT, row-major source with leading dimension ld
  static constexpr int EPV = 16 / sizeof(SrcT);          // elements per 16-byte vector
  static constexpr int VPR = 64 / EPV;                   // vectors per row
  static constexpr int NV = ROWS * VPR / FD_THREADS;     // vectors per thread
  typedef typename std::conditional<sizeof(SrcT) == 4, f32x4, u16x8>::type V;
  V r[NV];
  __device__ __forceinline__ void load(const SrcT* __restrict__ src, long ld, int row0, int nrows, int k0, int K, int tid) {
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int v = tid + u * FD_THREADS, rr = v / VPR, kk = (v % VPR) * EPV;
      V x;
#pragma unroll
      for (int e = 0; e < EPV; ++e) x[e] = 0;
      if (row0 + rr < nrows && k0 + kk < K) x = *(const V*)(src + (long)(row0 + rr) * ld + k0 + kk);
      r[u] = x;
    }
  }
  __device__ __forceinline__ void store(typename P::T* dst, int tid) const {
    constexpr int LDT = 64 * P::LDMUL + GemmPad<P>::value;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
      const int v = tid + u * FD_THREADS, rr = v / VPR, kk = (v % VPR) * EPV;
      typename P::T* d = dst + rr * LDT + kk;
      if constexpr (sizeof(SrcT) == 4 && sizeof(typename P::T) == 4) {
        *(f32x4*)d = r[u];
      } else if constexpr (sizeof(SrcT) == 4) {
        u16x4 h = {f2h(r[u][0]), f2h(r[u][1]), f2h(r[u][2]), f2h(r[u][3])};
        *(u16x4*)d = h;
        if constexpr (P::LDMUL == 2) {  // split operands: the lo parts follow the row's 64 hi parts
          u16x4 l;
#pragma unroll
          for (int q = 0; q < 4; ++q) l[q] = f2h(r[u][q] - h2f(h[q]));
          *(u16x4*)(d + 64) = l;
        }
      } else {
        static_assert(P::LDMUL == 1, "split operands are built from fp32 sources");
        *(u16x8*)d = r[u];
      }
    }
  }
};

// main loop shared by every GEMM kernel: acc[TM][TN] (32x32 tiles of this wave) = A[m0.., :K] * W[n0.., :K]^T
// SWAP: the MFMA operands exchanged -> the tiles come out transposed (lane = row m, registers = 4-runs of columns n)
template <class P, class AT, class WT, int BM, int BN, bool SWAP = false>
__device__ __forceinline__ void gemm_tile(f32x16 (&acc)[BM / 64][BN / 64], int M, int N, int K,
                                          const AT* __restrict__ A, int lda, const WT* __restrict__ W, int ldw,
                                          typename P::T* smem, int m0, int n0, int tid) {
  constexpr int BKL = 64, LDT = BKL * P::LDMUL + GemmPad<P>::value;
  constexpr int TM = BM / 64, TN = BN / 64;
  constexpr int STAGE = (BM + BN) * LDT;  // elements per pipeline stage: A tile then W tile
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1, hi = lane >> 5;
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int jn = 0; jn < TN; ++jn)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][jn][r] = 0.f;
  TilePrefetch<P, AT, BM> pa;
  TilePrefetch<P, WT, BN> pw;
  pa.load(A, lda, m0, M, 0, K, tid);
  pw.load(W, ldw, n0, N, 0, K, tid);
  pa.store(smem, tid);
  pw.store(smem + BM * LDT, tid);
  __syncthreads();
  const int nk = (K + BKL - 1) / BKL;
  for (int kt = 0; kt < nk; ++kt) {
    const typename P::T* Ac = smem + (kt & 1) * STAGE;
    const typename P::T* Wc = Ac + BM * LDT;
    typename P::T* An = smem + ((kt + 1) & 1) * STAGE;
    if (kt + 1 < nk) {
      pa.load(A, lda, m0, M, (kt + 1) * BKL, K, tid);
      pw.load(W, ldw, n0, N, (kt + 1) * BKL, K, tid);
    }
    if constexpr (std::is_same<P, PrecF32>::value) {
      // fp32: operands as 16 B runs.  Within 8 consecutive k the lane half `hi` reads k0 + 4 hi .. + 3 and MFMA j multiplies the pairs
      // (k0 + j | k0 + 4 + j) - the same pairing on both operands, so nothing is permuted (only the order of the k sum differs from a
      // (k, k + 1) walk).  The operand runs of a k-group are read once per row / column tile and shared by the tiles that use them.
#pragma unroll
      for (int k = 0; k < BKL; k += 8) {
        f32x4 av[TM], wv[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) av[i] = *(const f32x4*)(Ac + ((wr * TM + i) * 32 + (lane & 31)) * LDT + k + 4 * hi);
#pragma unroll
        for (int jn = 0; jn < TN; ++jn) wv[jn] = *(const f32x4*)(Wc + ((wc * TN + jn) * 32 + (lane & 31)) * LDT + k + 4 * hi);
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int jn = 0; jn < TN; ++jn)
              acc[i][jn] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x2f32(wv[jn][j], av[i][j], acc[i][jn], 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_32x32x2f32(av[i][j], wv[jn][j], acc[i][jn], 0, 0, 0);
      }
    } else
#pragma unroll
    for (int k = 0; k < BKL; k += P::KS)
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int jn = 0; jn < TN; ++jn)
          if constexpr (SWAP)
            P::mma(acc[i][jn], Wc + ((wc * TN + jn) * 32 + (lane & 31)) * LDT + k,
                   Ac + ((wr * TM + i) * 32 + (lane & 31)) * LDT + k, hi);
          else
            P::mma(acc[i][jn], Ac + ((wr * TM + i) * 32 + (lane & 31)) * LDT + k,
                   Wc + ((wc * TN + jn) * 32 + (lane & 31)) * LDT + k, hi);
    if (kt + 1 < nk) {
      pa.store(An, tid);
      pw.store(An + BM * LDT, tid);
    }
    __syncthreads();
  }
}

template <class P, class AT, class WT, int BM, int BN>
__global__ __launch_bounds__(FD_THREADS) void linear_kernel(int M, int N, int K, const AT* __restrict__ A, int lda,
                                                            const WT* __restrict__ W, int ldw,
                                                            const float* __restrict__ bias,
                                                            const float* __restrict__ residual, int ldr,
                                                            const float* __restrict__ rowmask, int relu,
                                                            float* __restrict__ out, int ldo) {
  constexpr int LDT = 64 * P::LDMUL + GemmPad<P>::value;
  constexpr int TM = BM / 64, TN = BN / 64;  // 32x32 MFMA tiles per wave in each direction
  __shared__ __attribute__((aligned(16))) typename P::T smem[2 * (BM + BN) * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  f32x16 acc[TM][TN];
  gemm_tile<P, AT, WT, BM, BN>(acc, M, N, K, A, lda, W, ldw, smem, m0, n0, tid);
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int n = n0 + (wc * TN + jn) * 32 + (lane & 31);
    if (n >= N) continue;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = m0 + (wr * TM + i) * 32 + c_row(r, lane);
        if (m < M) {
          float v = acc[i][jn][r] + bv;
          if (relu) v = fmaxf(v, 0.f);
          if (rowmask) v *= rowmask[m];
          if (residual) v += residual[(long)m * ldr + n];
          out[(long)m * ldo + n] = v;
        }
      }
  }
}

// Split-K variant for the "long K, few columns" products (IPA output projection: M = B*N rows, N = c_s, K = 2688): the
// 64 x 64 tile grid alone is ~150 blocks of 42 dependent k-iterations each; blockIdx.z takes a K slice and writes its
// partial product to parts[z] (bias added by slice 0, row mask by every slice) - the LayerNorm that follows sums them.
template <class P, class AT, class WT, int BN = 64>
__global__ __launch_bounds__(FD_THREADS) void linear_splitk_kernel(int M, int N, int K, int kslice, const AT* __restrict__ A,
                                                                   int lda, const WT* __restrict__ W, int ldw,
                                                                   const float* __restrict__ bias,
                                                                   const float* __restrict__ rowmask,
                                                                   float* __restrict__ parts, long part_stride, int ldo) {
  constexpr int BM = 64, TN = BN / 64;
  extern __shared__ __attribute__((aligned(16))) char splitk_smem[];  // 2 * (BM + BN) * LDT elements (dynamic: 68 / 102 KB with split operands)
  typename P::T* smem = (typename P::T*)splitk_smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, z = blockIdx.z;
  const int k0 = z * kslice, klen = K - k0 < kslice ? K - k0 : kslice;
  f32x16 acc[1][TN];
  gemm_tile<P, AT, WT, BM, BN>(acc, M, N, klen, A + k0, lda, W + k0, ldw, smem, m0, n0, tid);
  float* out = parts + z * part_stride;
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int n = n0 + (wc * TN + jn) * 32 + (lane & 31);
    if (n >= N) continue;
    const float bv = (bias && z == 0) ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wr * 32 + c_row(r, lane);
      if (m < M) {
        float v = acc[0][jn][r] + bv;
        if (rowmask) v *= rowmask[m];
        out[(long)m * ldo + n] = v;
      }
    }
  }
}

// ------------------------------------------------------------------ IPA projection with attention-operand epilogue
// [q | kv | q_pts | kv_pts] = s W^T + b (ipa_pytorch.py:202-239) written directly as the operand images the
// register attention kernel (attention3.hip) consumes:
//   Qb [B,H,N,C]  bf16, pre-multiplied by sqrt(1/(3C))            (A/B fragments: 16-byte loads)
//   Kb [B,H,Np/32,16,64,8] bf16, MFMA FRAGMENT order: key 32t + (lane & 31), channel 16s + 8(lane >> 5) + e — one A fragment
//      of attention3 is ONE linear 1 KB load (a row-per-lane 16 B gather costs 8x the TA cycles, tools/micro/io_pattern.hip)
//   Qb the same with queries (B fragments);  rows >= N of Kb are zeroed (kv_zero_pad_kernel)
//   Vt [B,H,C/32,Np/16,64,8] bf16: V transposed in fragment order, row = channel 32dt + (lane & 31), key position
//      16s + 8(lane >> 5) + e with the keys permuted inside every 16-group (perm16: C/D fragment -> B fragment order)
__device__ __forceinline__ int g_perm16(int pos) { return 4 * (pos >> 3) + (pos & 3) + 8 * ((pos & 7) >> 2); }

__global__ __launch_bounds__(FD_THREADS) void ipa_proj_kernel(ProjArgs a) {
  typedef PrecHalf P;
  constexpr int BM = 128, BN = 128, LDT = 64 * P::LDMUL + P::PAD, TM = 2, TN = 2;
  __shared__ __attribute__((aligned(16))) P::T smem[2 * (BM + BN) * LDT];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;
  const int M = a.B * a.N, HC = a.H * a.C, NOUT = 3 * HC + a.PT;
  f32x16 acc[TM][TN];
  const int ntl_ = a.Np >> 5;
  // Q and K column blocks (uniform per block: H*C and 2C are multiples of 128) run with the MFMA operands exchanged: a lane
  // then owns a row and 4-runs of channels, i.e. 8 B pieces of the fragment-order images; the two lane halves and the 32 rows
  // of a tile make 512 contiguous bytes per store instruction (the untransposed epilogue wrote single bf16 values)
  const bool qk_block = (HC % BN) == 0 && (a.C % BN) == 0 && (n0 < HC || (n0 < 3 * HC && ((n0 - HC) % (2 * a.C)) < a.C));
  if (qk_block) {
    gemm_tile<P, float, half_t, BM, BN, true>(acc, M, NOUT, a.K, a.A, a.lda, (const half_t*)a.W, a.K, smem, m0, n0, tid);
    const bool isq = n0 < HC;
    const int nn0 = isq ? n0 : n0 - HC, hh = isq ? nn0 / a.C : nn0 / (2 * a.C), cb = isq ? nn0 % a.C : nn0 % (2 * a.C);
    half_t* dst0 = isq ? a.Qb : a.Kb;
    const float sc = isq ? a.qscale : 1.f;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      const int m = m0 + (wr * TM + i) * 32 + (lane & 31);
      if (m >= M) continue;
      const int b = m / a.N, r = m - b * a.N;
#pragma unroll
      for (int jn = 0; jn < TN; ++jn) {
        const int ct = cb + (wc * TN + jn) * 32;  // first channel of this 32-channel tile (within the head)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int cc = ct + 8 * g + 4 * (lane >> 5);
          const f32x4 bv = *(const f32x4*)(a.bias + n0 + (wc * TN + jn) * 32 + 8 * g + 4 * (lane >> 5));
          u16x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = f2h((acc[i][jn][4 * g + q] + bv[q]) * sc);
          *(u16x4*)(dst0 + ((((((long)b * a.H + hh) * ntl_ + (r >> 5)) * (a.C >> 4) + (cc >> 4)) * 64 + ((cc >> 3) & 1) * 32 + (r & 31)) << 3) +
                    (cc & 7)) = o;
        }
      }
    }
    return;
  }
  gemm_tile<P, float, half_t, BM, BN>(acc, M, NOUT, a.K, a.A, a.lda, (const half_t*)a.W, a.K, smem, m0, n0, tid);
#pragma unroll
  for (int jn = 0; jn < TN; ++jn) {
    const int n = n0 + (wc * TN + jn) * 32 + (lane & 31);
    if (n >= NOUT) continue;
    const float bv = a.bias[n];
    // column class (uniform per lane)
    int kind, hh = 0, cc = 0;
    if (n < HC) { kind = 0; hh = n / a.C; cc = n % a.C; }
    else if (n < 3 * HC) {
      const int nn = n - HC;
      hh = nn / (2 * a.C);
      cc = nn % (2 * a.C);
      kind = cc < a.C ? 1 : 2;
      if (kind == 2) cc -= a.C;
    } else { kind = 3; cc = n - 3 * HC; }
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int mg = m0 + (wr * TM + i) * 32 + 8 * g + 4 * (lane >> 5);  // 4 consecutive rows mg .. mg+3
        float v[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = acc[i][jn][4 * g + q] + bv;
        const int ntl = a.Np >> 5;
        if (kind == 2 && mg + 3 < M && (a.N & 3) == 0) {
          const int b = mg / a.N, key = mg - b * a.N;  // 4 keys of one sample (N % 4 == 0), contiguous after perm16
          const int pp = (key & ~15) + g_perm16(key & 15);
          u16x4 o = {f2h(v[0]), f2h(v[1]), f2h(v[2]), f2h(v[3])};
          *(u16x4*)(a.Vt + ((((((long)b * a.H + hh) * (a.C >> 5) + (cc >> 5)) * (2 * ntl) + (pp >> 4)) * 64 + ((pp >> 3) & 1) * 32 +
                             (cc & 31)) << 3) + (pp & 7)) = o;
          continue;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int m = mg + q;
          if (m >= M) continue;
          const int b = m / a.N, r = m - b * a.N;
          if (kind == 0 || kind == 1) {
            half_t* dst = kind == 0 ? a.Qb : a.Kb;
            dst[((((((long)b * a.H + hh) * ntl + (r >> 5)) * (a.C >> 4) + (cc >> 4)) * 64 + ((cc >> 3) & 1) * 32 + (r & 31)) << 3) +
                (cc & 7)] = f2h(kind == 0 ? v[q] * a.qscale : v[q]);
          } else if (kind == 2) {
            const int pp = (r & ~15) + g_perm16(r & 15);
            a.Vt[((((((long)b * a.H + hh) * (a.C >> 5) + (cc >> 5)) * (2 * ntl) + (pp >> 4)) * 64 + ((pp >> 3) & 1) * 32 + (cc & 31)) << 3) +
                 (pp & 7)] = f2h(v[q]);
          } else a.pts[(long)m * a.PT + cc] = v[q];
        }
      }
  }
}

// zero the padded keys [N, Np) of Kb and Vt (P is exactly 0 there and the logits are masked, but the operands must not
// be NaN/Inf).  One thread per (bh, padded key, 8-channel group).
__global__ void kv_zero_pad_kernel(long BH, int N, int Np, int C, half_t* __restrict__ Kb, half_t* __restrict__ Vt,
                                   half_t* __restrict__ Vt2, uint4* __restrict__ extra, long extra_n16) {
  // another once-per-forward zero fill riding on this launch (the value-point image of attention3: 16 B units)
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < extra_n16; i += (long)gridDim.x * blockDim.x)
    extra[i] = make_uint4(0, 0, 0, 0);
  const int pad = Np - N, ntl = Np >> 5, cg = C >> 3;
  const long n = BH * pad * cg;
  const u16x8 z8 = {0, 0, 0, 0, 0, 0, 0, 0};
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int g = (int)(i % cg);
    const long r2 = i / cg;
    const int key = N + (int)(r2 % pad);
    const long bh = r2 / pad;
    // Kb: channels 8g .. 8g+7 of this key are one 16 B unit
    *(u16x8*)(Kb + ((((bh * ntl + (key >> 5)) * (C >> 4) + (g >> 1)) * 64 + (g & 1) * 32 + (key & 31)) << 3)) = z8;
    // Vt: this key in channels 8g .. 8g+7
    const int pp = (key & ~15) + g_perm16(key & 15);
    for (int c = 8 * g; c < 8 * g + 8; ++c) {
      const long o = ((((bh * (C >> 5) + (c >> 5)) * (2 * ntl) + (pp >> 4)) * 64 + ((pp >> 3) & 1) * 32 + (c & 31)) << 3) + (pp & 7);
      Vt[o] = 0;
      if (Vt2) Vt2[o] = 0;
    }
  }
}

// zero the padded keys of Kb / Vt (once per forward), for callers that run the projection elsewhere (ipa_proj2.hip)
int fd_ipa_proj_zero_pads(const ProjArgs& a, void* extra, size_t extra_bytes, hipStream_t st) {
  if (extra_bytes & 15) return FDIPT_EINVAL;
  if (a.Np > a.N || extra_bytes)
    hipLaunchKernelGGL(kv_zero_pad_kernel, dim3(512), dim3(256), 0, st, (long)a.B * a.H, a.N, a.Np, a.C, a.Kb, a.Vt, a.Vt_lo, (uint4*)extra,
                       (long)(extra_bytes >> 4));
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
int fd_ipa_proj(const ProjArgs& a, hipStream_t st) {
  const int M = a.B * a.N, NOUT = 3 * a.H * a.C + a.PT;
  if ((a.K & 7) || (a.lda & 3)) return FDIPT_EINVAL;
  if ((a.Np & 31) || (a.C & 31)) return FDIPT_EINVAL;
  if (a.Np > a.N && a.zero_pads)  // the pads are never written by the epilogue: once per forward is enough
    hipLaunchKernelGGL(kv_zero_pad_kernel, dim3(256), dim3(256), 0, st, (long)a.B * a.H, a.N, a.Np, a.C, a.Kb, a.Vt, (half_t*)nullptr, (uint4*)nullptr, 0L);
  hipLaunchKernelGGL(ipa_proj_kernel, dim3(cdiv(M, 128), cdiv(NOUT, 128)), dim3(FD_THREADS), 0, st, a);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

template <class P, class AT, class WT>
static void launch_tiles(int M, int N, int K, const AT* A, int lda, const WT* W, int ldw, const float* bias,
                         const float* residual, int ldr, const float* rowmask, int relu, float* out, int ldo,
                         hipStream_t st) {
  // fp32: 64 x 64 tiles at every size (70 KB of LDS: two blocks per CU cover each other's four exposed k-tile round trips; the merged IPA
  // projection of the fp32 mode, 2400 x 6816 x 256: 121 against 143 us on 128 x 128 tiles with one block per CU)
  if (!std::is_same<P, PrecF32>::value && M >= 1024 && N >= 1024) {
    hipLaunchKernelGGL((linear_kernel<P, AT, WT, 128, 128>), dim3(cdiv(M, 128), cdiv(N, 128)), dim3(FD_THREADS), 0, st, M,
                       N, K, A, lda, W, ldw, bias, residual, ldr, rowmask, relu, out, ldo);
  } else {
    hipLaunchKernelGGL((linear_kernel<P, AT, WT, 64, 64>), dim3(cdiv(M, 64), cdiv(N, 64)), dim3(FD_THREADS), 0, st, M, N, K,
                       A, lda, W, ldw, bias, residual, ldr, rowmask, relu, out, ldo);
  }
}

static int launch_linear(int precision, int M, int N, int K, const float* A, int lda, const void* W, int ldw,
                         const float* bias, const float* residual, int ldr, const float* rowmask, int relu, float* out,
                         int ldo, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || !A || !W || !out || (K & 7) || (lda & 3) || (ldw & 7)) return FDIPT_EINVAL;
  if (precision == FDIPT_PREC_F32)
    launch_tiles<PrecF32, float, float>(M, N, K, A, lda, (const float*)W, ldw, bias, residual, ldr, rowmask, relu, out, ldo,
                                        st);
  else
    launch_tiles<PrecHalf, float, half_t>(M, N, K, A, lda, (const half_t*)W, ldw, bias, residual, ldr, rowmask, relu, out,
                                          ldo, st);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// same product with the pair representation as A (float in fp32 mode, bf16 in bf16 mode): IPA pair bias linear_b(z)
int fd_linear_z(int precision, long M, int N, int K, const void* A, const void* W, const float* bias, float* out,
                hipStream_t st) {
  if (M <= 0 || (K & 7)) return FDIPT_EINVAL;
  if (precision == FDIPT_PREC_F32)
    launch_tiles<PrecF32, float, float>((int)M, N, K, (const float*)A, K, (const float*)W, K, bias, nullptr, 0, nullptr, 0,
                                        out, N, st);
  else
    launch_tiles<PrecHalf, half_t, half_t>((int)M, N, K, (const half_t*)A, K, (const half_t*)W, K, bias, nullptr, 0, nullptr,
                                           0, out, N, st);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

int fd_linear(int precision, int M, int N, int K, const float* A, int lda, const void* W, int ldw, const float* bias,
              const float* residual, int ldr, const float* rowmask, int relu, float* out, int ldo, hipStream_t st) {
  return launch_linear(precision, M, N, K, A, lda, W, ldw, bias, residual, ldr, rowmask, relu, out, ldo, st);
}

// LayerNorm over the last dim (eps 1e-5, biased variance = torch.nn.LayerNorm), one wave per row, D <= 1024.
__global__ __launch_bounds__(FD_THREADS) void layernorm_kernel(int M, int D, const float* __restrict__ x, int ldx,
                                                               const float* __restrict__ residual, int ldr, int nparts,
                                                               long part_stride,
                                                               const float* __restrict__ gamma,
                                                               const float* __restrict__ beta,
                                                               const float* __restrict__ rowmask,
                                                               float* __restrict__ out, int ldo,
                                                               const float* __restrict__ extra, int ld_extra, int n_extra,
                                                               L2Warm warm) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (FD_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= M) return;
  // optional: columns D .. D + n_extra of the output row are a copy of extra[row] (the skip embedding of the sequence transformer's
  // input, computed for all blocks in one launch at the start of the forward)
  for (int c = lane; c < n_extra; c += 64) out[(long)row * ldo + D + c] = extra[(long)row * ld_extra + c];
  float v[16];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    float t = 0.f;
    if (c < D) {
      t = x[(long)row * ldx + c];
      if (residual) {  // split-K partial products: all requested together (a rolled loop is nparts dependent round trips)
        float pr[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) pr[k] = k < nparts ? residual[k * part_stride + (long)row * ldr + c] : 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) t += pr[k];
      }
    }
    v[i] = t;
    s += t;
  }
  // the block's first wave touches the weights of the kernel launched next (common.hpp: L2 warm-up hand-over)
  const unsigned warm_tok = threadIdx.x < 64 ? fd_l2_warm(warm, blockIdx.x, gridDim.x, lane, 64) : 0u;
  const float mu = wave_sum(s) / (float)D;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    if (c < D) {
      const float d = v[i] - mu;
      q += d * d;
    }
  }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + 1e-5f);
  const float rm = rowmask ? rowmask[row] : 1.f;
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int c = lane + i * 64;
    if (c < D) out[(long)row * ldo + c] = ((v[i] - mu) * rstd * gamma[c] + beta[c]) * rm;
  }
  fd_l2_warm_done(warm_tok);
}

// D == 256, split-K partial products: one row per wave, a lane owns 4 consecutive features -> every operand is ONE 16 B load per
// lane (the generic kernel issues four 4 B loads per operand), all of them in flight together; same summation order.
__global__ __launch_bounds__(FD_THREADS) void layernorm256_parts_kernel(int M, const float* __restrict__ x, int ldx,
                                                                        const float* __restrict__ parts, int ldr, int nparts,
                                                                        long part_stride, const float* __restrict__ gamma,
                                                                        const float* __restrict__ beta,
                                                                        const float* __restrict__ rowmask, float* __restrict__ out,
                                                                        int ldo, const float* __restrict__ extra, int ld_extra,
                                                                        int n_extra, L2Warm warm) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (FD_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= M) return;
  const f32x4 xv = *(const f32x4*)(x + (long)row * ldx + 4 * lane);
  f32x4 pr[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    pr[k] = k < nparts ? *(const f32x4*)(parts + k * part_stride + (long)row * ldr + 4 * lane) : f32x4{0.f, 0.f, 0.f, 0.f};
  const f32x4 gm = *(const f32x4*)(gamma + 4 * lane), bt = *(const f32x4*)(beta + 4 * lane);
  f32x4 ex = {0.f, 0.f, 0.f, 0.f};
  if (4 * lane < n_extra) ex = *(const f32x4*)(extra + (long)row * ld_extra + 4 * lane);
  const float rm = rowmask ? rowmask[row] : 1.f;
  const unsigned warm_tok = threadIdx.x < 64 ? fd_l2_warm(warm, blockIdx.x, gridDim.x, lane, 64) : 0u;
  if (4 * lane < n_extra) *(f32x4*)(out + (long)row * ldo + 256 + 4 * lane) = ex;
  f32x4 v = xv;
#pragma unroll
  for (int k = 0; k < 8; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] += pr[k][q];
  const float mu = wave_sum((v[0] + v[1]) + (v[2] + v[3])) * (1.0f / 256.0f);
  float qs = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) { const float d = v[q] - mu; qs += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(qs) * (1.0f / 256.0f) + 1e-5f);
  f32x4 o;
#pragma unroll
  for (int q = 0; q < 4; ++q) o[q] = ((v[q] - mu) * rstd * gm[q] + bt[q]) * rm;
  *(f32x4*)(out + (long)row * ldo + 4 * lane) = o;
  fd_l2_warm_done(warm_tok);
}

int fd_layernorm(int M, int D, const float* x, int ldx, const float* residual, int ldr, const float* gamma,
                 const float* beta, const float* rowmask, float* out, int ldo, hipStream_t st) {
  if (M <= 0 || D <= 0 || D > 1024 || !x || !gamma || !beta || !out) return FDIPT_EINVAL;
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(M, FD_THREADS / 64)), dim3(FD_THREADS), 0, st, M, D, x, ldx, residual, ldr, 1,
                     0L, gamma, beta, rowmask, out, ldo, (const float*)nullptr, 0, 0, L2Warm{});
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
// LayerNorm(x + sum_k parts[k]) for the split-K products of fd_linear_splitk
int fd_layernorm_parts(int M, int D, const float* x, int ldx, const float* parts, int ldr, int nparts, long part_stride,
                       const float* gamma, const float* beta, const float* rowmask, float* out, int ldo, const float* extra,
                       int ld_extra, int n_extra, const L2Warm* warm, hipStream_t st) {
  if (M <= 0 || D <= 0 || D > 1024 || !x || !parts || nparts < 1 || nparts > 8 || !gamma || !beta || !out) return FDIPT_EINVAL;
  if (D == 256 && !((ldx | ldr | ldo | ld_extra) & 3) && !(part_stride & 3) && (!extra || (n_extra & 3) == 0) && n_extra <= 256 &&
      !FD_DEV_ENV("FDIPT_LN_GENERIC")) {
    hipLaunchKernelGGL(layernorm256_parts_kernel, dim3(cdiv(M, FD_THREADS / 64)), dim3(FD_THREADS), 0, st, M, x, ldx, parts, ldr,
                       nparts, part_stride, gamma, beta, rowmask, out, ldo, extra, ld_extra, extra ? n_extra : 0,
                       warm ? *warm : L2Warm{});
    FD_CHECK_LAUNCH();
    return FDIPT_OK;
  }
  hipLaunchKernelGGL(layernorm_kernel, dim3(cdiv(M, FD_THREADS / 64)), dim3(FD_THREADS), 0, st, M, D, x, ldx, parts, ldr, nparts,
                     part_stride, gamma, beta, rowmask, out, ldo, extra, ld_extra, extra ? n_extra : 0, warm ? *warm : L2Warm{});
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
// parts[z][M, ldo] = (A[:, z-th K slice] W[:, slice]^T (+ bias for z = 0)) * rowmask; bf16 operands, fp32 activations in
int fd_linear_splitk(int M, int N, int K, int nsplit, const float* A, int lda, const void* W, int ldw, const float* bias,
                     const float* rowmask, float* parts, long part_stride, int ldo, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || nsplit < 1 || (K & 7) || (lda & 3) || (ldw & 7)) return FDIPT_EINVAL;
  const int kslice = ((K + nsplit - 1) / nsplit + 63) / 64 * 64;
  if ((long)kslice * (nsplit - 1) >= K) return FDIPT_EINVAL;  // an empty slice
  hipLaunchKernelGGL((linear_splitk_kernel<PrecHalf, float, half_t>), dim3(cdiv(M, 64), cdiv(N, 64), nsplit), dim3(FD_THREADS),
                     (size_t)2 * 128 * (64 + PrecHalf::PAD) * sizeof(half_t), st, M, N, K, kslice, A, lda, (const half_t*)W, ldw, bias, rowmask, parts, part_stride, ldo);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
// the same on split operands (fp32 activations and the fp32 weight matrix, both split into hi + lo parts while they are staged)
int fd_linear_splitk_split(int M, int N, int K, int nsplit, const float* A, int lda, const float* W, int ldw, const float* bias,
                           const float* rowmask, float* parts, long part_stride, int ldo, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || nsplit < 1 || (K & 7) || (lda & 3) || (ldw & 3)) return FDIPT_EINVAL;
  const int kslice = ((K + nsplit - 1) / nsplit + 63) / 64 * 64;
  if ((long)kslice * (nsplit - 1) >= K) return FDIPT_EINVAL;  // an empty slice
  // (128-column blocks - the fp32 A tile read and split by two column blocks instead of four - measured 38 us against 30: 228 blocks)
#ifndef FD_SPLITK_BN
#define FD_SPLITK_BN 64
#endif
  constexpr int BN = (FD_SPLITK_BN);
  constexpr size_t smem = (size_t)2 * (64 + BN) * (64 * PrecSplit::LDMUL + PrecSplit::PAD) * sizeof(half_t);
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)linear_splitk_kernel<PrecSplit, float, float, BN>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  hipLaunchKernelGGL((linear_splitk_kernel<PrecSplit, float, float, BN>), dim3(cdiv(M, 64), cdiv(N, BN), nsplit), dim3(FD_THREADS), smem,
                     st, M, N, K, kslice, A, lda, W, ldw, bias, rowmask, parts, part_stride, ldo);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
// ------------------------------------------------------------------ IPA output projection on split operands: the dedicated kernel
// linear_out (ipa_pytorch.py:324-329) at the reference widths: M = B N rows, K = 2688 features, 256 columns, every product on split
// operands (x = hi + lo, W = hi + lo, x_hi W_lo + x_lo W_hi + x_hi W_hi).  The generic split-K tile loop above re-splits the fp32 weight
// matrix in each of its 38 row blocks and runs 14 dependent load -> split -> LDS -> barrier -> MFMA rounds per block (29 us, MFMA 11 %).
// Here: a block = 64 rows x ALL 256 columns x one K slice of KSL k-steps (2688 = 6 x 28 x 16: 38 x 6 = 228 blocks, one round of the 256
// CUs); the block's activation slice is read once (coalesced fp32 rows), split, and parked in LDS as hi rows | lo rows (row stride
// 2 KW + 16 B: the 16 lanes of a b128 read hit 16 distinct slots); the weights are fragment images prepared once (fd_chain_build_image /
// _lo of the [256, K] matrix) and go STRAIGHT from L2 into registers — wave w owns column tile w, one linear 1 KB load per fragment,
// OP_DEPTH k-steps ahead, no LDS, no barrier after the staging one.  Per k-step and wave: 2 KB of weights, 4 LDS reads, 6 MFMAs on two
// independent accumulators (the two 32-row tiles).  The slice count is a constant of the kernel (the order of the partial sums is part
// of a row's result: any batch composition gives the same bits); the LayerNorm that follows sums the slices.
#define OP_DEPTH 6
template <int KSL>
__global__ __launch_bounds__(512, 1) void outproj_split_kernel(int M, const float* __restrict__ A, int lda, const char* __restrict__ w_hi,
                                                                const char* __restrict__ w_lo, int ks_total,
                                                                const float* __restrict__ bias, const float* __restrict__ rowmask,
                                                                float* __restrict__ parts, long part_stride, int ldo) {
  constexpr int KW = KSL * 16, XROW = KW * 2 + 16, XLO = 64 * XROW, C4 = KW / 4, NV = 64 * C4 / 512;
  static_assert((64 * C4) % 512 == 0, "whole float4 columns per thread");
  extern __shared__ __attribute__((aligned(16))) char op_smem[];
  float* rm = (float*)(op_smem + 2 * XLO);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
  const int m0 = blockIdx.x * 64, z = blockIdx.y;
  typedef fd_h op_hx4 __attribute__((ext_vector_type(4)));
  FD_STAMP(0);
  // weight fragments of the first k-steps: in flight before anything else
  const size_t woff = ((size_t)(wave * ks_total + z * KSL) * 64 + lane) * 16;
  const char* wh = w_hi + woff;
  const char* wl = w_lo + woff;
  hx8 Wh[OP_DEPTH], Wl[OP_DEPTH];
#pragma unroll
  for (int s = 0; s < OP_DEPTH - 1; ++s) {
    Wh[s] = __builtin_bit_cast(hx8, *(const u16x8*)(wh + (size_t)s * 1024));
    Wl[s] = __builtin_bit_cast(hx8, *(const u16x8*)(wl + (size_t)s * 1024));
  }
  FD_STAMP(1);
  {  // activation slice -> LDS (all loads of a thread in flight together)
    f32x4 xv[NV];
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int idx = tid + k * 512, r = idx / C4, c4 = idx - r * C4;
      const int gr = m0 + r < M ? m0 + r : M - 1;
      xv[k] = *(const f32x4*)(A + (long)gr * lda + z * KW + 4 * c4);
    }
    FD_STAMP(2);
    if (tid < 64) rm[tid] = rowmask ? rowmask[m0 + tid < M ? m0 + tid : M - 1] : 1.f;
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int idx = tid + k * 512, r = idx / C4, c4 = idx - r * C4;
      op_hx4 pk, pl;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        pk[q] = (fd_h)xv[k][q];
        pl[q] = (fd_h)(xv[k][q] - (float)pk[q]);
      }
      *(op_hx4*)(op_smem + r * XROW + 8 * c4) = pk;
      *(op_hx4*)(op_smem + XLO + r * XROW + 8 * c4) = pl;
    }
  }
  FD_STAMP(3);
  __syncthreads();
  FD_STAMP(4);
  f32x16 acc[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[0][r] = acc[1][r] = 0.f;
  const char* xa = op_smem + li * XROW + 16 * hi;
  auto a_frag = [&](int s, int rt, int lo) { return __builtin_bit_cast(hx8, *(const u16x8*)(xa + lo * XLO + rt * 32 * XROW + 32 * s)); };
  // activation fragments one k-step ahead (a read issued at the top of its own step is waited for by the step's first product)
  hx8 ah[2][2], al[2][2];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt) { ah[0][rt] = a_frag(0, rt, 0); al[0][rt] = a_frag(0, rt, 1); }
#pragma unroll
  for (int s = 0; s < KSL; ++s) {
#ifndef OP_ABL
#define OP_ABL 0  // timing ablation (tools/micro/op_bench.hip; results wrong): 1 = no weight stream beyond the first OP_DEPTH - 1 fragments (the
#endif            // upper bound of what larger row tiles / a weight-stationary split could save)
    if (s + OP_DEPTH - 1 < KSL && !(OP_ABL & 1)) {
      Wh[(s + OP_DEPTH - 1) % OP_DEPTH] = __builtin_bit_cast(hx8, *(const u16x8*)(wh + (size_t)(s + OP_DEPTH - 1) * 1024));
      Wl[(s + OP_DEPTH - 1) % OP_DEPTH] = __builtin_bit_cast(hx8, *(const u16x8*)(wl + (size_t)(s + OP_DEPTH - 1) * 1024));
    }
    if (s + 1 < KSL)
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) { ah[(s + 1) & 1][rt] = a_frag(s + 1, rt, 0); al[(s + 1) & 1][rt] = a_frag(s + 1, rt, 1); }
    const hx8 bh = Wh[s % OP_DEPTH], bl = Wl[s % OP_DEPTH];
    acc[0] = fd_mfma32(ah[s & 1][0], bl, acc[0]);
    acc[1] = fd_mfma32(ah[s & 1][1], bl, acc[1]);
    acc[0] = fd_mfma32(al[s & 1][0], bh, acc[0]);
    acc[1] = fd_mfma32(al[s & 1][1], bh, acc[1]);
    acc[0] = fd_mfma32(ah[s & 1][0], bh, acc[0]);
    acc[1] = fd_mfma32(ah[s & 1][1], bh, acc[1]);
    __builtin_amdgcn_sched_barrier(0);  // pin the step: hipcc otherwise sinks the reads of step s + 1 to their uses
  }
  FD_STAMP(5);
  // D[row, column]: lane = column 32 wave + li, registers = rows c_row(r, lane) of a row tile.  Stored from here a store instruction
  // would move 2 x 128 B (measured: 9.7 k cycles for the 64 KB of a block); the tile crosses LDS instead (the activation rows are dead)
  // and leaves as whole 1 KB rows, 16 B per lane.  Row stride 264 floats: the two lane halves (rows 4 apart) land 32 banks apart.
  constexpr int OROW = 264;
  static_assert(64 * OROW * 4 <= 2 * XLO, "output tile fits the activation buffers");
  __syncthreads();
  float* ot = (float*)op_smem;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[(32 * rt + c_row(r, lane)) * OROW + 32 * wave + li] = acc[rt][r];
  __syncthreads();
  FD_STAMP(6);
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias && z == 0) bv = *(const f32x4*)(bias + 4 * lane);
  float* out = parts + z * part_stride;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int lr = 8 * wave + k, m = m0 + lr;
    f32x4 v = *(const f32x4*)(ot + lr * OROW + 4 * lane);
    const float mk = rm[lr];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = (v[q] + bv[q]) * mk;
    if (m < M) *(f32x4*)(out + (long)m * ldo + 4 * lane) = v;
  }
  FD_STAMP(7);
}
#define OP_KSL 28
#define OP_NS 6
int fd_outproj_split_supported(int N, int K) { return N == 256 && K == OP_KSL * OP_NS * 16; }
int fd_outproj_split_slices() { return OP_NS; }
// parts[z][M, ldo], z < fd_outproj_split_slices(): w_hi / w_lo = fd_chain_build_image / _lo of the [256, K] weight matrix
int fd_outproj_split(int M, int N, int K, const float* A, int lda, const void* w_hi, const void* w_lo, const float* bias, const float* rowmask,
                     float* parts, long part_stride, int ldo, hipStream_t st) {
  if (M <= 0 || !fd_outproj_split_supported(N, K) || (lda & 3) || (ldo & 3) || (part_stride & 3) || !w_hi || !w_lo) return FDIPT_EINVAL;
  constexpr size_t smem = (size_t)2 * 64 * (OP_KSL * 32 + 16) + 256;
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)outproj_split_kernel<OP_KSL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  hipLaunchKernelGGL((outproj_split_kernel<OP_KSL>), dim3(cdiv(M, 64), OP_NS), dim3(512), smem, st, M, A, lda, (const char*)w_hi, (const char*)w_lo,
                     K / 16, bias, rowmask, parts, part_stride, ldo);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
int fd_linear_splitk_a16(int M, int N, int K, int nsplit, const half_t* A, int lda, const void* W, int ldw, const float* bias,
                     const float* rowmask, float* parts, long part_stride, int ldo, hipStream_t st) {
  if (M <= 0 || N <= 0 || K <= 0 || nsplit < 1 || (K & 7) || (lda & 7) || (ldw & 7)) return FDIPT_EINVAL;
  const int kslice = ((K + nsplit - 1) / nsplit + 63) / 64 * 64;
  if ((long)kslice * (nsplit - 1) >= K) return FDIPT_EINVAL;  // an empty slice
  hipLaunchKernelGGL((linear_splitk_kernel<PrecHalf, half_t, half_t>), dim3(cdiv(M, 64), cdiv(N, 64), nsplit), dim3(FD_THREADS),
                     (size_t)2 * 128 * (64 + PrecHalf::PAD) * sizeof(half_t), st, M, N, K, kslice, A, lda, (const half_t*)W, ldw, bias, rowmask, parts, part_stride, ldo);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

__global__ void f32_to_half_kernel(long n, const float* __restrict__ in, half_t* __restrict__ out) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) out[i] = f2h(in[i]);
}
int fd_f32_to_half(long n, const float* in, half_t* out, hipStream_t st) {
  if (n <= 0) return FDIPT_OK;
  hipLaunchKernelGGL(f32_to_half_kernel, dim3((unsigned)((n + 255) / 256 > 2048 ? 2048 : (n + 255) / 256)), dim3(256), 0, st,
                     n, in, out);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

extern "C" {

int fdipt_linear(int precision, int M, int N, int K, const float* A, int lda, const void* W, int ldw, const float* bias,
                 const float* residual, int ldr, const float* rowmask, int relu, float* out, int ldo, fdipt_stream_t s) {
  return launch_linear(precision, M, N, K, A, lda, W, ldw, bias, residual, ldr, rowmask, relu, out, ldo, (hipStream_t)s);
}

int fdipt_layernorm(int M, int D, const float* x, const float* residual, const float* gamma, const float* beta,
                    const float* rowmask, float* out, fdipt_stream_t s) {
  return fd_layernorm(M, D, x, D, residual, D, gamma, beta, rowmask, out, D, (hipStream_t)s);
}

// MFMA fragment-map self-test: asymmetric 96x80x72 product (exercises edge guards) against fp64 host math.
int fdipt_selftest_mfma(int precision, double* max_err_host) {
  const int M = 96, N = 80, K = 72;
  float *hA = new float[M * K], *hW = new float[N * K], *hO = new float[M * N];
  for (int i = 0; i < M * K; ++i) hA[i] = (float)((i * 37 % 101) - 50) / 64.f;
  for (int i = 0; i < N * K; ++i) hW[i] = (float)((i * 53 % 89) - 44) / 32.f;
  float *dA, *dW, *dO;
  half_t* dWb;
  int rc = FDIPT_OK;
  if (hipMalloc(&dA, M * K * 4) != hipSuccess || hipMalloc(&dW, N * K * 4) != hipSuccess ||
      hipMalloc(&dO, M * N * 4) != hipSuccess || hipMalloc(&dWb, N * K * 2) != hipSuccess)
    return FDIPT_ELAUNCH;
  hipMemcpy(dA, hA, M * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, hW, N * K * 4, hipMemcpyHostToDevice);
  if (precision == FDIPT_PREC_HALF) {
    fd_f32_to_half(N * K, dW, dWb, 0);
    rc = launch_linear(precision, M, N, K, dA, K, dWb, K, nullptr, nullptr, 0, nullptr, 0, dO, N, 0);
  } else {
    rc = launch_linear(precision, M, N, K, dA, K, dW, K, nullptr, nullptr, 0, nullptr, 0, dO, N, 0);
  }
  if (hipDeviceSynchronize() != hipSuccess) rc = FDIPT_ELAUNCH;
  hipMemcpy(hO, dO, M * N * 4, hipMemcpyDeviceToHost);
  double me = 0;
  for (int m = 0; m < M; ++m)
    for (int n = 0; n < N; ++n) {
      double s = 0;
      for (int k = 0; k < K; ++k) s += (double)hA[m * K + k] * (double)hW[n * K + k];
      double e = s - hO[m * N + n];
      if (e < 0) e = -e;
      if (e > me) me = e;
    }
  if (max_err_host) *max_err_host = me;
  hipFree(dA); hipFree(dW); hipFree(dO); hipFree(dWb);
  delete[] hA; delete[] hW; delete[] hO;
  return rc;
}

const char* fdipt_version(void) { return "fdipt-hip 0.1 (gfx950)"; }
// Lengths N at which the half-precision forward switches kernel variants (a sample padded within one class keeps its bits):
// o_pair key passes at 320 / 640 / 960 (attention.hip: fd_opair), attention / sequence-attention key tiles per wave at 384 / 512 / 768
// (attention3.hip: fd_attention3, attention_seq.hip: fd_seq_attention_run), 16-row node-path kernels up to 512 (model.hip), register
// attention up to 1024.  framedipt_amd/sharding.py groups mixed-length batches by these classes; tests/test_host_cpu.py compares the lists.
int fdipt_kernel_class_bounds(int32_t* bounds_host, int capacity) {
  static const int32_t b[] = {320, 384, 512, 640, 768, 960, 1024};
  const int n = (int)(sizeof(b) / sizeof(b[0]));
  for (int i = 0; i < n && i < capacity; ++i) bounds_host[i] = b[i];
  return n;
}

}  // extern "C"
