// chain.hip — fused per-residue MLP chains (bf16 operands, fp32 accumulate) for the node path of the score network:
// StructureModuleTransition (ipa_pytorch.py:36-58), the transformer feed-forward + norm2 and out_proj + norm1
// (ipa_pytorch.py:433-443), post_tfmr (:539), the node embedder (score_network.py:86-96), in_proj and the small heads.
//
// M = B*N is only a few thousand rows, so these layers are launch/latency-bound as separate GEMM launches.  Here ONE
// WAVE owns 32 rows for a whole chain of up to 3 Linear layers (+ReLU) + residual + LayerNorm + mask:
//   * transposed MFMA scheme of edge_transition2.hip: D[out feature, row] = W * X^T, activations stay in registers as
//     B fragments between layers (C/D fragment -> B fragment with the 16-wise k permutation folded into the weights);
//   * weight fragments are read STRAIGHT FROM L2 in fragment order ([tile][k-step][lane][8 bf16] images built at
//     prepare time: one coalesced 1 KB load per MFMA) through a 16-deep register ring — no LDS, no barrier, waves are
//     independent (grid = M/32 single-wave blocks spread over the CUs);
//   * two output tiles are streamed at once so consecutive MFMAs never share an accumulator.
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

__host__ __device__ __forceinline__ int ch_perm16(int pos) { return 4 * (pos >> 3) + (pos & 3) + 8 * ((pos & 7) >> 2); }

// ------------------------------------------------------------------ weight image
// img[((T*KS + s)*64 + lane)*8 + e] = W[32T + (lane&31)][k], k = 16s + 8(lane>>5) + e  (natural)  or
//                                                             16s + perm16(8(lane>>5) + e) (permuted: layers >= 2)
// lo: the image of W - half(W) (second part of a split operand) instead of half(W)
__global__ void chain_image_kernel(const float* __restrict__ w, int N, int K, int ldw, int permuted, int NT, int KS, float scale,
                                   int lo, half_t* __restrict__ img) {
  const long n = (long)NT * KS * 64 * 8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long ts = i >> 9;
    const int s = (int)(ts % KS), T = (int)(ts / KS);
    const int row = 32 * T + (lane & 31), pos = 8 * (lane >> 5) + e;
    const int k = 16 * s + (permuted ? ch_perm16(pos) : pos);
    const float v = (row < N && k < K) ? w[(long)row * ldw + k] * scale : 0.f;
    img[i] = lo ? f2h(v - h2f(f2h(v))) : f2h(v);
  }
}
size_t fd_chain_image_bytes(int N, int K) { return (size_t)((N + 31) / 32) * ((K + 15) / 16) * 1024; }
int fd_chain_build_image_scaled(const float* w, int N, int K, int ldw, int permuted, float scale, void* img, hipStream_t st) {
  const int NT = (N + 31) / 32, KS = (K + 15) / 16;
  hipLaunchKernelGGL(chain_image_kernel, dim3(64), dim3(256), 0, st, w, N, K, ldw, permuted, NT, KS, scale, 0, (half_t*)img);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
int fd_chain_build_image_lo(const float* w, int N, int K, int ldw, void* img, hipStream_t st) {
  const int NT = (N + 31) / 32, KS = (K + 15) / 16;
  hipLaunchKernelGGL(chain_image_kernel, dim3(64), dim3(256), 0, st, w, N, K, ldw, 0, NT, KS, 1.0f, 1, (half_t*)img);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
int fd_chain_build_image_ex(const float* w, int N, int K, int ldw, int permuted, int lo, void* img, hipStream_t st) {
  const int NT = (N + 31) / 32, KS = (K + 15) / 16;
  hipLaunchKernelGGL(chain_image_kernel, dim3(64), dim3(256), 0, st, w, N, K, ldw, permuted, NT, KS, 1.0f, lo, (half_t*)img);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
// Fragment image for v_mfma_f32_16x16x32_f16 (16-row node-path blocks, rowblock.hip: tfmr_tail16_kernel): tiles of 16 output features,
// k-steps of 32 — element (T, s, lane, e) = W[16 T + lane % 16][32 s + 8 (lane / 16) + e]; lo = 1: the image of W - half(W).
// Same size as the 32-row image of the matrix.  Needs N % 16 == 0; K is zero-padded to Kpad (a multiple of 32).
__global__ void chain_image16_kernel(const float* __restrict__ w, int K, int ldw, int NT, int KS, int lo, half_t* __restrict__ img) {  // (columns >= K: zeros)
  const long n = (long)NT * KS * 64 * 8;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const int e = (int)(i & 7), lane = (int)((i >> 3) & 63);
    const long ts = i >> 9;
    const int s = (int)(ts % KS), T = (int)(ts / KS);
    const int kcol = 32 * s + 8 * (lane >> 4) + e;
    const float v = kcol < K ? w[(long)(16 * T + (lane & 15)) * ldw + kcol] : 0.f;
    img[i] = lo ? f2h(v - h2f(f2h(v))) : f2h(v);
  }
}
int fd_chain_build_image16(const float* w, int N, int K, int Kpad, int ldw, int lo, void* img, hipStream_t st) {
  if ((N & 15) || (Kpad & 31) || K > Kpad) return FDIPT_ESIZE;
  hipLaunchKernelGGL(chain_image16_kernel, dim3(64), dim3(256), 0, st, w, K, ldw, N / 16, Kpad / 32, lo, (half_t*)img);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
int fd_chain_build_image(const float* w, int N, int K, int ldw, int permuted, void* img, hipStream_t st) {
  return fd_chain_build_image_scaled(w, N, K, ldw, permuted, 1.0f, img, st);
}

// ------------------------------------------------------------------ device pieces
__device__ __forceinline__ hx8 ch_pack8(const float* v) {
  hx8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (fd_h)v[e];
  return o;
}
#ifndef CH_ABL
#define CH_ABL 0  // timing ablations for tools/micro/chain_bench.hip (results become wrong); always 0 in the library
#endif
typedef fd_h hx4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void ch_lds_t;
typedef __attribute__((address_space(1))) const void ch_gl_t;

// LDS-DMA of `bytes` (multiple of 1 KB) from a fragment image: linear copy, 256 threads
__device__ __forceinline__ void ch_dma(const char* __restrict__ src, char* dst, int bytes, int tid) {
  for (int off = 0; off < bytes; off += FD_THREADS * 16)
    if (off + tid * 16 < bytes)
      __builtin_amdgcn_global_load_lds((ch_gl_t*)(src + off + tid * 16), (ch_lds_t*)(dst + off + (tid & ~63) * 16), 16, 0, 0);
}
template <int K>
__device__ __forceinline__ void ch_wait_barrier() {  // all but the newest K vector-memory ops of this wave retired, then barrier
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt((K & 15) | ((K >> 4) << 14) | 0x0070);  // vmcnt(K) lgkmcnt(0)
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_barrier();  // bare barrier: __syncthreads()' release fence would drain vmcnt to 0
  __builtin_amdgcn_sched_barrier(0);
}
template <int I>
using ch_ic = std::integral_constant<int, I>;
template <int B, int E, class F>
__device__ __forceinline__ void ch_static_for(F&& f) {
  if constexpr (B < E) {
    f(ch_ic<B>{});
    ch_static_for<B + 1, E>(f);
  }
}

// two output tiles of one layer, D^T[feature, row] = W X^T, from an LDS-resident fragment image of the pair:
// [tile A: KS KB][tile B: KS KB]; the result chains into the next layer's B operand without leaving registers.
template <int KS, bool HASB>
#ifndef CH_DEPTH
#define CH_DEPTH 4
#endif
__device__ __forceinline__ void ch_pair(f32x16& accA, f32x16& accB, const char* wp, const hx8* Bin, int lane) {
  constexpr int DEPTH = CH_DEPTH;
  const char* pa = wp + lane * 16;
  const char* pb = pa + (HASB ? KS * 1024 : 0);
  hx8 ringA[DEPTH], ringB[DEPTH];
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) {
    ringA[s] = __builtin_bit_cast(hx8, *(const u16x8*)(pa + s * 1024));
    if (HASB) ringB[s] = __builtin_bit_cast(hx8, *(const u16x8*)(pb + s * 1024));
  }
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + DEPTH - 1 < KS) {
      ringA[(s + DEPTH - 1) % DEPTH] = __builtin_bit_cast(hx8, *(const u16x8*)(pa + (s + DEPTH - 1) * 1024));
      if (HASB) ringB[(s + DEPTH - 1) % DEPTH] = __builtin_bit_cast(hx8, *(const u16x8*)(pb + (s + DEPTH - 1) * 1024));
    }
    if (!(CH_ABL & 8)) accA = fd_mfma32(ringA[s % DEPTH], Bin[s], accA);
    if (HASB && !(CH_ABL & 8)) accB = fd_mfma32(ringB[s % DEPTH], Bin[s], accB);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// ------------------------------------------------------------------ compile-time schedule of a chain
// K0: input width; NH1/NH2: hidden widths (0 = absent); NOUT: output width; FLAGS bit0/1: ReLU after hidden 1/2,
// bit2: LayerNorm; YS: single-layer kinds only - the output tile pairs are dealt to YS blocks (gridDim.y).
template <int K0, int NH1, int NH2, int NOUT, int FLAGS, int YS>
struct ChainShape {
  static constexpr int KS0 = (K0 + 15) / 16, KS1 = NH1 / 16, KS2 = NH2 / 16;
  static constexpr int NT1 = NH1 / 32, NT2 = NH2 / 32, NTO = (NOUT + 31) / 32;
  static constexpr int NL = 1 + (NT1 > 0) + (NT2 > 0);
  static constexpr bool LN = (FLAGS & 4) != 0;
  static constexpr int KSMAX = (KS0 > KS1 ? (KS0 > KS2 ? KS0 : KS2) : (KS1 > KS2 ? KS1 : KS2));
  static constexpr int PAIRB = 2 * KSMAX * 1024;
  static constexpr int NBUF = 3;
  static constexpr int NPAD = 32 * NTO;
  static constexpr int ks(int l) { return l == 0 ? KS0 : (l == 1 ? KS1 : KS2); }
  static constexpr int nt(int l) { return l == NL - 1 ? NTO : (l == 0 ? NT1 : NT2); }
  static constexpr int npair(int l) { return l == NL - 1 ? ((NTO + 1) / 2) / YS : (nt(l) + 1) / 2; }
  static constexpr int NSTEP = npair(0) + (NL > 1 ? npair(1) : 0) + (NL > 2 ? npair(2) : 0);
  static constexpr int layer_of(int i) { return i < npair(0) ? 0 : (i < npair(0) + npair(1) || NL < 3 ? 1 : 2); }
  static constexpr int pair_of(int i) { return i - (layer_of(i) > 0 ? npair(0) : 0) - (layer_of(i) > 1 ? npair(1) : 0); }
  static constexpr int tiles_in(int i) { return (YS == 1 && 2 * pair_of(i) + 1 >= nt(layer_of(i))) ? 1 : 2; }
  static constexpr int bytes(int i) { return tiles_in(i) * ks(layer_of(i)) * 1024; }
  static constexpr int ninstr(int i) { return bytes(i) / (FD_THREADS * 16); }  // DMA instructions EVERY wave issues for step i
  static constexpr int STAGE = 4 * 32 * 256;  // per wave: [32 rows][64 fp32] output tile pair / [32][64 bf16] input chunk
  static constexpr int NCONST = NH1 + NH2 + (LN ? 3 : 1) * NPAD;  // biases (+ gamma, beta)
  static constexpr size_t SMEM = (size_t)NBUF * PAIRB + STAGE + (size_t)(NCONST + 256) * 4 + 16;
  static_assert(SMEM <= 160 * 1024, "LDS budget");
  static_assert(NOUT % 32 == 0, "output width must be whole 32-feature tiles");
  static_assert(YS == 1 || (NL == 1 && !LN && (NTO % (2 * YS)) == 0), "tile pairs can only be dealt out for single-layer kinds");
};

// ------------------------------------------------------------------ kernel
// A 256-thread block owns 128 rows (32 per wave).
//  * weights: fragment images stream L2 -> LDS by DMA, one tile PAIR per step, through a 3-deep ring (two steps of lead);
//  * activations chain through registers (transposed MFMA scheme, see the header);
//  * every global access to activations is "16 lanes x 16 B per row" (256 B row segments: measured 8x faster than the
//    16 B-per-lane fragment pattern, tools/micro/io_pattern.hip); a wave-private LDS tile transposes between that layout
//    and the MFMA fragments — no block barrier involved, LDS operations of one wave execute in order;
//  * LayerNorm kinds park the pre-norm rows in `out` (L2) and normalise them in a second, equally coalesced pass.
template <int K0, int NH1, int NH2, int NOUT, int FLAGS, int YS>
__global__ __launch_bounds__(FD_THREADS, 1) void chain_kernel(ChainArgs a) {
  using S = ChainShape<K0, NH1, NH2, NOUT, FLAGS, YS>;
  constexpr int KS0 = S::KS0, NT1 = S::NT1, NT2 = S::NT2, NTO = S::NTO, NL = S::NL, PAIRB = S::PAIRB, NPAD = S::NPAD;
  constexpr int NSTEP = S::NSTEP;
  constexpr bool LN = S::LN;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* wbuf = smem;                                              // [NBUF][PAIRB]
  char* stage_all = smem + S::NBUF * PAIRB;                       // [4 waves][8 KB]
  float* cst = (float*)(smem + S::NBUF * PAIRB + S::STAGE);       // [b0: NH1][b1: NH2][bo][gamma][beta: NPAD][pre, post: 128]
  float* c_b0 = cst;
  float* c_b1 = c_b0 + NH1;
  float* c_bo = c_b1 + NH2;
  float* c_g = c_bo + NPAD;                                       // gamma, beta: LayerNorm kinds only
  float* c_bt = c_g + NPAD;
  float* c_pre = cst + S::NCONST;
  float* c_post = c_pre + 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  const int y = blockIdx.y;
  char* stage = stage_all + wave * 8192;
  // coalesced ("seg") layout of a 32-row x 64-column tile: pass it (0..7) -> row 4 it + sr, columns 4 sc .. 4 sc + 3
  const int sr = lane >> 4, sc = lane & 15;
  const int wrow0 = blockIdx.x * 128 + wave * 32;
  const float* bo_g = a.b[NL - 1];
  const bool has_res = a.residual != nullptr;
  const float* rbase = has_res ? a.residual : a.out;  // branch-free residual loads: a dummy (discarded) source otherwise
  const int rld = has_res ? a.ld_res : a.ld_out;
  int roff[8], ooff[8];   // element offsets of this lane's 8 rows in residual / out (row clamped for loads; M * ld < 2^31)
  bool rok[8];            // ... filled in at the first output-layer step (keeps them out of the hidden layers' registers)

  auto issue = [&](auto I) {  // DMA of step I's tile pair into ring slot I % NBUF
    constexpr int i = decltype(I)::value;
    constexpr int l = S::layer_of(i), j = S::pair_of(i);
    const char* src = (const char*)a.w[l];
    if (YS > 1) src += (size_t)(y + j * YS) * 2 * S::ks(l) * 1024;
    else src += (size_t)j * 2 * S::ks(l) * 1024;
    ch_dma(src, wbuf + (i % S::NBUF) * PAIRB, S::bytes(i), tid);
  };

  // ---- prologue: first chunk in flight, constants, input rows -> bf16 B fragments, second chunk
  issue(ch_ic<0>{});
  for (int v = tid; v < S::NCONST; v += FD_THREADS) {
    float x = 0.f;
    if (v < NH1) x = a.b[0][v];
    else if (v < NH1 + NH2) x = a.b[1][v - NH1];
    else if (v < NH1 + NH2 + NPAD) { const int c = v - NH1 - NH2; x = bo_g ? bo_g[c] : 0.f; }
    else if (v < NH1 + NH2 + 2 * NPAD) { const int c = v - NH1 - NH2 - NPAD; x = LN ? a.gamma[c] : 0.f; }
    else { const int c = v - NH1 - NH2 - 2 * NPAD; x = LN ? a.beta[c] : 0.f; }
    cst[v] = x;
  }
  if (tid < 128) {
    const int gr = blockIdx.x * 128 + tid < a.M ? blockIdx.x * 128 + tid : a.M - 1;
    c_pre[tid] = a.rowmask_pre ? a.rowmask_pre[gr] : 1.f;
    c_post[tid] = a.rowmask_post ? a.rowmask_post[gr] : 1.f;
  }
  hx8 X[KS0];
  {
    constexpr int NCH = (K0 + 63) / 64, GRP = 3;  // 64-column chunks, loaded GRP at a time (register pressure)
    ch_static_for<0, (NCH + GRP - 1) / GRP>([&](auto G) {
      constexpr int c0 = decltype(G)::value * GRP, c1 = c0 + GRP < NCH ? c0 + GRP : NCH;
      f32x4 xin[GRP][8];
#pragma unroll
      for (int c = c0; c < c1; ++c)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int gr = wrow0 + 4 * it + sr < a.M ? wrow0 + 4 * it + sr : a.M - 1;
          xin[c - c0][it] = f32x4{0.f, 0.f, 0.f, 0.f};
          if (64 * c + 4 * sc < K0) xin[c - c0][it] = *(const f32x4*)(a.in + gr * a.ld_in + 64 * c + 4 * sc);
        }
#pragma unroll
      for (int c = c0; c < c1; ++c) {
        // [32 rows][128 B] bf16 tile; 16 B unit u of row r lives at unit u ^ (r & 7)
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const int r = 4 * it + sr;
          hx4 pk;
#pragma unroll
          for (int q = 0; q < 4; ++q) pk[q] = (fd_h)xin[c - c0][it][q];
          *(hx4*)(stage + r * 128 + (((sc >> 1) ^ (r & 7)) << 4) + 8 * (sc & 1)) = pk;
        }
#pragma unroll
        for (int sp = 0; sp < 4; ++sp)
          if (4 * c + sp < KS0)
            X[4 * c + sp] = __builtin_bit_cast(hx8, *(const u16x8*)(stage + li * 128 + (((2 * sp + hi) ^ (li & 7)) << 4)));
      }
      __builtin_amdgcn_sched_barrier(0);
    });
  }
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (NSTEP > 1) {
    issue(ch_ic<1>{});
    ch_wait_barrier<S::ninstr(1)>();
  } else {
    ch_wait_barrier<0>();
  }
  // single-step kinds: nothing of the block is in flight any more -> touch the weights of the kernel launched next
  unsigned warm_tok = 0;
  if constexpr (NSTEP == 1) warm_tok = fd_l2_warm(a.warm, blockIdx.x * gridDim.y + blockIdx.y, gridDim.x * gridDim.y, tid, FD_THREADS);
  const float pre = c_pre[wave * 32 + li];

  hx8 H1[NT1 > 0 ? 2 * NT1 : 1], H2[NT2 > 0 ? 2 * NT2 : 1];
  float S1[8];  // LayerNorm: this lane's partial row sums (rows 4 it + sr)
#pragma unroll
  for (int it = 0; it < 8; ++it) S1[it] = 0.f;

  ch_static_for<0, NSTEP>([&](auto I) {
    constexpr int i = decltype(I)::value;
    constexpr int l = S::layer_of(i), j = S::pair_of(i), KS = S::ks(l), tiles = S::tiles_in(i);
    constexpr bool LAST = l == NL - 1;
    const int T0 = YS > 1 ? 2 * (y + j * YS) : 2 * j;
    // residual rows of this step's tile pair: ordinary loads issued BEFORE the step's DMA, consumed after its MFMA stream
    f32x4 rv[LAST ? 8 : 1];
    if constexpr (LAST && j == 0) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int gr = wrow0 + 4 * it + sr;
        rok[it] = gr < a.M;
        const int grc = rok[it] ? gr : a.M - 1;
        roff[it] = grc * rld + 4 * sc;
        ooff[it] = grc * a.ld_out + 4 * sc;
      }
    }
    if constexpr (LAST) {
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        rv[it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (!(CH_ABL & 2) && (tiles == 2 || sc < 8)) rv[it] = *(const f32x4*)(rbase + roff[it] + 32 * T0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (i + 2 < NSTEP && !(CH_ABL & 4)) issue(ch_ic<i + 2>{});
    f32x16 accA, accB;
#pragma unroll
    for (int r = 0; r < 16; ++r) { accA[r] = 0.f; accB[r] = 0.f; }
    const hx8* Bin = l == 0 ? X : (l == 1 ? H1 : H2);
    ch_pair<KS, tiles == 2>(accA, accB, wbuf + (i % S::NBUF) * PAIRB, Bin, lane);
    // chunk i+1 has landed everywhere and slot i % NBUF is free again after this barrier; chunk i+2 stays in flight
    if constexpr (i + 1 < NSTEP) ch_wait_barrier<(i + 2 < NSTEP ? S::ninstr(i + 2) : 0)>();
    if constexpr (!LAST) {  // hidden layer: + bias, ReLU, C/D fragment -> two B fragments of the next layer
      const float* cb = l == 0 ? c_b0 : c_b1;
      hx8* Hn = l == 0 ? H1 : H2;
      constexpr bool RELU = (FLAGS >> l) & 1;
#pragma unroll
      for (int u = 0; u < tiles; ++u) {
        const f32x16& acc = u ? accB : accA;
        float v[16];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *(const f32x4*)(cb + 32 * (2 * j + u) + 8 * g + 4 * hi);
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            v[4 * g + q] = acc[4 * g + q] + bv[q];
            if (RELU) v[4 * g + q] = fmaxf(v[4 * g + q], 0.f);
          }
        }
        Hn[2 * (2 * j + u)] = ch_pack8(v);
        Hn[2 * (2 * j + u) + 1] = ch_pack8(v + 8);
      }
    } else {
      // output pair: (acc + bias) * pre-mask -> wave-private [32 rows][64 fp32] tile (16 B unit u of row r at u ^ (r & 15))
#pragma unroll
      for (int u = 0; u < tiles; ++u) {
        const f32x16& acc = u ? accB : accA;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *(const f32x4*)(c_bo + 32 * (T0 + u) + 8 * g + 4 * hi);
          f32x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = (acc[4 * g + q] + bv[q]) * pre;
          *(f32x4*)(stage + li * 256 + (((8 * u + 2 * g + hi) ^ (li & 15)) << 4)) = o;
        }
      }
      // ... read back as row segments, + residual; plain kinds finish here, LayerNorm kinds park the row and keep sums
#pragma unroll
      for (int it = 0; it < 8; ++it) {
        const int r = 4 * it + sr;
        f32x4 v = *(const f32x4*)(stage + r * 256 + ((sc ^ (r & 15)) << 4));
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] += has_res ? rv[it][q] : 0.f;
        if constexpr (LN) {
          S1[it] += (v[0] + v[1]) + (v[2] + v[3]);
        } else {
          const float pm = c_post[wave * 32 + r];
#pragma unroll
          for (int q = 0; q < 4; ++q) v[q] *= pm;
        }
        if ((!(CH_ABL & 1) || v[0] == 1234.5f) && rok[it] && (tiles == 2 || sc < 8)) {
          *(f32x4*)(a.out + ooff[it] + 32 * T0) = v;
          if (!LN && a.out_h16) {  // optional bf16 copy of the rows (row stride NOUT): consumed by LDS-DMA in edge_transition3
            hx4 hb;
#pragma unroll
            for (int q = 0; q < 4; ++q) hb[q] = (fd_h)v[q];
            *(hx4*)(a.out_h16 + (long)(wrow0 + 4 * it + sr) * NOUT + 4 * sc + 32 * T0) = hb;
          }
        }
      }
    }
  });

  if constexpr (LN) {  // torch.nn.LayerNorm over the NOUT features of a row (16 lanes x NOUT/64 chunks), * post-mask
    constexpr int NCH = (NOUT + 63) / 64;
    f32x4 V[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int it = 0; it < 8; ++it) {  // every lane re-reads exactly the values it parked above
        V[c][it] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (64 * c + 4 * sc < NOUT) V[c][it] = *(const f32x4*)(a.out + ooff[it] + 64 * c);
      }
    float mu[8], rs[8];
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      float t = S1[it];
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) t += __shfl_xor(t, o, 64);
      mu[it] = t * (1.0f / NOUT);
    }
#pragma unroll
    for (int it = 0; it < 8; ++it) {
      float t = 0.f;
#pragma unroll
      for (int c = 0; c < NCH; ++c)
        if (64 * c + 4 * sc < NOUT) {
#pragma unroll
          for (int q = 0; q < 4; ++q) { const float dlt = V[c][it][q] - mu[it]; t += dlt * dlt; }
        }
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) t += __shfl_xor(t, o, 64);
      rs[it] = 1.0f / sqrtf(t * (1.0f / NOUT) + 1e-5f);
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
      if (64 * c + 4 * sc < NOUT) {
        const f32x4 gm = *(const f32x4*)(c_g + 64 * c + 4 * sc), bt = *(const f32x4*)(c_bt + 64 * c + 4 * sc);
#pragma unroll
        for (int it = 0; it < 8; ++it) {
          const float pm = c_post[wave * 32 + 4 * it + sr];
          f32x4 o;
#pragma unroll
          for (int q = 0; q < 4; ++q) o[q] = ((V[c][it][q] - mu[it]) * rs[it] * gm[q] + bt[q]) * pm;
          if (rok[it]) *(f32x4*)(a.out + ooff[it] + 64 * c) = o;
        }
      }
  }
  fd_l2_warm_done(warm_tok);
}

template <int K0, int NH1, int NH2, int NOUT, int FLAGS, int YS = 1>
static int ch_launch(const ChainArgs& a, hipStream_t st) {
  using S = ChainShape<K0, NH1, NH2, NOUT, FLAGS, YS>;
  if (a.M <= 0 || (a.ld_in & 3) || (a.ld_out & 3) || (a.residual && (a.ld_res & 3)) || (NOUT & 3)) return FDIPT_EINVAL;
  if (S::LN && a.residual == a.out) return FDIPT_EINVAL;  // pre-norm rows are parked in `out`
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)chain_kernel<K0, NH1, NH2, NOUT, FLAGS, YS>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)S::SMEM) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  hipLaunchKernelGGL((chain_kernel<K0, NH1, NH2, NOUT, FLAGS, YS>), dim3(cdiv(a.M, 128), YS), dim3(FD_THREADS), S::SMEM, st, a);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// shapes of the reference network (c_s 256, c_skip 64 -> d_t 320); returns FDIPT_EINVAL for anything else
int fd_chain(int kind, const ChainArgs& a, hipStream_t st) {
  switch (kind) {
    case FD_CHAIN_TRANSITION: return ch_launch<256, 256, 256, 256, 1 | 2 | 4>(a, st);   // t1 relu t2 relu t3 +res LN *mask
    case FD_CHAIN_FFN: return ch_launch<320, 320, 0, 320, 1 | 4>(a, st);                // l1 relu l2 +res LN
    case FD_CHAIN_OUTPROJ: return ch_launch<320, 0, 0, 320, 4>(a, st);                  // out_proj +res LN
    case FD_CHAIN_POST: return ch_launch<320, 0, 0, 256, 0, 4>(a, st);                  // post_tfmr +res
    case FD_CHAIN_INPROJ: return ch_launch<320, 0, 0, 960, 0, 5>(a, st);                // in_proj (15 tile pairs / 5)
    case FD_CHAIN_SKIP: return ch_launch<256, 0, 0, 64, 0>(a, st);                      // skip_embed
    case FD_CHAIN_ETINIT: return ch_launch<256, 0, 0, 128, 0, 2>(a, st);                // EdgeTransition.initial_embed
    case FD_CHAIN_A1: return ch_launch<128, 0, 0, 384, 0, 6>(a, st);                    // W1[:, e_i cols] e_i + b1
    case FD_CHAIN_AF: return ch_launch<128, 0, 0, 128, 0, 2>(a, st);                    // Wf[:, e_i cols] e_i + bf
    case FD_CHAIN_NODE_EMBED_72: return ch_launch<72, 256, 256, 256, 1 | 2 | 4>(a, st); // node embedder (de novo)
    case FD_CHAIN_NODE_EMBED_88: return ch_launch<88, 256, 256, 256, 1 | 2 | 4>(a, st); // node embedder (aatype)
    case FD_CHAIN_TORSION: return ch_launch<256, 256, 0, 256, 1>(a, st);                // l1 relu l2 +res
    default: return FDIPT_EINVAL;
  }
}
