// pair_mlp.hip — the N^2-row MLPs of the score network, fused per tile of pair rows.
//
//   edge_embed_kernel      : Embedder.forward pair branch, framedipt/model/score_network.py:98-105,173-196
//                            (cross-concat | rel-idx embedding | self-conditioning distogram -> 3-layer MLP -> LN)
//   edge_transition_kernel : EdgeTransition.forward, framedipt/model/ipa_pytorch.py:84-102
//
// A block owns TM consecutive pair rows p = (b*N+i)*N+j.  Activations never leave LDS between layers
// ([TM][K] operand-precision buffers); weights stream L2 -> LDS in [TN][32] tiles; the [N^2,384] concat
// tensor, the [N^2,120] feature tensor and the distogram of the reference are never materialised.
// Data layout in HBM: z[B,N,N,c_z] (ZT = float in fp32 mode, bf16 in bf16 mode), row-major.
#include "common.hpp"
#include "kernels.hpp"

template <class ZT>
__device__ __forceinline__ float z_load(const ZT* p) {
  if constexpr (sizeof(ZT) == 4) return *p; else return h2f(*p);
}
template <class ZT>
__device__ __forceinline__ void z_store(ZT* p, float v) {
  if constexpr (sizeof(ZT) == 4) *p = v; else *p = f2h(v);
}

// acc(32x32 per wave) = Act[TM x K] * W[n0 .. n0+TN, K]^T, Act resident in LDS, W streamed through Ws.
template <class P, class WT, int WR, int WC>
__device__ __forceinline__ void act_gemm(f32x16& acc, const typename P::T* act, int lda, int K,
                                         const WT* __restrict__ W, int ldw, int n0, int n_rows_w,
                                         typename P::T* Ws, int tid) {
  constexpr int LDT = P::BK + P::PAD;
  const int lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  if constexpr (sizeof(WT) == 4) {  // fp32 weights: the next k-tile's rows travel under this tile's products (common.hpp: StageRegs)
    StageRegs<32 * WC, P::BK> rw;
    stage_load<P, 32 * WC>(rw, W, ldw, n0, n_rows_w, 0, K, tid);
    for (int k0 = 0; k0 < K; k0 += P::BK) {
      stage_store<P, 32 * WC>(Ws, rw, tid);
      __syncthreads();
      if (k0 + P::BK < K) stage_load<P, 32 * WC>(rw, W, ldw, n0, n_rows_w, k0 + P::BK, K, tid);
      wave_mma<P>(acc, act + (wr * 32 + (lane & 31)) * lda + k0, Ws + (wc * 32 + (lane & 31)) * LDT, lane);
      __syncthreads();
    }
    return;
  }
  for (int k0 = 0; k0 < K; k0 += P::BK) {
    stage_tile<P, WT, 32 * WC>(Ws, W, ldw, n0, n_rows_w, k0, K, tid);
    __syncthreads();
    wave_mma<P>(acc, act + (wr * 32 + (lane & 31)) * lda + k0, Ws + (wc * 32 + (lane & 31)) * LDT, lane);
    __syncthreads();
  }
}

// LayerNorm of y[TM][CZ] (fp32 in LDS, row stride CZ+4) -> z_out rows, times mask_i*mask_j.
template <class ZT, int TM, int CZ>
__device__ __forceinline__ void ln_store(const float* ybuf, const float* __restrict__ gamma,
                                         const float* __restrict__ beta, const float* __restrict__ res_mask, long p0,
                                         long n_pairs, int N, ZT* __restrict__ z_out, float* __restrict__ trace,
                                         int tid) {
  constexpr int LDY = CZ + 4;
  constexpr int PER = (CZ + 63) / 64;
  const int lane = tid & 63, wave = tid >> 6;
  for (int m = wave; m < TM; m += FD_THREADS / 64) {
    const long p = p0 + m;
    if (p >= n_pairs) break;
    float v[PER];
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = lane + q * 64;
      v[q] = c < CZ ? ybuf[m * LDY + c] : 0.f;
      s += v[q];
    }
    const float mu = wave_sum(s) / (float)CZ;
    float qq = 0.f;
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = lane + q * 64;
      const float d = c < CZ ? v[q] - mu : 0.f;
      qq += d * d;
    }
    const float rstd = 1.0f / sqrtf(wave_sum(qq) / (float)CZ + 1e-5f);
    const long bi = p / N;  // b*N + i
    const int j = (int)(p - bi * N);
    const long bb = bi / N;
    const float em = res_mask[bi] * res_mask[bb * N + j];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int c = lane + q * 64;
      if (c < CZ) {
        const float o = ((v[q] - mu) * rstd * gamma[c] + beta[c]) * em;
        z_store<ZT>(z_out + p * CZ + c, o);
        if (trace) trace[p * CZ + c] = o;
      }
    }
  }
}


template <class P, class WT, class ZT, int TM, int WR, int WC, int CZ, int CB>
__global__ __launch_bounds__(FD_THREADS) void edge_transition_kernel(EdgeTransArgs a) {
  constexpr int H = CZ + 2 * CB;
  constexpr int LDA = H + P::PAD + (sizeof(typename P::T) == 2 ? 0 : 0);
  constexpr int LDT = P::BK + P::PAD;
  constexpr int TN = 32 * WC;
  constexpr int LDY = CZ + 4;
  constexpr size_t BUF = ((size_t)TM * LDA * sizeof(typename P::T) + 15) / 16 * 16;
  static_assert(BUF >= (size_t)TM * LDY * 4, "ybuf must fit in buf1");
  constexpr size_t WS = (size_t)TN * LDT * sizeof(typename P::T);
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF + WS];
  typename P::T* buf0 = (typename P::T*)smem;
  typename P::T* buf1 = (typename P::T*)(smem + BUF);
  typename P::T* Ws = (typename P::T*)(smem + 2 * BUF);
  float* ybuf = (float*)(smem + BUF);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int N = a.N;
  const long n_pairs = (long)a.B * N * N;
  const long p0 = (long)blockIdx.x * TM;
  const ZT* z_in = (const ZT*)a.z_in;

  // ---- stage X0 = [z_ij | e_i | e_j] into buf0
  for (int v = tid; v < TM * (H / 4); v += FD_THREADS) {
    const int m = v / (H / 4), c = (v % (H / 4)) * 4;
    const long p = p0 + m;
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (p < n_pairs) {
      const long bi = p / N;
      const int j = (int)(p - bi * N);
      const long bb = bi / N;
      if (c < CZ) {
#pragma unroll
        for (int q = 0; q < 4; ++q) x[q] = z_load<ZT>(z_in + p * CZ + c + q);
      } else if (c < CZ + CB) {
        const f32x4 t = *(const f32x4*)(a.e + bi * CB + (c - CZ));
        x[0] = t[0]; x[1] = t[1]; x[2] = t[2]; x[3] = t[3];
      } else {
        const f32x4 t = *(const f32x4*)(a.e + (bb * N + j) * CB + (c - CZ - CB));
        x[0] = t[0]; x[1] = t[1]; x[2] = t[2]; x[3] = t[3];
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) buf0[m * LDA + c + q] = P::from_f32(x[q]);
  }
  __syncthreads();

  f32x16 acc;
  const int ncol = wc * 32 + (lane & 31);
  // ---- layer 1: buf1 = relu(W1 x + b1)
  for (int n0 = 0; n0 < H; n0 += TN) {
    act_gemm<P, WT, WR, WC>(acc, buf0, LDA, H, (const WT*)a.w1, H, n0, H, Ws, tid);
    const int n = n0 + ncol;
    if (n < H) {
      const float bv = a.b1[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) buf1[(wr * 32 + c_row(r, lane)) * LDA + n] = P::from_f32(fmaxf(acc[r] + bv, 0.f));
    }
  }
  __syncthreads();
  // ---- layer 2 (+ residual): buf0 = relu(W2 h1 + b2) + x      (in place: element-wise same-thread RMW)
  for (int n0 = 0; n0 < H; n0 += TN) {
    act_gemm<P, WT, WR, WC>(acc, buf1, LDA, H, (const WT*)a.w2, H, n0, H, Ws, tid);
    const int n = n0 + ncol;
    if (n < H) {
      const float bv = a.b2[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        typename P::T* d = buf0 + (wr * 32 + c_row(r, lane)) * LDA + n;
        *d = P::from_f32(fmaxf(acc[r] + bv, 0.f) + P::to_f32(*d));
      }
    }
  }
  __syncthreads();
  // ---- final layer: y = Wf (h2 + x) + bf  -> ybuf (fp32, aliases buf1)
  for (int n0 = 0; n0 < CZ; n0 += TN) {
    act_gemm<P, WT, WR, WC>(acc, buf0, LDA, H, (const WT*)a.wf, H, n0, CZ, Ws, tid);
    const int n = n0 + ncol;
    if (n < CZ) {
      const float bv = a.bf[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) ybuf[(wr * 32 + c_row(r, lane)) * LDY + n] = acc[r] + bv;
    }
  }
  __syncthreads();
  ln_store<ZT, TM, CZ>(ybuf, a.gamma, a.beta, a.res_mask, p0, n_pairs, N, (ZT*)a.z_out, a.trace, tid);
}


// ------------------------------------------------------------------ fp32 EdgeTransition for the reference widths (round 2)
// Same scheme as edge_transition_kernel - 32 pair rows per block, activations resident in LDS, v_mfma_f32_32x32x2_f32 - with the
// weight stream software-pipelined: the (pass, k-tile) loops of the three layers are one flat sequence of 84 [128 x 32] weight
// tiles that travel L2 -> registers -> a ring of three LDS slots -> operand registers, each hop one tile ahead of its use: ONE
// barrier per k-tile and no exposed latency between the matrix instructions of consecutive tiles (the first generation loaded,
// stored, synchronised, multiplied and synchronised again for every tile: 1.0 k matrix cycles out of ~3 k per tile = the 34 % of the
// fp32 matrix peak it measured at).  Operands come out of LDS as 16-byte runs: within a group of 8 consecutive k the lane half hi
// reads k0 + 4 hi .. + 3 and MFMA j multiplies the pairs (k0 + j | k0 + 4 + j) - the same pairing on both operands, so no data is
// permuted (row strides 388 / 36 words: conflict-free ds_read_b128).  Everything a block reads before its first matrix instruction
// (X0 = [z | e_i | e_j], biases, pair masks, three weight tiles) is requested up front.
#define ETF_H 384
#define ETF_CZ 128
#define ETF_LDA 388
#define ETF_LDW 36
#define ETF_LDY 132
#define ETF_BUF (32 * ETF_LDA * 4)
#define ETF_WS (128 * ETF_LDW * 4)
#define ETF_LDS (2 * ETF_BUF + 3 * ETF_WS)
// The weight tiles of one block in stream order: layer 1 (3 passes x 12 k-tiles), layer 2 (3 x 12), final layer (1 x 12).
#define ETF_TILES 84
// phase profile (-DETF_PROF, tools/micro/etf_bench.hip): cycles of wave 0 of the first 256 blocks
#ifdef ETF_PROF2
__device__ unsigned long long etf_prof2[16];
#endif
#ifdef ETF_PROF
__device__ unsigned etf_prof[256 * 8];
#define ETF_STAMP(k)                                             \
  do {                                                           \
    __builtin_amdgcn_sched_barrier(0);                           \
    const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); \
    ph[k] += t_ - tlast;                                         \
    tlast = t_;                                                  \
    __builtin_amdgcn_sched_barrier(0);                           \
  } while (0)
#else
#define ETF_STAMP(k) \
  do {               \
  } while (0)
#endif
struct EtfTile {
  f32x4 r[4];
  __device__ __forceinline__ void load(const float* __restrict__ W, int n0, int k0, int tid) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = tid + u * FD_THREADS, row = v >> 3, kk = (v & 7) * 4;
      r[u] = *(const f32x4*)(W + (long)(n0 + row) * ETF_H + k0 + kk);
    }
  }
  __device__ __forceinline__ void store(float* Ws, int tid) const {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int v = tid + u * FD_THREADS, row = v >> 3, kk = (v & 7) * 4;
      *(f32x4*)(Ws + row * ETF_LDW + kk) = r[u];
    }
  }
};
struct EtfStream {
  const float *w1, *w2, *wf;
  __device__ __forceinline__ const float* ptr(int t) const {  // first element of weight tile t; the stream is cyclic (next block)
    t = t < ETF_TILES ? t : t - ETF_TILES;
    const float* W = t < 36 ? w1 : (t < 72 ? w2 : wf);
    const int tl = t < 36 ? t : (t < 72 ? t - 36 : t - 72);
    return W + (long)((tl / 12) * 128) * ETF_H + (tl % 12) * 32;
  }
  // the same as an integer, without a run-time choice among pointers (hipcc turns that into a table in scratch memory)
  __device__ __forceinline__ unsigned long addr(int t) const {
    t = t < ETF_TILES ? t : t - ETF_TILES;
    const unsigned long a1 = (unsigned long)w1, a2 = (unsigned long)w2, af = (unsigned long)wf;
    const unsigned long base = a1 + (t >= 36 ? a2 - a1 : 0ul) + (t >= 72 ? af - a2 : 0ul);
    const int tl = t < 36 ? t : (t < 72 ? t - 36 : t - 72);
    return base + ((unsigned long)((tl / 12) * 128) * ETF_H + (tl % 12) * 32) * 4;
  }
  __device__ __forceinline__ void load(EtfTile& r, int t, int tid) const {
    if (t >= ETF_TILES) return;
    const float* W = t < 36 ? w1 : (t < 72 ? w2 : wf);
    const int tl = t < 36 ? t : (t < 72 ? t - 36 : t - 72);
    r.load(W, (tl / 12) * 128, (tl % 12) * 32, tid);
  }
};
// MFMA operands of one k-tile in registers: 4 x 16-byte runs of the activation row and of the weight row of this lane
struct EtfOps {
  f32x4 a[4], w[4];
  __device__ __forceinline__ void read(const float* arow, const float* wrow) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      a[q] = *(const f32x4*)(arow + 8 * q);
      w[q] = *(const f32x4*)(wrow + 8 * q);
    }
  }
  __device__ __forceinline__ void mma(f32x16& acc, int q) const {
#pragma unroll
    for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][j], w[q][j], acc, 0, 0, 0);
  }
};
// One layer = NP passes x 12 k-tiles, starting at stream position t0.  Weight tiles live in a ring of THREE LDS slots (tile t in
// slot t % 3): during tile t the block multiplies from registers (operands of tile t, read during tile t - 1), reads the operands of
// tile t + 1 (stored during tile t - 1, visible since the barrier that closed it), stores tile t + 2 (requested from L2 during
// tile t - 1) into the slot tile t - 1 occupied, and requests tile t + 4.  Entry state: tiles t0, t0 + 1 in their slots, tile t0 + 2
// in g2, tile t0 + 3 in g0 (t0 a multiple of 6).  (The first tile of a layer reads its operands on entry: its activations were completed by the previous
// layer's last epilogue.)  epi(pass, acc) runs once per pass.
template <int NP, class Epi>
__device__ __forceinline__ void etf_layer(const float* act, const EtfStream& st, int t0, float* Ws0, EtfTile& g0, EtfTile& g1, EtfTile& g2,
                                          int tid, Epi epi) {
  const int lane = tid & 63, wc = tid >> 6, hi = lane >> 5;
  const float* arow = act + (lane & 31) * ETF_LDA + 4 * hi;
  const int woff = (wc * 32 + (lane & 31)) * ETF_LDW + 4 * hi;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  EtfOps o0, o1;
  __syncthreads();  // the previous layer's last epilogue / the prologue's stores
  o0.read(arow, Ws0 + (t0 % 3) * (ETF_WS / 4) + woff);
  // A step = the 16 matrix instructions of tile t with everything else of the pipeline placed in the gaps BETWEEN them, a few
  // instructions per gap (one wave per SIMD issues in order: whatever stands between two MFMAs is hidden only up to the 64 cycles
  // the first one runs; the first generations of this loop had loads, LDS reads, LDS stores and the barrier in three clumps and
  // lost ~600 of 1640 cycles per tile to them).  Gap 0: barrier (publishes the previous step's stores = tile t + 1); gaps 1-4: L2
  // requests of tile t + 4; gaps 5-12: operand reads of tile t + 1; gaps 13-16 (the last one after the 16th MFMA): LDS stores of
  // tile t + 2 (requested two steps earlier).  Timing ablations (tools/micro/etf_bench.hip -DETF_ABL=..., N = 300, B = 8): 4.74 ms
  // as is; 3.93 without the LDS stores, 4.15 without the L2 requests, 3.86 without both and the barrier: what is left above the
  // matrix work (3.15 ms) is the VGPR <-> LDS / L2 movement of the weight stream itself, not its latency.
  const int goff = ((tid >> 3) * ETF_H + (tid & 7) * 4), loff = (tid >> 3) * ETF_LDW + (tid & 7) * 4;
  auto step = [&](int tl, EtfOps& cur, EtfOps& nxt, EtfTile& gs, EtfTile& gl) {
    const int t = t0 + tl, kt = tl % 12;
    const float* src = st.ptr(t + 4) + goff;
    const float* an = arow + ((kt + 1) % 12) * 32;
    const float* wn = Ws0 + ((t + 1) % 3) * (ETF_WS / 4) + woff;
    float* wd = Ws0 + ((t + 2) % 3) * (ETF_WS / 4) + loff;  // (beyond the stream: a free slot nobody reads)
#define ETF_MMA(i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[(i) >> 2][(i) & 3], cur.w[(i) >> 2][(i) & 3], acc, 0, 0, 0); \
                   __builtin_amdgcn_sched_barrier(0)
#define ETF_GAP(...) __VA_ARGS__; __builtin_amdgcn_sched_barrier(0)
#ifndef ETF_ABL
#define ETF_ABL 0  // timing ablations of tools/micro/etf_bench.hip (results are wrong with any bit set): 1 no step barrier, 2 no LDS weight
                   // stores, 4 no L2 requests, 8 (wave-specialised kernel) the requests go to registers nobody waits for
#endif
#ifndef ETF_ORDER
#define ETF_ORDER 1  // 1: LDS stores right behind the barrier, L2 requests at the end of the step; 0: the other way round
#endif
#define ETF_ST(k) if (!(ETF_ABL & 2)) *(f32x4*)(wd + (k) * 32 * ETF_LDW) = gs.r[k]
#define ETF_LD(k) if (!(ETF_ABL & 4)) gl.r[k] = *(const f32x4*)(src + (k) * 32 * ETF_H)
    ETF_MMA(0);  ETF_GAP(if (!(ETF_ABL & 1)) __syncthreads());
    if (ETF_ORDER) {
      ETF_MMA(1);  ETF_GAP(ETF_ST(0));
      ETF_MMA(2);  ETF_GAP(ETF_ST(1));
      ETF_MMA(3);  ETF_GAP(ETF_ST(2));
      ETF_MMA(4);  ETF_GAP(ETF_ST(3));
    } else {
      ETF_MMA(1);  ETF_GAP(ETF_LD(0));
      ETF_MMA(2);  ETF_GAP(ETF_LD(1));
      ETF_MMA(3);  ETF_GAP(ETF_LD(2));
      ETF_MMA(4);  ETF_GAP(ETF_LD(3));
    }
    ETF_MMA(5);  ETF_GAP(nxt.a[0] = *(const f32x4*)(an));
    ETF_MMA(6);  ETF_GAP(nxt.w[0] = *(const f32x4*)(wn));
    ETF_MMA(7);  ETF_GAP(nxt.a[1] = *(const f32x4*)(an + 8));
    ETF_MMA(8);  ETF_GAP(nxt.w[1] = *(const f32x4*)(wn + 8));
    ETF_MMA(9);  ETF_GAP(nxt.a[2] = *(const f32x4*)(an + 16));
    ETF_MMA(10); ETF_GAP(nxt.w[2] = *(const f32x4*)(wn + 16));
    ETF_MMA(11); ETF_GAP(nxt.a[3] = *(const f32x4*)(an + 24));
    ETF_MMA(12); ETF_GAP(nxt.w[3] = *(const f32x4*)(wn + 24));
    if (ETF_ORDER) {
      ETF_MMA(13); ETF_GAP(ETF_LD(0));
      ETF_MMA(14); ETF_GAP(ETF_LD(1));
      ETF_MMA(15); ETF_GAP(ETF_LD(2));
      ETF_GAP(ETF_LD(3));
    } else {
      ETF_MMA(13); ETF_GAP(ETF_ST(0));
      ETF_MMA(14); ETF_GAP(ETF_ST(1));
      ETF_MMA(15); ETF_GAP(ETF_ST(2));
      ETF_GAP(ETF_ST(3));
    }
#undef ETF_ST
#undef ETF_LD
#undef ETF_MMA
#undef ETF_GAP
    if (kt == 11) {
      epi(tl / 12, acc);
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
  };
  // register roles: G[k] carries the tiles with index = k (mod 3): step t stores G[(t + 2) % 3] and re-loads G[(t + 1) % 3]
  for (int tl = 0; tl < NP * 12; tl += 6) {
    step(tl, o0, o1, g2, g1);
    step(tl + 1, o1, o0, g0, g2);
    step(tl + 2, o0, o1, g1, g0);
    step(tl + 3, o1, o0, g2, g1);
    step(tl + 4, o0, o1, g0, g2);
    step(tl + 5, o1, o0, g1, g0);
  }
}

// Persistent blocks (one per CU: 151 KB of LDS): a block walks row tiles blk, blk + gridDim.x, ...; the weight stream is cyclic
// (84 tiles, a multiple of the ring's 3 slots and of the 2 register roles), so the pipeline never drains between tiles, and the
// next tile's X0 / pair masks are requested before the final layer and stored once it has read buf0 for the last time.
// LayerNorm of the block's 32 rows in ybuf, times the pair mask, -> z (both fp32 kernels; round 6 form).  A lane owns 16 values of ONE row
// (row 8 wave + lane / 8, columns 32 q + 4 (lane % 8) ..: its four 16 B pieces of a row, eight lanes = 128 consecutive bytes per piece):
// the statistics are 15 lane-local additions + three exchange steps among eight neighbouring lanes, the output leaves as four 16 B stores.
// (Rounds 2 - 5: two columns of eight rows per lane - 2 x 6 butterfly steps x 8 rows of ds_bpermute and sixteen 4 B stores per lane,
//  ~5.5 k cycles per row tile with the matrix cores idle = 5 % of the launch, tools/micro/etf_bench.hip -DETF_ABL=64.)
struct EtfLnConst {
  f32x4 g[4], b[4];
  __device__ __forceinline__ void load(const float* gamma, const float* beta, int lane) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      g[q] = *(const f32x4*)(gamma + 32 * q + 4 * (lane & 7));
      b[q] = *(const f32x4*)(beta + 32 * q + 4 * (lane & 7));
    }
  }
};
template <class ZT>
__device__ __forceinline__ void etf_ln_store(const float* ybuf, const EtfLnConst& C, float em_row, long p0, long n_pairs, ZT* z_out, float* trace,
                                             int wc, int lane) {
  const int r = 8 * wc + (lane >> 3), c8 = lane & 7;
  f32x4 v[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) v[q] = *(const f32x4*)(ybuf + r * ETF_LDY + 32 * q + 4 * c8);
  float s1 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) s1 += (v[q][0] + v[q][1]) + (v[q][2] + v[q][3]);
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) s1 += __shfl_xor(s1, o, 64);
  const float mu = s1 * (1.0f / ETF_CZ);
  float s2 = 0.f;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
#pragma unroll
    for (int i = 0; i < 4; ++i) v[q][i] -= mu;
    s2 += (v[q][0] * v[q][0] + v[q][1] * v[q][1]) + (v[q][2] * v[q][2] + v[q][3] * v[q][3]);
  }
#pragma unroll
  for (int o = 1; o < 8; o <<= 1) s2 += __shfl_xor(s2, o, 64);
  const float rstd = 1.0f / sqrtf(s2 * (1.0f / ETF_CZ) + 1e-5f);
  const float em = __shfl(em_row, r, 64);
  const long p = p0 + r;
  if (p < n_pairs) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      f32x4 o;
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i] = (v[q][i] * rstd * C.g[q][i] + C.b[q][i]) * em;
      const long at = p * ETF_CZ + 32 * q + 4 * c8;
      if constexpr (sizeof(ZT) == 4) *(f32x4*)((float*)z_out + at) = o;
      else
#pragma unroll
        for (int i = 0; i < 4; ++i) z_store<ZT>(z_out + at + i, o[i]);
      if (trace) *(f32x4*)(trace + at) = o;
    }
  }
}

template <class ZT>
__global__ __launch_bounds__(FD_THREADS) void edge_transition_f32_kernel(EdgeTransArgs a, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  FD_CLK_BEGIN;
#ifdef ETF_PROF
  unsigned ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = (unsigned)__builtin_amdgcn_s_memtime();
#endif
  float* buf0 = (float*)smem;
  float* buf1 = (float*)(smem + ETF_BUF);
  float* Ws = (float*)(smem + 2 * ETF_BUF);
  float* ybuf = buf1;
  const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
  const int N = a.N;
  const long n_pairs = (long)a.B * N * N;
  const ZT* z_in = (const ZT*)a.z_in;
  const EtfStream st = {(const float*)a.w1, (const float*)a.w2, (const float*)a.wf};
  const int ncol = wc * 32 + (lane & 31);
  // ---- everything the block reads from memory before its first matrix instruction is requested up front: the first three
  // weight tiles, the 12 16-byte pieces per thread of X0 = [z_ij | e_i | e_j] (32 rows x 384), the seven bias values of this
  // lane's output columns and (lanes 0..31) the pair mask of a row
  EtfTile g0, g1, g2;
  st.load(g0, 0, tid);
  st.load(g1, 1, tid);
  st.load(g2, 2, tid);  // (entry state of a layer / row tile: G[2] = tile t0 + 2, G[0] = tile t0 + 3, both possibly still in flight)
  f32x4 xr[12];
  float em_next = 0.f;
  auto request_x0 = [&](long p0) {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int part = u >> 2, v = tid + (u & 3) * FD_THREADS, m = v >> 5, c = (v & 31) * 4;  // 32 pieces of 16 B per 128-float part
      const long pr = p0 + m, p = pr < n_pairs ? pr : n_pairs - 1;
      const long bi = p / N;
      const int j = (int)(p - bi * N);
      const long bb = bi / N;
      if (part == 0) {
        if constexpr (sizeof(ZT) == 4) xr[u] = *(const f32x4*)((const float*)z_in + p * ETF_CZ + c);
        else
#pragma unroll
          for (int q = 0; q < 4; ++q) xr[u][q] = z_load<ZT>(z_in + p * ETF_CZ + c + q);
      } else {
        xr[u] = *(const f32x4*)(a.e + (part == 1 ? bi : bb * N + j) * ETF_CZ + c);
      }
      // (rows beyond the last pair carry the last pair's values: they are never stored, and a select on the loaded value here would
      //  make the compiler wait for the requests on the spot)
    }
    em_next = 0.f;  // lanes 0..31 of every wave: pair mask of row `lane`
    if (lane < 32) {
      const long p = p0 + lane;
      if (p < n_pairs) {
        const long bi = p / N;
        em_next = a.res_mask[bi] * a.res_mask[(bi / N) * N + (p - bi * N)];
      }
    }
  };
  auto store_x0 = [&]() {
#pragma unroll
    for (int u = 0; u < 12; ++u) {
      const int part = u >> 2, v = tid + (u & 3) * FD_THREADS, m = v >> 5, c = (v & 31) * 4;
      *(f32x4*)(buf0 + m * ETF_LDA + part * ETF_CZ + c) = xr[u];
    }
  };
  int blk = blockIdx.x;
  request_x0((long)blk * 32);
  float bias1[3], bias2[3];
#pragma unroll
  for (int q = 0; q < 3; ++q) { bias1[q] = a.b1[q * 128 + ncol]; bias2[q] = a.b2[q * 128 + ncol]; }
  const float biasf = a.bf[ncol];
  EtfLnConst lnc;
    lnc.load(a.gamma, a.beta, lane);
  store_x0();
  g0.store(Ws, tid);
  g1.store(Ws + ETF_WS / 4, tid);
  st.load(g0, 3, tid);
  ETF_STAMP(0);
  for (; blk < n_blocks; blk += gridDim.x) {
    const long p0 = (long)blk * 32;
    const float em_row = em_next;
    // layer 1: buf1 = relu(W1 x + b1)   (a layer has an even number of tiles and 36 / 72 / 84 are multiples of 3: the register and
    // slot roles at the entry of every layer, and of every row tile, are the same)
    etf_layer<3>(buf0, st, 0, Ws, g0, g1, g2, tid, [&](int pass, const f32x16& acc) {
      const int n = pass * 128 + ncol;
      const float bv = pass == 0 ? bias1[0] : (pass == 1 ? bias1[1] : bias1[2]);
#pragma unroll
      for (int r = 0; r < 16; ++r) buf1[c_row(r, lane) * ETF_LDA + n] = fmaxf(acc[r] + bv, 0.f);
    });
    ETF_STAMP(1);
    // layer 2 (+ residual): buf0 = relu(W2 h1 + b2) + x   (in place: element-wise same-thread read-modify-write)
    etf_layer<3>(buf1, st, 36, Ws, g0, g1, g2, tid, [&](int pass, const f32x16& acc) {
      const int n = pass * 128 + ncol;
      const float bv = pass == 0 ? bias2[0] : (pass == 1 ? bias2[1] : bias2[2]);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float* d = buf0 + c_row(r, lane) * ETF_LDA + n;
        *d = fmaxf(acc[r] + bv, 0.f) + *d;
      }
    });
    ETF_STAMP(2);
    // the next row tile's inputs travel under the final layer
    if (blk + (int)gridDim.x < n_blocks) request_x0((long)(blk + gridDim.x) * 32);
    // final layer: y = Wf (h2 + x) + bf -> ybuf (aliases buf1, which the final layer does not read)
    etf_layer<1>(buf0, st, 72, Ws, g0, g1, g2, tid, [&](int, const f32x16& acc) {
#pragma unroll
      for (int r = 0; r < 16; ++r) ybuf[c_row(r, lane) * ETF_LDY + ncol] = acc[r] + biasf;
    });
    __syncthreads();
    ETF_STAMP(3);
    store_x0();  // buf0 is free (the next layer 1 starts with a barrier)
    etf_ln_store<ZT>(ybuf, lnc, em_row, p0, n_pairs, (ZT*)a.z_out, a.trace, wc, lane);
    ETF_STAMP(4);
  }
  FD_CLK_END(a.clock);
#ifdef ETF_PROF
  if (tid == 0 && blockIdx.x < 256)
    for (int k = 0; k < 8; ++k) etf_prof[blockIdx.x * 8 + k] = ph[k];
#endif
}

// ------------------------------------------------------------------ wave-specialised form (round 2, end)
// Same arithmetic and accumulation order as edge_transition_f32_kernel above (bit-identical outputs), but the weight stream no longer
// shares an instruction stream with the matrix cores: a block has EIGHT waves, 0..3 multiply (16 MFMAs + the 8 operand reads of the
// next tile per step, epilogues, LayerNorm), 4..7 move (round 6: per step five LDS-DMA requests of tile t + 3, see EtfDma; rounds 2 - 5: the
// 4 LDS stores of tile t + 2 and the 4 L2 requests of tile t + 4; per row tile: the next tile's X0 rows).  One wave per SIMD issues in order: every ds_write_b128 / global_load between two MFMAs of
// the fused kernel that took longer than the 64 cycles the matrix instruction runs was a bubble (ablations above: 0.8 ms of 4.74 for
// the stores, 0.6 ms for the requests).  Both roles execute the same sequence of barriers (one per layer entry, one per step, one after the final layer).
#ifndef ETF_TOUCH
#define ETF_TOUCH 0  // n > 0: the multiplier waves touch the L2 lines of tile t + n during step t (measured: no gain, the requests are L2 hits already)
#endif
template <int NP, class Epi>
__device__ __forceinline__ void etfs_layer_compute(const float* act, const EtfStream& st, int t0, float* Ws0, int tid, unsigned& tok, Epi epi) {
  const int lane = tid & 63, wc = tid >> 6, hi = lane >> 5;
  const float* arow = act + (lane & 31) * ETF_LDA + 4 * hi;
  const int woff = (wc * 32 + (lane & 31)) * ETF_LDW + 4 * hi;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  EtfOps o0, o1;
  __syncthreads();
  o0.read(arow, Ws0 + (t0 % 3) * (ETF_WS / 4) + woff);
#ifdef ETF_PROF2  // where a multiplier's step goes (tools/micro/etf_bench.hip): cycles at the step barrier / per step, wave 0 of every block
  unsigned long long p2_bar = 0, p2_tot = 0, p2_last = __builtin_amdgcn_s_memtime();
#define ETF_STEP_BARRIER()                                                         \
  do {                                                                             \
    const unsigned long long b0_ = __builtin_amdgcn_s_memtime();                   \
    __syncthreads();                                                               \
    const unsigned long long b1_ = __builtin_amdgcn_s_memtime();                   \
    p2_bar += b1_ - b0_; p2_tot += b1_ - p2_last; p2_last = b1_;                   \
  } while (0)
#else
#define ETF_STEP_BARRIER() do { if (!(ETF_ABL & 1)) __syncthreads(); } while (0)
#endif
  auto step = [&](int tl, EtfOps& cur, EtfOps& nxt) {
    const int t = t0 + tl, kt = tl % 12;
    const float* an = arow + ((kt + 1) % 12) * 32;
    const float* wn = Ws0 + ((t + 1) % 3) * (ETF_WS / 4) + woff;
#define ETF_MMA(i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a[(i) >> 2][(i) & 3], cur.w[(i) >> 2][(i) & 3], acc, 0, 0, 0); \
                   __builtin_amdgcn_sched_barrier(0)
#define ETF_GAP(...) __VA_ARGS__; __builtin_amdgcn_sched_barrier(0)
    ETF_MMA(0);  ETF_GAP(ETF_STEP_BARRIER());
    // one dword of each of the tile's 128 lines (row = wave * 32 + lane, 128 B per row) into ONE register that stays allocated for the
    // whole kernel (`tok`: a dead destination would be re-used by the compiler while the load is still in flight), never waited for: the
    // tile is in L2 when the movers ask for it four steps later (the z stream evicts the weights from the XCD's L2 between two uses)
    ETF_MMA(1);  ETF_GAP(if (ETF_TOUCH) asm volatile("global_load_dword %0, %1, off" : "+v"(tok) : "v"(st.addr(t + ETF_TOUCH) + (unsigned long)(wc * 32 + (lane & 31)) * (ETF_H * 4)) : "memory"));
    ETF_MMA(2);  ETF_GAP(if (!(ETF_ABL & 128)) nxt.a[0] = *(const f32x4*)(an));
    ETF_MMA(3);  ETF_GAP(if (!(ETF_ABL & 128)) nxt.w[0] = *(const f32x4*)(wn));
    ETF_MMA(4);
    ETF_MMA(5);  ETF_GAP(if (!(ETF_ABL & 128)) nxt.a[1] = *(const f32x4*)(an + 8));
    ETF_MMA(6);  ETF_GAP(if (!(ETF_ABL & 128)) nxt.w[1] = *(const f32x4*)(wn + 8));
    ETF_MMA(7);
    ETF_MMA(8);  ETF_GAP(if (!(ETF_ABL & 128)) nxt.a[2] = *(const f32x4*)(an + 16));
    ETF_MMA(9);  ETF_GAP(if (!(ETF_ABL & 128)) nxt.w[2] = *(const f32x4*)(wn + 16));
    ETF_MMA(10);
    ETF_MMA(11); ETF_GAP(if (!(ETF_ABL & 128)) nxt.a[3] = *(const f32x4*)(an + 24));
    ETF_MMA(12); ETF_GAP(if (!(ETF_ABL & 128)) nxt.w[3] = *(const f32x4*)(wn + 24));
    ETF_MMA(13);
    ETF_MMA(14);
    ETF_MMA(15);
#undef ETF_MMA
#undef ETF_GAP
    if (kt == 11) {
      if (ETF_ABL & 256) {  // (timing only: the pass epilogue never runs, the accumulators stay alive)
        float sum_ = 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) sum_ += acc[r];
        if (sum_ == 12345.678f) epi(tl / 12, acc);
      } else {
        epi(tl / 12, acc);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    }
  };
#pragma unroll 1
  for (int tl = 0; tl < NP * 12; tl += 6) {
    step(tl, o0, o1);
    step(tl + 1, o1, o0);
    step(tl + 2, o0, o1);
    step(tl + 3, o1, o0);
    step(tl + 4, o0, o1);
    step(tl + 5, o1, o0);
  }
#ifdef ETF_PROF2
  if (tid == 0) {
    atomicAdd(&etf_prof2[0], p2_bar); atomicAdd(&etf_prof2[1], p2_tot); atomicAdd(&etf_prof2[2], (unsigned long long)(NP * 12));
    atomicAdd(&etf_prof2[8 + 2 * (t0 / 36)], p2_bar); atomicAdd(&etf_prof2[9 + 2 * (t0 / 36)], p2_tot);  // per layer
  }
#endif
}
// The mover's side of a layer: tile t + 2 (in registers since two steps) -> its LDS slot, tile t + 4 requested (the register roles of
// etf_layer), all four mover waves every step; `mt` = 0..255.  With the requests ablated the launch takes 3.89 ms, with them 4.45:
// the movers still arrive late at some barriers.  What was tried against that and measured SLOWER (N = 300, B = 8; fused kernel 4.62 ms):
//  * six register sets per wave, requests eight tiles ahead, compiler-visible loads: across the back-edge of a rolled loop hipcc's vmcnt
//    bookkeeping falls back to vmcnt(0) - it waits for the requests it has just issued (5.07 ms); fully unrolled it spills 558 registers;
//  * the same with inline-asm requests and an explicit vmcnt(20) before a set is stored (exact waits in the ISA): 4.93 ms;
//  * two groups of half-tile movers with two sets each (four-step distance): vmcnt(0) again behind the wave-uniform branches, 5.29 ms;
//  * one mover wave per whole tile (four-step distance, exact wait): 16 ds_write_b128 + 16 requests do not fit one step, 6.7 ms
//    (7.1 ms while the run-time choice among the three weight pointers made the requests FLAT loads, which count in lgkmcnt: every
//    barrier then waited for them);
//  * L2 touches of tile t + 8 from the multiplier waves (ETF_TOUCH): 4.63 ms - the requests are L2 hits already.
// The decisive ablation (ETF_ABL = 8): the SAME requests into registers nobody ever waits for: 4.44 ms, i.e. nothing gained - not the
// latency.  Rounds 2 - 5 read that as "the traffic through the shared VGPR file"; round 6's s_memtime probe (-DETF_PROF2) found the
// movers 1.2 k cycles busy ISSUING a step's requests: their address arithmetic waits for a turn at the vector ALU behind the multiplier's
// back-to-back MFMAs (EtfDma below, ETF_DMA = 1, is the form that has none; this register form stays for A/B as ETF_DMA = 0).
template <int NP>
__device__ __forceinline__ void etfs_layer_move(const EtfStream& st, int t0, float* Ws0, EtfTile& g0, EtfTile& g1, EtfTile& g2, int mt,
                                                f32x4 (&g_dummy)[4]) {
  const int goff = ((mt >> 3) * ETF_H + (mt & 7) * 4), loff = (mt >> 3) * ETF_LDW + (mt & 7) * 4;
  __syncthreads();
  auto step = [&](int tl, EtfTile& gs, EtfTile& gl) {
    const int t = t0 + tl;
    const float* src = st.ptr(t + 4) + goff;
    float* wd = Ws0 + ((t + 2) % 3) * (ETF_WS / 4) + loff;
    __syncthreads();
    __builtin_amdgcn_sched_barrier(0);  // stores BEFORE the requests: hipcc hoists the loads otherwise and then has to wait for them
#pragma unroll                          // (vmcnt is in order) before it can store the tile requested two steps ago
    for (int k = 0; k < 4; ++k) if (!(ETF_ABL & 2)) *(f32x4*)(wd + k * 32 * ETF_LDW) = gs.r[k];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (ETF_ABL & 8) {  // timing experiment (wrong results): the same requests, into registers nobody waits for
        asm volatile("global_load_dwordx4 %0, %1, off" : "+v"(g_dummy[k]) : "v"(src + k * 32 * ETF_H) : "memory");
      } else if (!(ETF_ABL & 4)) {
        gl.r[k] = *(const f32x4*)(src + k * 32 * ETF_H);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  };
#pragma unroll 1
  for (int tl = 0; tl < NP * 12; tl += 6) {
    step(tl, g2, g1);
    step(tl + 1, g0, g2);
    step(tl + 2, g1, g0);
    step(tl + 3, g2, g1);
    step(tl + 4, g0, g2);
    step(tl + 5, g1, g0);
  }
}

// ------------------------------------------------------------------ the movers as LDS-DMA (round 6)
// The decisive ablation above says the cost of the weight stream is the traffic through the VGPR file of the SIMD a mover shares with a
// multiplier (16 B x 64 lanes x 4 requests written back per wave and step, read again by 4 ds_write_b128), not its latency.
// `global_load_lds_dwordx4` (the path the half-precision kernel streams its fragments on, edge_transition4.hip) takes the tile from L2 to
// LDS without touching a VGPR: an instruction writes 64 x 16 B = 1 KB of CONSECUTIVE LDS (M0 + 16 lane) from 64 free global addresses, so
// the padded slot layout ([128 rows][36 words], 9 pieces of 16 B per row, 18 KB = 18 instructions per tile) comes out of the address
// arithmetic of the lanes: piece u = 64 i + lane of the slot is (row u / 9, words 4 (u % 9) ..), the ninth piece of a row is padding (its
// lane asks for the eighth again).  Four mover waves, five instructions each per step (instructions 18, 19 repeat 14, 15: every wave
// waits with the same count).  Ring of three slots as before, but a tile now needs its slot only from the request on: during step t
// (behind the step's barrier, which has seen the last operand read of tile t) tile t + 3 is requested into slot t % 3; before a mover
// arrives at the next barrier it waits until the requests of the PREVIOUS step have landed (vmcnt counts an LDS-DMA down when the data
// is in LDS): tile t + 2 is complete when step t + 1 reads it - the same two steps of latency budget as the register ring had.
// The X0 rows of the next row tile still travel through registers (buf0 is the final layer's input until its last step): they are
// requested right behind the DMAs of the final layer's first step and the two waits that have them in front of the awaited DMAs allow
// 12 more operations to stay out.
#ifndef ETF_DMA
#define ETF_DMA 1
#endif
struct EtfDma {
  unsigned off[5];  // byte offset of this lane's 16 B within the [128][ETF_H] window of a weight tile, per instruction of this wave
  unsigned ws;      // LDS byte address of this wave's first 1 KB in slot 0 (wave-uniform, an SGPR)
  // NO vector instruction per request: the multiplier wave of the SIMD issues matrix instructions back to back and a vector-ALU
  // instruction of the mover gets a turn about once per MFMA (measured with s_memtime: 1.2 k cycles for five requests while each had a
  // v_readfirstlane and a 64-bit v_add in front of it - the movers were late at every step barrier).  M0 comes out of scalar arithmetic
  // and the address is SGPR base (tile) + 32-bit VGPR offset (lane), both set up once.
  __device__ __forceinline__ void init(int mw, int lane, const float* Ws) {
    ws = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)Ws) + (unsigned)mw * 1024u;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      int i = mw + 4 * j;
      if (i >= 18) i -= 4;
      const int u = 64 * i + lane, row = u / 9, c = u % 9;
      off[j] = (ETF_ABL & 16) ? (unsigned)(row * 32 + 4 * (c < 8 ? c : 7)) * 4u  // (timing only: the tile as 16 KB of consecutive memory)
                              : (unsigned)(row * ETF_H + 4 * (c < 8 ? c : 7)) * 4u;
    }
  }
  __device__ __forceinline__ void tile(unsigned long src, int slot, int mw) const {
    const unsigned long sb = ((unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)(src >> 32)) << 32) |
                             (unsigned long)(unsigned)__builtin_amdgcn_readfirstlane((unsigned)src);  // (the builtin returns int: no sign extension)
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      // instruction i = mw + 4 j of the slot (18, 19 repeat 14, 15): mw < 2 -> j * 4 KB, else the last one stays at j = 3
      const unsigned m0v = ws + slot * ETF_WS + ((j == 4 && mw >= 2) ? 3u : (unsigned)j) * 4096u;
      if (!(ETF_ABL & 4)) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" : : "s"(m0v), "v"(off[j]), "s"(sb) : "memory", "m0");
    }
  }
};
#define ETF_VMWAIT(n)                                                                                 \
  do {                                                                                                \
    if (!(ETF_ABL & 32) || (n) == 0) asm volatile("s_waitcnt vmcnt(" #n ")" : : : "memory");          \
  } while (0)  // (ETF_ABL & 32: timing only, nobody waits for the tiles)
// mover's side of a layer, DMA form: t0 a multiple of 6; `first(k)` runs behind the DMAs of the layer's first step (k = 0) and decides
// whether 12 more operations stay in front of the awaited DMAs for two steps
template <int NP>
__device__ __forceinline__ void etfs_layer_dma(const EtfStream& st, int t0, const EtfDma& D, int mw) {
  __syncthreads();
#ifdef ETF_PROF2
  unsigned long long mv_busy = 0, mv_issue = 0;
#endif
  auto step = [&](int t, int k) {
    if (!(ETF_ABL & 1)) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
#ifdef ETF_PROF2
    const unsigned long long m0_ = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    D.tile((ETF_ABL & 16) ? (unsigned long)st.w1 + (unsigned long)((t + 3) % ETF_TILES) * 16384ul : st.addr(t + 3), k % 3, mw);
#ifdef ETF_PROF2
    __builtin_amdgcn_sched_barrier(0);
    const unsigned long long m1_ = __builtin_amdgcn_s_memtime();
    __builtin_amdgcn_sched_barrier(0);
#endif
    ETF_VMWAIT(5);
    __builtin_amdgcn_sched_barrier(0);
#ifdef ETF_PROF2
    const unsigned long long m2_ = __builtin_amdgcn_s_memtime();
    mv_issue += m1_ - m0_; mv_busy += m2_ - m0_;
    __builtin_amdgcn_sched_barrier(0);
#endif
  };
#pragma unroll 1
  for (int tl = 0; tl < NP * 12; tl += 6) {
    step(t0 + tl, 0); step(t0 + tl + 1, 1); step(t0 + tl + 2, 2); step(t0 + tl + 3, 3); step(t0 + tl + 4, 4); step(t0 + tl + 5, 5);
  }
#ifdef ETF_PROF2
  if (threadIdx.x == FD_THREADS) { atomicAdd(&etf_prof2[3], mv_busy); atomicAdd(&etf_prof2[4], mv_issue); atomicAdd(&etf_prof2[5], (unsigned long long)(NP * 12)); }
#endif
}
// the final layer's twelve steps, straight-line (a request inside a rolled loop makes hipcc protect its destination registers against
// the previous iteration's loads with waits that know nothing of the DMAs in the queue); `request()` = the X0 rows of the next row tile
// (12 loads) or nothing, behind the DMAs of the first step
template <class Request>
__device__ __forceinline__ void etfs_final_dma(const EtfStream& st, const EtfDma& D, int mw, bool more, Request request) {
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 12; ++k) {
    if (!(ETF_ABL & 1)) __syncthreads();
    __builtin_amdgcn_sched_barrier(0);
    D.tile((ETF_ABL & 16) ? (unsigned long)st.w1 + (unsigned long)((72 + k + 3) % ETF_TILES) * 16384ul : st.addr(72 + k + 3), k % 3, mw);
    if (k == 0 && more) request();
    __builtin_amdgcn_sched_barrier(0);
    if (k < 2 && more) ETF_VMWAIT(17);
    else ETF_VMWAIT(5);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <class ZT>
__global__ __launch_bounds__(2 * FD_THREADS) void edge_transition_f32ws_kernel(EdgeTransArgs a, int n_blocks) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  FD_CLK_BEGIN;
  float* buf0 = (float*)smem;
  float* buf1 = (float*)(smem + ETF_BUF);
  float* Ws = (float*)(smem + 2 * ETF_BUF);
  float* ybuf = buf1;
  const int N = a.N;
  const long n_pairs = (long)a.B * N * N;
  const bool mover = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) >= 4;
  if (mover) {
    // ================= movers: weight stream and the X0 rows of the next row tile
    const int mt = threadIdx.x - FD_THREADS;
    const ZT* z_in = (const ZT*)a.z_in;
    const EtfStream st = {(const float*)a.w1, (const float*)a.w2, (const float*)a.wf};
#if ETF_DMA
    static_assert(sizeof(ZT) == 4, "etfs_final_dma counts the 12 X0 requests of an fp32 z (16 B per request) in its vmcnt waits");
    EtfDma D;
    const int mw = __builtin_amdgcn_readfirstlane(mt >> 6);
#ifndef ETF_MOVER_PRIO
#define ETF_MOVER_PRIO 0  // 3 (the movers' instructions first, as the non-matrix phases of edge_embed_f32p_kernel): 3.780 against 3.765 ms - with no
#endif                    // vector instruction left in a step there is nothing to let through
    if (ETF_MOVER_PRIO) __builtin_amdgcn_s_setprio(ETF_MOVER_PRIO);
    D.init(mw, mt & 63, Ws);
    D.tile(st.addr(0), 0, mw);
    D.tile(st.addr(1), 1, mw);
    D.tile(st.addr(2), 2, mw);
#else
    EtfTile g0, g1, g2;
    st.load(g0, 0, mt);
    st.load(g1, 1, mt);
    st.load(g2, 2, mt);
#endif
    f32x4 xr[12];
    // X0 rows of a row tile: thread -> 16 B column c of the rows mt / 32 + 8 k.  ONE pair of 32-bit divisions per thread (B N^2 < 2^32), the
    // three other rows by stepping (i, j) eight pairs on: next to a multiplier wave every vector instruction of a mover waits for a turn
    // (see EtfDma), and the 64-bit divisions of rounds 2 - 5 (two per row) held the first steps of the final layer up by ~190 cycles each.
    auto request_x0 = [&](long p0) {
      const unsigned uN = (unsigned)N, np = (unsigned)n_pairs, pr0 = (unsigned)p0 + (unsigned)(mt >> 5);
      unsigned bi = pr0 / uN, jj = pr0 - bi * uN, bb = bi / uN, ii = bi - bb * uN;
      const int c = (mt & 31) * 4;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        // (rows beyond the last pair carry the last pair's values: they are never stored, and a select on the loaded value here would
        //  make the compiler wait for the requests on the spot)
        const bool in = pr0 + 8u * k < np;
        const unsigned p_ = in ? pr0 + 8u * k : np - 1, bi_ = in ? bi : (unsigned)a.B * uN - 1, bj_ = in ? bb * uN + jj : (unsigned)a.B * uN - 1;
        if constexpr (sizeof(ZT) == 4) xr[k] = *(const f32x4*)((const float*)z_in + (long)p_ * ETF_CZ + c);
        else
#pragma unroll
          for (int q = 0; q < 4; ++q) xr[k][q] = z_load<ZT>(z_in + (long)p_ * ETF_CZ + c + q);
        xr[4 + k] = *(const f32x4*)(a.e + (long)bi_ * ETF_CZ + c);
        xr[8 + k] = *(const f32x4*)(a.e + (long)bj_ * ETF_CZ + c);
        jj += 8;
        while (jj >= uN) {
          jj -= uN; ++bi; ++ii;
          if (ii >= uN) { ii = 0; ++bb; }
        }
      }
    };
    auto store_x0 = [&]() {
#pragma unroll
      for (int u = 0; u < 12; ++u) {
        const int part = u >> 2, v = mt + (u & 3) * FD_THREADS, mm = v >> 5, c = (v & 31) * 4;
        *(f32x4*)(buf0 + mm * ETF_LDA + part * ETF_CZ + c) = xr[u];
      }
    };
    int blk = blockIdx.x;
    request_x0((long)blk * 32);
    store_x0();
#if ETF_DMA
    ETF_VMWAIT(0);  // tiles 0 .. 2 are in their slots
    for (; blk < n_blocks; blk += gridDim.x) {
      etfs_layer_dma<3>(st, 0, D, mw);
      etfs_layer_dma<3>(st, 36, D, mw);
      const long pn = (long)(blk + gridDim.x) * 32;
      etfs_final_dma(st, D, mw, blk + (int)gridDim.x < n_blocks, [&]() { request_x0(pn); });
      __syncthreads();
      store_x0();  // buf0 is free (the next layer 1 starts with a barrier)
    }
    ETF_VMWAIT(0);  // the tiles requested beyond the last row tile land before the block gives its LDS back
#else
    g0.store(Ws, mt);
    g1.store(Ws + ETF_WS / 4, mt);
    st.load(g0, 3, mt);
    f32x4 g_dummy[4] = {};  // (ETF_ABL & 8 only)
    for (; blk < n_blocks; blk += gridDim.x) {
      etfs_layer_move<3>(st, 0, Ws, g0, g1, g2, mt, g_dummy);
      etfs_layer_move<3>(st, 36, Ws, g0, g1, g2, mt, g_dummy);
      if (blk + (int)gridDim.x < n_blocks) request_x0((long)(blk + gridDim.x) * 32);
      etfs_layer_move<1>(st, 72, Ws, g0, g1, g2, mt, g_dummy);
      __syncthreads();
      store_x0();  // buf0 is free (the next layer 1 starts with a barrier)
    }
    if (ETF_ABL & 8) asm volatile("s_waitcnt vmcnt(0)" : : "v"(g_dummy[0]), "v"(g_dummy[1]), "v"(g_dummy[2]), "v"(g_dummy[3]) : "memory");
#endif
  } else {
    // ================= multipliers
    const int tid = threadIdx.x, lane = tid & 63, wc = tid >> 6;
    const int ncol = wc * 32 + (lane & 31);
    const EtfStream st = {(const float*)a.w1, (const float*)a.w2, (const float*)a.wf};
    unsigned tok = 0;  // destination of the L2 touches (etfs_layer_compute)
    float em_next = 0.f;
    auto request_em = [&](long p0) {  // lanes 0..31 of every wave: pair mask of row `lane`
      em_next = 0.f;
      if (lane < 32) {
        const long p = p0 + lane;
        if (p < n_pairs) {
          const long bi = p / N;
          em_next = a.res_mask[bi] * a.res_mask[(bi / N) * N + (p - bi * N)];
        }
      }
    };
    int blk = blockIdx.x;
    request_em((long)blk * 32);
    float bias1[3], bias2[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) { bias1[q] = a.b1[q * 128 + ncol]; bias2[q] = a.b2[q * 128 + ncol]; }
    const float biasf = a.bf[ncol];
    EtfLnConst lnc;
    lnc.load(a.gamma, a.beta, lane);
    for (; blk < n_blocks; blk += gridDim.x) {
      const long p0 = (long)blk * 32;
      const float em_row = em_next;
#ifdef ETF_PROF2
      const unsigned long long tile_t0 = __builtin_amdgcn_s_memtime();
#endif
      etfs_layer_compute<3>(buf0, st, 0, Ws, tid, tok, [&](int pass, const f32x16& acc) {
        const int n = pass * 128 + ncol;
        const float bv = pass == 0 ? bias1[0] : (pass == 1 ? bias1[1] : bias1[2]);
#pragma unroll
        for (int r = 0; r < 16; ++r) buf1[c_row(r, lane) * ETF_LDA + n] = fmaxf(acc[r] + bv, 0.f);
      });
      etfs_layer_compute<3>(buf1, st, 36, Ws, tid, tok, [&](int pass, const f32x16& acc) {
        const int n = pass * 128 + ncol;
        const float bv = pass == 0 ? bias2[0] : (pass == 1 ? bias2[1] : bias2[2]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* d = buf0 + c_row(r, lane) * ETF_LDA + n;
          *d = fmaxf(acc[r] + bv, 0.f) + *d;
        }
      });
      if (blk + (int)gridDim.x < n_blocks) request_em((long)(blk + gridDim.x) * 32);
      etfs_layer_compute<1>(buf0, st, 72, Ws, tid, tok, [&](int, const f32x16& acc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) ybuf[c_row(r, lane) * ETF_LDY + ncol] = acc[r] + biasf;
      });
      __syncthreads();
      if (!(ETF_ABL & 64))  // (ETF_ABL & 64: timing only, no LayerNorm and no stores)
        etf_ln_store<ZT>(ybuf, lnc, em_row, p0, n_pairs, (ZT*)a.z_out, a.trace, wc, lane);
#ifdef ETF_PROF2
      if (tid == 0) { atomicAdd(&etf_prof2[6], __builtin_amdgcn_s_memtime() - tile_t0); atomicAdd(&etf_prof2[7], 1ull); }
#endif
    }
    asm volatile("s_waitcnt vmcnt(0)" : : "v"(tok) : "memory");  // the last touches land before the register is released
  }
  FD_CLK_END(a.clock);
}

template <class P, class WT, class ZT, int TM, int WR, int WC, int CZ>
__global__ __launch_bounds__(FD_THREADS) void edge_embed_kernel(EdgeEmbedArgs a) {
  constexpr int LDA = CZ + P::PAD;
  constexpr int LDT = P::BK + P::PAD;
  constexpr int TN = 32 * WC;
  constexpr int LDY = CZ + 4;
  constexpr size_t BUF = ((size_t)TM * LDA * sizeof(typename P::T) + 15) / 16 * 16;
  constexpr size_t WS = ((size_t)TN * LDT * sizeof(typename P::T) + 15) / 16 * 16;
  __shared__ __attribute__((aligned(16))) char smem[2 * BUF + WS + (size_t)TM * LDY * 4 + TM * 16];
  typename P::T* buf0 = (typename P::T*)smem;
  typename P::T* buf1 = (typename P::T*)(smem + BUF);
  typename P::T* Ws = (typename P::T*)(smem + 2 * BUF);
  float* ybuf = (float*)(smem + 2 * BUF + WS);
  int* rowinfo = (int*)(smem + 2 * BUF + WS + (size_t)TM * LDY * 4);  // [TM][4]: bi, bj, rel, bin

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave / WC, wc = wave % WC;
  const int N = a.N;
  const long n_pairs = (long)a.B * N * N;
  const long p0 = (long)blockIdx.x * TM;

  if (tid < TM) {
    const long p = p0 + tid;
    int bi = -1, bj = 0, rel = 0, bin = a.num_bins;
    if (p < n_pairs) {
      const long lbi = p / N;
      const int j = (int)(p - lbi * N);
      const long bb = lbi / N;
      bi = (int)lbi;
      bj = (int)(bb * N + j);
      rel = (int)(bb * a.n_rel) + a.seq_idx[bi] - a.seq_idx[bj] + a.rel_off;
      const float dx = a.sc_ca[bi * 3 + 0] - a.sc_ca[bj * 3 + 0];
      const float dy = a.sc_ca[bi * 3 + 1] - a.sc_ca[bj * 3 + 1];
      const float dz = a.sc_ca[bi * 3 + 2] - a.sc_ca[bj * 3 + 2];
      const float d = sqrtf(dx * dx + dy * dy + dz * dz);
      // calc_distogram (framedipt/data/utils.py:541-550): strict inequalities, last upper edge 1e8
      for (int k = 0; k < a.num_bins; ++k) {
        const float lo = a.edges[k], up = (k + 1 < a.num_bins) ? a.edges[k + 1] : 1e8f;
        if (d > lo && d < up) bin = k;
      }
    }
    rowinfo[tid * 4 + 0] = bi; rowinfo[tid * 4 + 1] = bj; rowinfo[tid * 4 + 2] = rel; rowinfo[tid * 4 + 3] = bin;
  }
  __syncthreads();
  // ---- layer 1 without a GEMM: h1 = relu(Pi[i] + Pj[j] + R[rel] + D[bin])
  for (int v = tid; v < TM * (CZ / 4); v += FD_THREADS) {
    const int m = v / (CZ / 4), c = (v % (CZ / 4)) * 4;
    const int bi = rowinfo[m * 4 + 0];
    float x[4] = {0.f, 0.f, 0.f, 0.f};
    if (bi >= 0) {
      const f32x4 p1 = *(const f32x4*)(a.pi + (long)bi * CZ + c);
      const f32x4 p2 = *(const f32x4*)(a.pj + (long)rowinfo[m * 4 + 1] * CZ + c);
      const f32x4 p3 = *(const f32x4*)(a.rtab + (long)rowinfo[m * 4 + 2] * CZ + c);
      const f32x4 p4 = *(const f32x4*)(a.dtab + (long)rowinfo[m * 4 + 3] * CZ + c);
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = fmaxf(p1[q] + p2[q] + p3[q] + p4[q], 0.f);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) buf0[m * LDA + c + q] = P::from_f32(x[q]);
  }
  __syncthreads();
  f32x16 acc;
  const int ncol = wc * 32 + (lane & 31);
  for (int n0 = 0; n0 < CZ; n0 += TN) {
    act_gemm<P, WT, WR, WC>(acc, buf0, LDA, CZ, (const WT*)a.w2, CZ, n0, CZ, Ws, tid);
    const int n = n0 + ncol;
    if (n < CZ) {
      const float bv = a.b2[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) buf1[(wr * 32 + c_row(r, lane)) * LDA + n] = P::from_f32(fmaxf(acc[r] + bv, 0.f));
    }
  }
  __syncthreads();
  for (int n0 = 0; n0 < CZ; n0 += TN) {
    act_gemm<P, WT, WR, WC>(acc, buf1, LDA, CZ, (const WT*)a.w3, CZ, n0, CZ, Ws, tid);
    const int n = n0 + ncol;
    if (n < CZ) {
      const float bv = a.b3[n];
#pragma unroll
      for (int r = 0; r < 16; ++r) ybuf[(wr * 32 + c_row(r, lane)) * LDY + n] = acc[r] + bv;
    }
  }
  __syncthreads();
  ln_store<ZT, TM, CZ>(ybuf, a.gamma, a.beta, a.res_mask, p0, n_pairs, N, (ZT*)a.z_out, a.trace, tid);
}

// edge_embed_f32p_kernel — the fp32 pair embedder (score_network.py:98-105,173-196) for c_z = 128 as a PERSISTENT block with both
// 128 x 128 weight matrices RESIDENT in LDS: edge_embed_kernel<PrecF32> above re-streams the 128 KB of weights through LDS for every
// 32 pairs (125 k blocks at N = 1000, B = 4) with four barriers per 32-column k-tile and reaches a third of the fp32 matrix peak.
//
// Here a block has EIGHT waves = two TEAMS of four (one wave of each team per SIMD); a team walks 32-pair tiles through six phases
//   P0 gather h1 = relu(Pi[i] + Pj[j] + R[rel] + D[bin]) into its activation tile     P1 layer 2: 64 v_mfma_f32_32x32x2_f32 per wave
//   P2 h2 = relu(. + b2) back into the tile       P3 layer 3: 64 MFMAs       P4 y = . + b3 into the tile       P5 LayerNorm, mask, stores
// separated by block barriers, and the second team runs THREE PHASES (half a tile) ahead of the first: each matrix phase of one team
// (4.1 k cycles of the SIMD's matrix pipe) then faces a gather / epilogue / LayerNorm phase of the other (one team alone — one wave per
// SIMD — exposes every non-matrix phase: 697 us at N = 300, B = 8, 0.45 of the fp32 matrix peak).
// LDS is exactly 160 KB: W2 | W3 | two 32 x 128 fp32 activation tiles, all as 512 B rows whose 16 B chunk c sits at c ^ (row & 15)
// (conflict-free b128 operand reads, b128 gather stores, b32 epilogue stores and LayerNorm reads); the distogram edges live in one
// register per lane (lane k = edge k, fetched with a lane shuffle), the pair masks in the gathering lanes.  The table rows of a team's
// NEXT tile and the indices of the one after it travel under its current tile.  Same formulas and dtype flow as the reference (fp32
// operands, fp32 accumulation, two-pass LayerNorm statistics); only the summation order over k differs from the tiled kernel
// (k in pairs (i, i + 4) of every 8-group).
#define EEP_LDS (2 * 128 * 128 * 4 + 2 * 32 * 128 * 4)
#ifdef EEP_PROF  // phase profile (tools/micro/eep_bench.hip -DEEP_PROF): cycles of wave 0 / wave 4 of block 0 per phase, and in the barriers
__device__ unsigned long long eep_prof[2][8];
#endif
template <int N_, class F>
__device__ __forceinline__ void eep_for(F&& f) {  // f(integral_constant<0>) ... f(integral_constant<N_ - 1>)
  if constexpr (N_ > 0) {
    eep_for<N_ - 1>(f);
    f(std::integral_constant<int, N_ - 1>{});
  }
}
__device__ __forceinline__ int eep_off(int row, int col) {  // float offset of element (row, col) of a swizzled [rows][128] fp32 tile
  return row * 128 + ((((col >> 2) ^ (row & 15)) << 2) | (col & 3));
}
__global__ __launch_bounds__(2 * FD_THREADS, 1) void edge_embed_f32p_kernel(EdgeEmbedArgs a, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* W2s = (float*)smem;
  float* W3s = W2s + 128 * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, team = wave >> 2, wc = wave & 3, hi = lane >> 5, li = lane & 31;
  float* act = W3s + 128 * 128 + team * (32 * 128);
  const int N = a.N, nb = a.num_bins;
  const long n_pairs = (long)a.B * N * N;
  {  // weights -> LDS once ([out][in] rows, swizzled chunks)
    const float* w2 = (const float*)a.w2;
    const float* w3 = (const float*)a.w3;
#pragma unroll 4
    for (int k = 0; k < 8; ++k) {
      const int v = tid + k * 2 * FD_THREADS, r = v >> 5, c = (v & 31) * 4;
      *(f32x4*)(W2s + eep_off(r, c)) = *(const f32x4*)(w2 + r * 128 + c);
      *(f32x4*)(W3s + eep_off(r, c)) = *(const f32x4*)(w3 + r * 128 + c);
    }
  }
  const float edge_reg = lane < nb ? a.edges[lane] : 1e8f;  // lane k holds lower edge k (num_bins <= 64); beyond the last one: 1e8
  const float e0 = __shfl(edge_reg, 0, 64), inv_step = 1.0f / (__shfl(edge_reg, 1, 64) - e0);
  const int ncol = wc * 32 + li;
  const float b2v = a.b2[ncol], b3v = a.b3[ncol];
  const f32x4 gq0 = *(const f32x4*)(a.gamma + 4 * (lane & 15)), gq1 = *(const f32x4*)(a.gamma + 64 + 4 * (lane & 15));
  const f32x4 bq0 = *(const f32x4*)(a.beta + 4 * (lane & 15)), bq1 = *(const f32x4*)(a.beta + 64 + 4 * (lane & 15));
  // The per-lane constants above have ARRIVED before the first tile request is issued: the vmcnt counter is in order, and a use of one of
  // them inside the loop would otherwise be guarded by a wait (hipcc merges the first-iteration state into the loop) that also waits for
  // the table rows requested a moment earlier.
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_sched_barrier(0);
  // gather role of a thread: pair m of the team's tile (8 lanes per pair), 16 B chunks (lane & 7) + 8 q of its 512 B rows
  const int m = wc * 8 + (lane >> 3), c0 = (lane & 7) * 4;
  // (32-bit index arithmetic: the launcher refuses B N N >= 2^31; a 64-bit division is a subroutine with branches on this target)
  struct Idx { int si, sj; float mi, mj, ci[3], cj[3]; unsigned bi, bj, bb; };
  const unsigned np_u = (unsigned)n_pairs, Nu = (unsigned)N;
  auto idx_request = [&](int tile, Idx& x) {
    const unsigned pr = (unsigned)tile * 32u + (unsigned)m, p = pr < np_u ? pr : np_u - 1u;
    x.bi = p / Nu;
    const unsigned j = p - x.bi * Nu;
    x.bb = x.bi / Nu;
    x.bj = x.bb * Nu + j;
    x.si = a.seq_idx[x.bi]; x.sj = a.seq_idx[x.bj];
    x.mi = a.res_mask[x.bi]; x.mj = a.res_mask[x.bj];
#pragma unroll
    for (int c = 0; c < 3; ++c) { x.ci[c] = a.sc_ca[x.bi * 3u + c]; x.cj[c] = a.sc_ca[x.bj * 3u + c]; }
  };
  f32x4 tr[4][4];  // [table][chunk] rows of the team's next tile
  float em_next = 0.f, em_cur = 0.f;
  auto rows_request = [&](const Idx& x) {
    const int rel = (int)(x.bb * a.n_rel) + x.si - x.sj + a.rel_off;
    const float dx = x.ci[0] - x.cj[0], dy = x.ci[1] - x.cj[1], dz = x.ci[2] - x.cj[2];
    const float d = sqrtf(dx * dx + dy * dy + dz * dz);
    // calc_distogram (framedipt/data/utils.py:541-550): strict inequalities against the stored edges, last upper edge 1e8.  The
    // candidate comes from the edge spacing (the edges are a linspace), its neighbours are re-tested, so the result is the one a
    // full scan finds (the same search as edge_embed2_kernel's)
    int k0 = (int)((d - e0) * inv_step);
    k0 = k0 < 1 ? 1 : (k0 > nb - 2 ? nb - 2 : k0);
    const float ea = __shfl(edge_reg, k0 - 1, 64), eb = __shfl(edge_reg, k0, 64), ec = __shfl(edge_reg, k0 + 1, 64);
    const float ed_ = __shfl(edge_reg, (k0 + 2) & 63, 64), ed = k0 + 2 < nb ? ed_ : 1e8f;
    int bin = nb;
    bin = (d > ea && d < eb) ? k0 - 1 : bin;
    bin = (d > eb && d < ec) ? k0 : bin;
    bin = (d > ec && d < ed) ? k0 + 1 : bin;
    em_next = x.mi * x.mj;
    const float* s0 = a.pi + (long)x.bi * 128 + c0;
    const float* s1 = a.pj + (long)x.bj * 128 + c0;
    const float* s2 = a.rtab + (long)rel * 128 + c0;
    const float* s3 = a.dtab + (long)bin * 128 + c0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      tr[0][q] = *(const f32x4*)(s0 + 32 * q);
      tr[1][q] = *(const f32x4*)(s1 + 32 * q);
      tr[2][q] = *(const f32x4*)(s2 + 32 * q);
      tr[3][q] = *(const f32x4*)(s3 + 32 * q);
    }
  };
  // the team's tiles: blockIdx.x + (2 i + team) gridDim.x, i = 0, 1, ...
  const int stride = 2 * (int)gridDim.x;
  const int tile0 = (int)blockIdx.x + team * (int)gridDim.x;
  const int n_mine = tile0 < n_tiles ? (n_tiles - tile0 + stride - 1) / stride : 0;
  const int n_first = (int)blockIdx.x < n_tiles ? (n_tiles - (int)blockIdx.x + stride - 1) / stride : 0;  // team 0's count >= team 1's
  Idx ix;
  if (n_mine > 0) {
    idx_request(tile0, ix);
    rows_request(ix);
    idx_request(tile0 + (n_mine > 1 ? stride : 0), ix);
  }
  const int sw = li & 15;
  const float* arow = act + li * 128;
  const float* w2row = W2s + ncol * 128;
  const float* w3row = W3s + ncol * 128;
  const int swn = ncol & 15;
  auto layer = [&](const float* wrow) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // operand chunks in two halves of 8 (64 registers in flight: the kernel keeps 64 more for the next tile's table rows)
#pragma unroll 1
    for (int half = 0; half < 2; ++half) {
      f32x4 av[8], wv[8];
#pragma unroll
      for (int k8 = 0; k8 < 8; ++k8) {
        const int kk = 8 * half + k8;
        av[k8] = *(const f32x4*)(arow + (((2 * kk + hi) ^ sw) << 2));
        wv[k8] = *(const f32x4*)(wrow + (((2 * kk + hi) ^ swn) << 2));
      }
#pragma unroll
      for (int k8 = 0; k8 < 8; ++k8)
#pragma unroll
        for (int i = 0; i < 4; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[k8][i], wv[k8][i], acc, 0, 0, 0);
    }
    return acc;
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  int done = 0;       // tiles of this team whose P0 has run
  long p0 = 0;        // first pair of the tile in flight
  auto phase = [&](auto PH) {
    constexpr int ph = decltype(PH)::value;
    // issue priority: a wave in a matrix phase re-issues its next (dependent) MFMA the moment the pipe frees and, being served first,
    // lets the SIMD's other wave issue about one instruction per MFMA (measured: the 32-instruction y store 0.5 k cycles beside a
    // LayerNorm, 2.4 k beside a matrix phase); with the non-matrix phases at a higher priority they slip into the MFMA shadows
    __builtin_amdgcn_s_setprio(ph == 1 || ph == 3 ? 0 : 3);
    if constexpr (ph == 0) {  // gather: h1 from the rows requested one tile ago; request the next tile's rows / the indices after it
      if (done < n_mine) {
        p0 = (long)(tile0 + done * stride) * 32;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4 h;
#pragma unroll
          for (int e = 0; e < 4; ++e) h[e] = fmaxf(tr[0][q][e] + tr[1][q][e] + tr[2][q][e] + tr[3][q][e], 0.f);
          *(f32x4*)(act + m * 128 + ((((lane & 7) + 8 * q) ^ (m & 15)) << 2)) = h;
        }
        em_cur = em_next;
        // Unconditional (tile indices clamped to the team's last tile: a few redundant requests at the very end): a request under a
        // branch makes the loaded registers phi values of the join, and hipcc then copies them there — behind an s_waitcnt vmcnt that
        // exposes the whole round trip (measured: this phase 2.1 k -> 5.6 k cycles).  The fence keeps the products of the old indices
        // (em_next, rel, bin) in front of the new index loads, so those land in the old registers.
        rows_request(ix);
        __builtin_amdgcn_sched_barrier(0);
        idx_request(tile0 + (done + 2 < n_mine ? done + 2 : n_mine - 1) * stride, ix);
      }
    } else if constexpr (ph == 1) {
      if (done < n_mine) acc = layer(w2row);
    } else if constexpr (ph == 2) {
      if (done < n_mine)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[eep_off(c_row(r, lane), ncol)] = fmaxf(acc[r] + b2v, 0.f);
    } else if constexpr (ph == 3) {
      if (done < n_mine) acc = layer(w3row);
    } else if constexpr (ph == 4) {
      if (done < n_mine)
#pragma unroll
        for (int r = 0; r < 16; ++r) act[eep_off(c_row(r, lane), ncol)] = acc[r] + b3v;
    } else {
      if (done < n_mine) {  // LayerNorm (two-pass statistics as torch's) + pair mask; wave wc owns the rows it gathered: 8 wc .. 8 wc + 7
        // four rows per pass, 16 lanes per row: lane (rp, fl) holds features 4 fl .. + 3 and 64 + 4 fl .. + 3 (two conflict-free b128
        // reads, two 256 B row segments per store instruction); the statistics need four butterfly steps inside a 16-lane row
        float* z_out = (float*)a.z_out;
        const int rp = lane >> 4, fl = lane & 15;
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int q = 4 * pass + rp, r = 8 * wc + q;
          const f32x4 x0 = *(const f32x4*)(act + r * 128 + ((fl ^ (r & 15)) << 2));
          const f32x4 x1 = *(const f32x4*)(act + r * 128 + (((fl + 16) ^ (r & 15)) << 2));
          float s1 = ((x0[0] + x0[1]) + (x0[2] + x0[3])) + ((x1[0] + x1[1]) + (x1[2] + x1[3]));
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) s1 += __shfl_xor(s1, o, 64);
          const float mu = s1 * (1.0f / 128);
          f32x4 d0, d1;
          float s2 = 0.f;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            d0[e] = x0[e] - mu;
            d1[e] = x1[e] - mu;
            s2 += d0[e] * d0[e] + d1[e] * d1[e];
          }
#pragma unroll
          for (int o = 8; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
          const float rstd = 1.0f / sqrtf(s2 * (1.0f / 128) + 1e-5f);
          const float em = __shfl(em_cur, 8 * q, 64);
          const long p = p0 + r;
          if (p < n_pairs) {
            f32x4 o0, o1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              o0[e] = (d0[e] * rstd * gq0[e] + bq0[e]) * em;
              o1[e] = (d1[e] * rstd * gq1[e] + bq1[e]) * em;
            }
            *(f32x4*)(z_out + p * 128 + 4 * fl) = o0;
            *(f32x4*)(z_out + p * 128 + 64 + 4 * fl) = o1;
            if (a.trace) { *(f32x4*)(a.trace + p * 128 + 4 * fl) = o0; *(f32x4*)(a.trace + p * 128 + 64 + 4 * fl) = o1; }
          }
        }
        ++done;
      }
    }
  };
  __syncthreads();  // weights in LDS
  // Barrier slots s = 0..5 of a round: team 0 runs phase s of its round-th tile, team 1 phase (s + 3) % 6 — the last three phases of
  // its previous tile in slots 0..2 (nothing in round 0), the first three of its next one in slots 3..5.  n_first + 1 rounds: the last
  // one only finishes team 1's last tile.  Every wave executes every barrier.
  bool started = team == 0;  // team 1 has no tile in flight during slots 0..2 of round 0
  for (int round = 0; round <= n_first; ++round) {
    eep_for<6>([&](auto S) {
      constexpr int s = decltype(S)::value;
#ifdef EEP_PROF
      const unsigned long long t0_ = __builtin_amdgcn_s_memtime();
#endif
      if (team == 0) {
        phase(std::integral_constant<int, s>{});
      } else {
        constexpr int ph = (s + 3) % 6;
        if (ph < 3) started = true;
        if (started) phase(std::integral_constant<int, ph>{});
      }
#ifdef EEP_PROF
      __builtin_amdgcn_sched_barrier(0);
      const unsigned long long t1_ = __builtin_amdgcn_s_memtime();
#endif
      __syncthreads();
#ifdef EEP_PROF
      if (blockIdx.x == 0 && (tid & 255) == 0) {
        eep_prof[team][team == 0 ? s : (s + 3) % 6] += t1_ - t0_;
        eep_prof[team][6] += __builtin_amdgcn_s_memtime() - t1_;
      }
#endif
    });
  }
}

// ------------------------------------------------------------------ host launchers (CZ/CB dispatch)
template <int CZ, int CB>
static int launch_et(int precision, const EdgeTransArgs& a, hipStream_t st) {
  const long n_pairs = (long)a.B * a.N * a.N;
  if (precision == FDIPT_PREC_F32) {
    constexpr int TM = 32;
    if constexpr (CZ == ETF_CZ && CB == ETF_CZ) {
#ifndef ETF_SPEC
#define ETF_SPEC 1  // 1: edge_transition_f32ws_kernel (8 waves: 4 multiply, 4 move), 0: the fused 4-wave kernel
#endif
      static FdPerDevice attr_dev;
      const int dev_ = fd_device();
      if (!attr_dev.get(dev_)) {
        if (hipFuncSetAttribute((const void*)edge_transition_f32_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, ETF_LDS) !=
                hipSuccess ||
            hipFuncSetAttribute((const void*)edge_transition_f32ws_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, ETF_LDS) !=
                hipSuccess)
          return FDIPT_ELAUNCH;
        attr_dev.set(dev_, 1);
      }
      const int n_blocks = (int)cdiv(n_pairs, TM);
      if (ETF_SPEC)
        hipLaunchKernelGGL(edge_transition_f32ws_kernel<float>, dim3(n_blocks < 256 ? n_blocks : 256), dim3(2 * FD_THREADS), ETF_LDS, st, a,
                           n_blocks);
      else
        hipLaunchKernelGGL(edge_transition_f32_kernel<float>, dim3(n_blocks < 256 ? n_blocks : 256), dim3(FD_THREADS), ETF_LDS, st, a,
                           n_blocks);
    } else {
      hipLaunchKernelGGL((edge_transition_kernel<PrecF32, float, float, TM, 1, 4, CZ, CB>), dim3(cdiv(n_pairs, TM)),
                         dim3(FD_THREADS), 0, st, a);
    }
  } else {
    constexpr int TM = 64;
    hipLaunchKernelGGL((edge_transition_kernel<PrecHalf, half_t, half_t, TM, 2, 2, CZ, CB>), dim3(cdiv(n_pairs, TM)),
                       dim3(FD_THREADS), 0, st, a);
  }
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

int fd_edge_transition(int precision, int cz, int cb, const EdgeTransArgs& a, hipStream_t st) {
  if (cz == 128 && cb == 128) return launch_et<128, 128>(precision, a, st);
  if (cz == 32 && cb == 32) return launch_et<32, 32>(precision, a, st);
  return FDIPT_EINVAL;
}

template <int CZ>
static int launch_ee(int precision, const EdgeEmbedArgs& a, hipStream_t st) {
  const long n_pairs = (long)a.B * a.N * a.N;
  if (precision == FDIPT_PREC_F32) {
    constexpr int TM = 32;
    if constexpr (CZ == 128) {
      if (a.num_bins >= 3 && a.num_bins <= 64 && !FD_DEV_ENV("FDIPT_EE_F32_TILED")) {  // persistent kernel, weights resident in LDS
        static FdPerDevice attr_dev;
        const int dev_ = fd_device();
        if (!attr_dev.get(dev_)) {
          if (hipFuncSetAttribute((const void*)edge_embed_f32p_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, EEP_LDS) != hipSuccess)
            return FDIPT_ELAUNCH;
          attr_dev.set(dev_, 1);
        }
        const int n_tiles = (int)cdiv(n_pairs, 32);
        hipLaunchKernelGGL(edge_embed_f32p_kernel, dim3(n_tiles < fd_cu_count() ? n_tiles : fd_cu_count()), dim3(2 * FD_THREADS), EEP_LDS, st, a, n_tiles);
        FD_CHECK_LAUNCH();
        return FDIPT_OK;
      }
    }
    hipLaunchKernelGGL((edge_embed_kernel<PrecF32, float, float, TM, 1, 4, CZ>), dim3(cdiv(n_pairs, TM)),
                       dim3(FD_THREADS), 0, st, a);
  } else {
    constexpr int TM = 64;
    hipLaunchKernelGGL((edge_embed_kernel<PrecHalf, half_t, half_t, TM, 2, 2, CZ>), dim3(cdiv(n_pairs, TM)),
                       dim3(FD_THREADS), 0, st, a);
  }
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

int fd_edge_embed(int precision, int cz, const EdgeEmbedArgs& a, hipStream_t st) {
  if (cz == 128) return launch_ee<128>(precision, a, st);
  if (cz == 32) return launch_ee<32>(precision, a, st);
  return FDIPT_EINVAL;
}
