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

// ------------------------------------------------------------------ host launchers (CZ/CB dispatch)
template <int CZ, int CB>
static int launch_et(int precision, const EdgeTransArgs& a, hipStream_t st) {
  const long n_pairs = (long)a.B * a.N * a.N;
  if (precision == FDIPT_PREC_F32) {
    constexpr int TM = 32;
    hipLaunchKernelGGL((edge_transition_kernel<PrecF32, float, float, TM, 1, 4, CZ, CB>), dim3(cdiv(n_pairs, TM)),
                       dim3(FD_THREADS), 0, st, a);
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
