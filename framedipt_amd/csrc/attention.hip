// attention.hip — invariant point attention core and the sequence-transformer attention.
//
//   attn_kernel<P, IPA=true>  : InvariantPointAttention.forward, framedipt/model/ipa_pytorch.py:251-313
//        logits = QK^T/sqrt(3C) + b/sqrt(3) - 0.5 gamma_h sum_p |q_pt - k_pt|^2 + 1e5 (m_i m_j - 1); softmax_j;
//        o = a v ; o_pt = R_i^T (a v_pts - t_i) ; |o_pt|.  The [N,N,H,P,3] displacement tensor is never built.
//   attn_kernel<P, IPA=false> : nn.MultiheadAttention inside nn.TransformerEncoderLayer, ipa_pytorch.py:433-443,536-538
//   opair_kernel              : o_pair = sum_j a[h,i,j] down_z(z[i,j]) (ipa_pytorch.py:317-322) evaluated as
//                               down_z(sum_j a z) + b * sum_j a, so pair_z [N,N,c_z/4] is never materialised.
//   points_kernel             : "split-thirds then stack" + Rigid.apply of the projected points (ipa_pytorch.py:213-239)
//
// One block = 32 query rows of one (batch, head); logits of the 32 rows stay in LDS as fp32 [32][N].
#include <stdlib.h>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"


// transposing stage: Ws[n][kk] = src[(k0+kk)*ld + n0 + n], kk < 32, n < TN (zero fill out of range)
template <class P, int TN>
__device__ __forceinline__ void stage_tile_T(typename P::T* dst, const float* __restrict__ src, long ld, int k0, int Kmax,
                                             int n0, int Nmax, int tid) {
  constexpr int LDT = P::BK + P::PAD;
  constexpr int VPR = TN / 4;
  for (int v = tid; v < P::BK * VPR; v += FD_THREADS) {
    const int r = v / VPR, c4 = (v % VPR) * 4;
    f32x4 x = {0.f, 0.f, 0.f, 0.f};
    if (k0 + r < Kmax && n0 + c4 < Nmax) x = *(const f32x4*)(src + (long)(k0 + r) * ld + n0 + c4);
#pragma unroll
    for (int qd = 0; qd < 4; ++qd) dst[(c4 + qd) * LDT + r] = P::from_f32(x[qd]);
  }
}

// (StageRegs / stage_load / stage_store: common.hpp)
// transposing: Ws[n][kk] = src[(k0 + kk) * ld + n0 + n], kk < BK, n < TN
template <class P, int TN>
__device__ __forceinline__ void stage_load_T(StageRegs<P::BK, TN>& R, const float* __restrict__ src, long ld, int k0, int Kmax, int n0, int Nmax, int tid) {
  constexpr int VPR = TN / 4, NV = (P::BK * VPR + FD_THREADS - 1) / FD_THREADS;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int v = tid + u * FD_THREADS, r = v / VPR, c4 = (v % VPR) * 4;
    R.v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (v < P::BK * VPR && k0 + r < Kmax && n0 + c4 < Nmax) R.v[u] = *(const f32x4*)(src + (long)(k0 + r) * ld + n0 + c4);
  }
}
template <class P, int TN>
__device__ __forceinline__ void stage_store_T(typename P::T* dst, const StageRegs<P::BK, TN>& R, int tid) {
  constexpr int LDT = P::BK + P::PAD, VPR = TN / 4, NV = (P::BK * VPR + FD_THREADS - 1) / FD_THREADS;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int v = tid + u * FD_THREADS, r = v / VPR, c4 = (v % VPR) * 4;
    if (v < P::BK * VPR)
#pragma unroll
      for (int qd = 0; qd < 4; ++qd) dst[(c4 + qd) * LDT + r] = P::from_f32(R.v[u][qd]);
  }
}

template <class P, bool IPA>
__global__ __launch_bounds__(FD_THREADS) void attn_kernel(AttnArgs a) {
  constexpr int LDT = P::BK + P::PAD;
  constexpr int TN = 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.N;
  const int LDS_S = a.lds_s;
  float* S = (float*)smem;                                   // [32][LDS_S]
  typename P::T* As = (typename P::T*)(smem + (size_t)32 * LDS_S * 4);   // [32][LDT]
  typename P::T* Ws = As + 32 * LDT;                          // [128][LDT]
  float* qps = (float*)(Ws + TN * LDT);                       // [32][Pq*3]   (IPA)
  float* vps = (float*)As;                                    // [64][Pv*3]   (IPA, aliases As/Ws after the S phase)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i0 = blockIdx.x * 32, h = blockIdx.y, b = blockIdx.z;
  const long rb = (long)b * N;
  const float* qbase = a.q + rb * a.q_ld + (long)h * a.q_hs;
  const float* kbase = a.k + rb * a.k_ld + (long)h * a.k_hs;
  const float* vbase = a.v + rb * a.v_ld + (long)h * a.v_hs;
  const int P3 = IPA ? a.Pq * 3 : 0;

  if constexpr (IPA) {
    for (int v = tid; v < 32 * P3; v += FD_THREADS) {
      const int r = v / P3, c = v % P3;
      qps[v] = (i0 + r < N) ? a.qp[((rb + i0 + r) * a.H + h) * P3 + c] : 0.f;
    }
  }
  // ---------------- phase 1: logits S[32][N]
  for (int j0 = 0; j0 < N; j0 += TN) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    StageRegs<32, P::BK> ra;
    StageRegs<TN, P::BK> rw;
    stage_load<P, 32>(ra, qbase, a.q_ld, i0, N, 0, a.C, tid);
    stage_load<P, TN>(rw, kbase, a.k_ld, j0, N, 0, a.C, tid);
    for (int k0 = 0; k0 < a.C; k0 += P::BK) {
      stage_store<P, 32>(As, ra, tid);
      stage_store<P, TN>(Ws, rw, tid);
      __syncthreads();
      if (k0 + P::BK < a.C) {  // the next k-tile's rows travel under this tile's products
        stage_load<P, 32>(ra, qbase, a.q_ld, i0, N, k0 + P::BK, a.C, tid);
        stage_load<P, TN>(rw, kbase, a.k_ld, j0, N, k0 + P::BK, a.C, tid);
      }
      wave_mma<P>(acc, As + (lane & 31) * LDT, Ws + (wave * 32 + (lane & 31)) * LDT, lane);
      __syncthreads();
    }
    const int j = j0 + wave * 32 + (lane & 31);
    if (j < N) {
      const float mj = a.res_mask[rb + j];
      float kpv[24];
      float gam = 0.f;
      if constexpr (IPA) {
        gam = -0.5f * a.gamma[h];
#pragma unroll
        for (int c = 0; c < 24; ++c) kpv[c] = c < P3 ? a.kp[((rb + j) * a.H + h) * P3 + c] : 0.f;
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ir = c_row(r, lane), i = i0 + ir;
        float s = acc[r] * a.scale;
        if (i < N) {
          if constexpr (IPA) {
            s += a.bias[((rb + i) * N + j) * a.H + h];
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < 24; ++c) {
              if (c < P3) {
                const float d = qps[ir * P3 + c] - kpv[c];
                d2 += d * d;
              }
            }
            s += gam * d2;
            s += 1e5f * (a.res_mask[rb + i] * mj - 1.f);
          } else {
            if (mj == 0.f) s = -1e30f;  // key padding (torch >= 2 inference fast-path semantics)
          }
        }
        S[ir * LDS_S + j] = s;
      }
    }
  }
  __syncthreads();
  // ---------------- phase 2: softmax over j (fp32), rows owned by waves
  for (int ir = wave; ir < 32; ir += FD_THREADS / 64) {
    float* row = S + ir * LDS_S;
    float v[16];
    float mx = -3.0e38f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int j = lane + t * 64;
      v[t] = j < N ? row[j] : -3.0e38f;
      mx = fmaxf(mx, v[t]);
    }
    mx = wave_max(mx);
    float sum = 0.f;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int j = lane + t * 64;
      v[t] = j < N ? expf(v[t] - mx) : 0.f;
      sum += v[t];
    }
    const float inv = 1.0f / wave_sum(sum);
    const int i = i0 + ir;
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int j = lane + t * 64;
      if (j < N) {
        const float p = v[t] * inv;
        row[j] = p;
        if (IPA && i < N) a.probs[(((long)b * a.H + h) * N + i) * N + j] = p;
      }
    }
  }
  __syncthreads();
  // ---------------- phase 3 (IPA): o_pt = R_i^T (sum_j p v_pts - t_i), fp32 VALU
  if constexpr (IPA) {
    const int Pv = a.Pv, V3 = Pv * 3;
    float ax[2] = {0.f, 0.f}, ay[2] = {0.f, 0.f}, az[2] = {0.f, 0.f};
    for (int j0 = 0; j0 < N; j0 += 64) {
      for (int v = tid; v < 64 * V3; v += FD_THREADS) {
        const int r = v / V3, c = v % V3;
        vps[v] = (j0 + r < N) ? a.vp[((rb + j0 + r) * a.H + h) * V3 + c] : 0.f;
      }
      __syncthreads();
      const int jn = min(64, N - j0);
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int it = tid + u * FD_THREADS;
        if (it < 32 * Pv) {
          const int ir = it / Pv, pt = it % Pv;
          const float* prow = S + ir * LDS_S + j0;
          float sx = 0.f, sy = 0.f, sz = 0.f;
          for (int jj = 0; jj < jn; ++jj) {
            const float p = prow[jj];
            sx += p * vps[jj * V3 + pt * 3 + 0];
            sy += p * vps[jj * V3 + pt * 3 + 1];
            sz += p * vps[jj * V3 + pt * 3 + 2];
          }
          ax[u] += sx; ay[u] += sy; az[u] += sz;
        }
      }
      __syncthreads();
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int it = tid + u * FD_THREADS;
      if (it < 32 * Pv) {
        const int ir = it / Pv, pt = it % Pv, i = i0 + ir;
        if (i < N) {
          const float* R = a.rot + (rb + i) * 9;
          const float* T = a.trans + (rb + i) * 3;
          const float x = ax[u] - T[0], y = ay[u] - T[1], z = az[u] - T[2];
          // invert_apply: R^T (p - t)   (openfold/utils/rigid_utils.py:1118-1130)
          const float ox = R[0] * x + R[3] * y + R[6] * z;
          const float oy = R[1] * x + R[4] * y + R[7] * z;
          const float oz = R[2] * x + R[5] * y + R[8] * z;
          float* o = a.out + (rb + i) * a.out_ld + a.pt_off + h * Pv + pt;
          const int HP = a.H * Pv;
          o[0] = ox; o[HP] = oy; o[2 * HP] = oz;
          o[3 * HP] = sqrtf(ox * ox + oy * oy + oz * oz + 1e-8f);
        }
      }
    }
  }
  // ---------------- phase 4: operand-precision copy of P in place (bf16 mode), then O = P V on MFMA
  if constexpr (sizeof(typename P::T) == 2) {
    for (int ir = wave; ir < 32; ir += FD_THREADS / 64) {
      float* row = S + ir * LDS_S;
      float v[16];
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int j = lane + t * 64;
        v[t] = j < N ? row[j] : 0.f;
      }
      typename P::T* prow = (typename P::T*)row;
      const int npad = (N + 31) & ~31;
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        const int j = lane + t * 64;
        if (j < npad) prow[j] = P::from_f32(v[t]);
      }
    }
  } else {
    const int npad = (N + 31) & ~31;
    for (int v = tid; v < 32 * (npad - N); v += FD_THREADS) S[(v / (npad - N)) * LDS_S + N + v % (npad - N)] = 0.f;
  }
  __syncthreads();
  const typename P::T* Pm = (const typename P::T*)S;
  const int ldp = LDS_S * (int)(4 / sizeof(typename P::T));
  for (int d0 = 0; d0 < a.Dv; d0 += TN) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    StageRegs<P::BK, TN> rv;
    stage_load_T<P, TN>(rv, vbase, a.v_ld, 0, N, d0, a.Dv, tid);
    for (int k0 = 0; k0 < N; k0 += P::BK) {
      stage_store_T<P, TN>(Ws, rv, tid);
      __syncthreads();
      if (k0 + P::BK < N) stage_load_T<P, TN>(rv, vbase, a.v_ld, k0 + P::BK, N, d0, a.Dv, tid);
      wave_mma<P>(acc, Pm + (lane & 31) * ldp + k0, Ws + (wave * 32 + (lane & 31)) * LDT, lane);
      __syncthreads();
    }
    const int d = d0 + wave * 32 + (lane & 31);
    if (d < a.Dv) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + c_row(r, lane);
        if (i < N) a.out[(rb + i) * a.out_ld + (long)h * a.Dv + d] = acc[r];
      }
    }
  }
}

template <class P>
static size_t attn_smem(int N, int Pq, int* lds_s) {
  const int npad = (N + 63) & ~63;
  *lds_s = npad + (sizeof(typename P::T) == 2 ? 4 : 1);
  constexpr int LDT = P::BK + P::PAD;
  return (size_t)32 * (*lds_s) * 4 + (size_t)(32 + 128) * LDT * sizeof(typename P::T) + (size_t)32 * Pq * 3 * 4 + 16;
}

template <class P, bool IPA>
static int launch_attn(AttnArgs a, hipStream_t st) {
  if (a.N > 1024 || a.Pq * 3 > 24 || a.Pv > 16 || (a.C & 3) || (a.Dv & 3)) return FDIPT_ESIZE;
  const size_t smem = attn_smem<P>(a.N, IPA ? a.Pq : 0, &a.lds_s);
  if (smem > 160 * 1024) return FDIPT_ESIZE;
  static FdPerDevice attr_dev;  // idempotent: raises the dynamic-LDS cap of this kernel instance once
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)attn_kernel<P, IPA>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) !=
        hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  hipLaunchKernelGGL((attn_kernel<P, IPA>), dim3(cdiv(a.N, 32), a.H, a.B), dim3(FD_THREADS), smem, st, a);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

int fd_attention(int precision, int ipa, const AttnArgs& a, hipStream_t st) {
  if (precision == FDIPT_PREC_F32) return ipa ? launch_attn<PrecF32, true>(a, st) : launch_attn<PrecF32, false>(a, st);
  return ipa ? launch_attn<PrecHalf, true>(a, st) : launch_attn<PrecHalf, false>(a, st);
}

// ------------------------------------------------------------------ o_pair

// One block per (b, i): streams the pair row z[b,i,:,:] ONCE with 16-byte loads (HBM-bound, N*CZ*sizeof(ZT) bytes),
// thread = (8-channel group, key slice); az[h][c] = sum_j a[h,i,j] z[i,j,c] in registers, slices folded by
// wave shuffles then across the 4 waves through LDS; then the (c_z -> c_z/4) down-projection per head.
// HBM-bound (one pass over z): the z rows of a thread are fetched in chunks of OP_CH rows, chunk c+1 in flight while
// chunk c is consumed, and the first chunk is issued before anything else; the down-projection weights sit in LDS.
#define OP_CH 8
template <class ZT, int CZ, int HT>
__global__ __launch_bounds__(FD_THREADS) void opair_kernel(OPairArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NCG = CZ / 8;              // channel groups of 8
  constexpr int NSL = FD_THREADS / NCG;    // key slices per block
  constexpr int EPT = sizeof(ZT) == 2 ? 1 : 2;  // 16-byte loads per row piece
  const int N = a.N, H = HT > 0 ? HT : a.H, CD = a.CD;
  float* ps = (float*)smem;                // [N][8]: attention weights of the 8 heads, key-major (vector reads per row)
  float* red = ps + 8 * N;                 // [4 waves][8 heads][CZ]
  float* psum = red + 4 * 8 * CZ;          // [8]
  float* wdl = psum + 8;                   // [CZ][CD] down-projection weights
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int i = blockIdx.x, b = blockIdx.y;
  const long rb = (long)b * N;
  const ZT* zrow = (const ZT*)a.z + (rb + i) * N * CZ;
  const int cg = tid % NCG, sl = tid / NCG;
  const int nk = (N - sl + NSL - 1) / NSL;  // rows of this thread: j = sl + k * NSL, k < nk
  f32x4 zr[2][OP_CH][EPT];
  auto fetch = [&](int buf, int k0) {
#pragma unroll
    for (int u = 0; u < OP_CH; ++u) {
      int k = k0 + u;
      if (k >= nk) k = nk > 0 ? nk - 1 : 0;  // harmless re-read, discarded below
      const ZT* zp = zrow + (long)(sl + k * NSL < N ? sl + k * NSL : 0) * CZ + cg * 8;
#pragma unroll
      for (int e = 0; e < EPT; ++e) zr[buf][u][e] = *(const f32x4*)((const char*)zp + 16 * e);
    }
  };
  FD_STAMP(0);
  // attention weights first (they gate the barrier), then the first z chunk, then the down-projection weights
  constexpr int NPV = 10;  // probs values per thread: 8 heads x N <= 2560
  float pr[NPV];
#pragma unroll
  for (int u = 0; u < NPV; ++u) {
    const int v = tid + u * FD_THREADS, hh = v / N, j = v % N;
    pr[u] = v < H * N ? a.probs[(((long)b * H + hh) * N + i) * N + j] : 0.f;
  }
  fetch(0, 0);
  if (H < 8) {
    for (int v = tid; v < 8 * N; v += FD_THREADS) ps[v] = 0.f;
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < NPV; ++u) {
    const int v = tid + u * FD_THREADS, hh = v / N, j = v % N;
    if (v < H * N) ps[j * 8 + hh] = pr[u];
  }
  for (int v = tid + NPV * FD_THREADS; v < H * N; v += FD_THREADS) {  // N > 320
    const int hh = v / N, j = v % N;
    ps[j * 8 + hh] = a.probs[(((long)b * H + hh) * N + i) * N + j];
  }
  for (int v = tid; v < CZ * CD; v += FD_THREADS) wdl[v] = a.wdz[v];
  __syncthreads();
  FD_STAMP(1);
  float acc[8][8];
#pragma unroll
  for (int hh = 0; hh < 8; ++hh)
#pragma unroll
    for (int c = 0; c < 8; ++c) acc[hh][c] = 0.f;
  auto consume = [&](auto BUF, int k0) {
    constexpr int bf = decltype(BUF)::value;
#pragma unroll
    for (int u = 0; u < OP_CH; ++u) {
      if (k0 + u < nk) {
        const int j = sl + (k0 + u) * NSL;
        float zv[8];
        if constexpr (sizeof(ZT) == 2) {
          const u16x8 raw = __builtin_bit_cast(u16x8, zr[bf][u][0]);
#pragma unroll
          for (int c = 0; c < 8; ++c) zv[c] = h2f(raw[c]);
        } else {
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            zv[c] = zr[bf][u][0][c];
            zv[4 + c] = zr[bf][u][EPT - 1][c];
          }
        }
        const f32x4 p0 = *(const f32x4*)(ps + j * 8), p1 = *(const f32x4*)(ps + j * 8 + 4);
#pragma unroll
        for (int hh = 0; hh < 8; ++hh) {
          if (hh < H) {
            const float pv = hh < 4 ? p0[hh & 3] : p1[hh & 3];
#pragma unroll
            for (int c = 0; c < 8; ++c) acc[hh][c] += pv * zv[c];
          }
        }
      }
    }
  };
  for (int k0 = 0; k0 < nk; k0 += 2 * OP_CH) {
    if (k0 + OP_CH < nk) fetch(1, k0 + OP_CH);
    consume(std::integral_constant<int, 0>{}, k0);
    if (k0 + 2 * OP_CH < nk) fetch(0, k0 + 2 * OP_CH);
    if (k0 + OP_CH < nk) consume(std::integral_constant<int, 1>{}, k0 + OP_CH);
  }
  FD_STAMP(2);
  // fold the slices that live in the same wave (lane bits above the channel-group bits)
#pragma unroll
  for (int o = NCG; o < 64; o <<= 1)
#pragma unroll
    for (int hh = 0; hh < 8; ++hh)
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[hh][c] += __shfl_xor(acc[hh][c], o, 64);
  if (lane < NCG) {
#pragma unroll
    for (int hh = 0; hh < 8; ++hh)
#pragma unroll
      for (int c = 0; c < 8; ++c) red[(wave * 8 + hh) * CZ + cg * 8 + c] = acc[hh][c];
  }
  // sum_j a[h,i,j] (= 1 up to rounding and masking): 32 threads per head
  {
    const int hh = tid >> 5, l5 = tid & 31;
    float sacc = 0.f;
    if (hh < H)
      for (int j = l5; j < N; j += 32) sacc += ps[j * 8 + hh];
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) sacc += __shfl_xor(sacc, o, 64);
    if (l5 == 0 && hh < 8) psum[hh] = sacc;
  }
  __syncthreads();
  FD_STAMP(3);
  for (int v = tid; v < 8 * CZ; v += FD_THREADS) red[v] = red[v] + red[8 * CZ + v] + red[16 * CZ + v] + red[24 * CZ + v];
  __syncthreads();
  FD_STAMP(4);
  for (int o = tid; o < H * CD; o += FD_THREADS) {
    const int hh = o / CD, d = o % CD;
    float s0 = a.bdz[d] * psum[hh], s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll 8
    for (int cc = 0; cc < CZ; cc += 4) {
      s0 += red[hh * CZ + cc] * wdl[cc * CD + d];
      s1 += red[hh * CZ + cc + 1] * wdl[(cc + 1) * CD + d];
      s2 += red[hh * CZ + cc + 2] * wdl[(cc + 2) * CD + d];
      s3 += red[hh * CZ + cc + 3] * wdl[(cc + 3) * CD + d];
    }
    a.out[(rb + i) * a.out_ld + a.off + hh * CD + d] = (s0 + s1) + (s2 + s3);
  }
  FD_STAMP(5);
}

// ------------------------------------------------------------------ o_pair on the matrix cores (bf16 z, c_z 128, 8 heads)
// az[h, c] = sum_j a[h,i,j] z[i,j,c] is a [8 x N] x [N x 128] product per (b, i) whose contraction index j is the ROW index
// of z in memory, while an MFMA B fragment wants 8 consecutive k per lane.  The z rows are therefore fetched as coalesced
// 16 B pieces, 64 keys at a time, and written TRANSPOSED into an LDS tile Zt[channel][key] (2-byte stores), from which
// B fragments are plain ds_read_b128; the attention weights are the A operand (heads = rows 0..7 of a 32-row tile, bf16,
// from LDS).  Wave w owns channels 32w..32w+31.  The tail (sum_j a, down_z projection) is the VALU code of opair_kernel.
#define OM_JC 64                    // keys per chunk
#define OM_ZROW (OM_JC * 2 + 16)    // bytes per Zt row (16 lanes of a b128 read hit 16 distinct slots)
// OM_NK: 16-key row groups per thread (all of a row's z pieces of one PASS are requested up front): 20 -> N <= 320, two blocks per
// CU.  NH passes over the key range (round 2): longer rows are taken in two passes of 32 groups (N <= 1024) so that the pieces of a
// pass fit 128 registers and TWO blocks share a CU and overlap each other's memory round trips - a single 64-group pass (256
// registers, one block per CU) streamed z at 2.7 TB/s at N = 724.
template <int N_, class F>
__device__ __forceinline__ void om_for(F&& f) {
  if constexpr (N_ > 0) {
    om_for<N_ - 1>(f);
    f(std::integral_constant<int, N_ - 1>{});
  }
}
// SB (round 4): ONE Zt buffer (two barriers per chunk instead of one), the cross-wave exchange tile `red` overlays it, and the down_z
// fragments are requested after the main loop instead of at the top: 24 KB of LDS and <= 128 registers -> FOUR blocks per CU instead
// of three.  A row is a latency chain (HBM round trip, five chunk hand-overs, a serial tail on one wave) whose matrix work is 2 % of
// its time: what hides it is other rows on the same CU.
template <int OM_NK, int LB, int NH = 1, bool SB = false>
__global__ __launch_bounds__(FD_THREADS, LB) void opair_mfma_kernel(OPairArgs a, int Np) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int CZ = 128, H = 8, CD = 32;
  const int N = a.N, nch = Np / OM_JC;
  const int prow = Np * 2 + 16;                             // bytes per attention-weight row (bf16)
  char* zt = smem;                                          // [2][CZ][OM_ZROW] ([1] with SB)
  char* pb = zt + (SB ? 1 : 2) * CZ * OM_ZROW;              // [8][prow] bf16 attention weights of this (b, i)
  constexpr int RS = CZ + 4;                                // floats per `red` row: the tail's b128 reads of 8 head rows then hit 8 bank groups
  float* red = SB ? (float*)zt : (float*)(pb + 8 * prow);   // [8][RS] (SB: the Zt buffer is dead when it is written)
  float* psum = SB ? (float*)(pb + 8 * prow) : red + 8 * RS;  // [8]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  const int i = blockIdx.x, b = blockIdx.y;
  const long rb = (long)b * N;
  const half_t* zrow = (const half_t*)a.z + (rb + i) * N * CZ;
  // this thread's pieces of the z rows: key PAIRS (2 jp, 2 jp + 1), jp = tid / 16 + 16 m, channels 8 (tid % 16) .. +7 — ALL
  // requested up front (one memory round trip per block); zr[2 m + w] = key 2 jp + w
  const int cg = tid & 15, jl0 = tid >> 4;
  u16x8 zr[OM_NK];
  auto request = [&](int pass) {
#pragma unroll
    for (int k = 0; k < OM_NK; ++k) {
      const int j = pass * (OM_NK * 16) + 2 * (jl0 + 16 * (k >> 1)) + (k & 1);
      zr[k] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      if (j < N) zr[k] = *(const u16x8*)(zrow + (long)j * CZ + 8 * cg);
    }
  };
  request(0);
  // down_z as B fragments (wave 0 only): requested now, used at the very end
  hx8 wdf[8], wdl[8];
  const bool dz_split = a.wdz_img_lo != nullptr;  // split operands for the down-projection (per-residue product: see OPairArgs)
  auto wdz_load = [&]() {
#pragma unroll
    for (int s = 0; s < 8; ++s) wdf[s] = __builtin_bit_cast(hx8, *(const u16x8*)((const char*)a.wdz_img + (s * 64 + lane) * 16));
    if (dz_split)
#pragma unroll
      for (int s = 0; s < 8; ++s) wdl[s] = __builtin_bit_cast(hx8, *(const u16x8*)((const char*)a.wdz_img_lo + (s * 64 + lane) * 16));
  };
  if (!SB && wave == 0) wdz_load();
  // attention weights -> bf16 rows (zero for padded keys) and sum_j a[h,i,j] (= 1 up to rounding and masking) in one pass:
  // 32 threads per head
  if (a.probs_h16) {  // already bf16 rows [b, i, h, probs_np] (attention3): 4 keys (8 B) per load, 32 threads per head
    const int hh = tid >> 5, l5 = tid & 31;
    const half_t* pr = a.probs_h16 + (((long)b * N + i) * H + hh) * a.probs_np;
    float sacc = 0.f;
    for (int j = 4 * l5; j < Np; j += 128) {
      u16x4 pv = {0, 0, 0, 0};
      if (j < a.probs_np) pv = *(const u16x4*)(pr + j);
      sacc += (h2f(pv[0]) + h2f(pv[1])) + (h2f(pv[2]) + h2f(pv[3]));
      *(u16x4*)(pb + hh * prow + 2 * j) = pv;
    }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) sacc += __shfl_xor(sacc, o, 64);
    if (l5 == 0) psum[hh] = sacc;
  } else {
    const int hh = tid >> 5, l5 = tid & 31;
    const float* pr = a.probs + (((long)b * H + hh) * N + i) * N;
    float sacc = 0.f;
    for (int j = l5; j < Np; j += 32) {
      const float pv = j < N ? pr[j] : 0.f;
      sacc += pv;
      *(half_t*)(pb + hh * prow + 2 * j) = f2h(pv);
    }
#pragma unroll
    for (int o = 1; o < 32; o <<= 1) sacc += __shfl_xor(sacc, o, 64);
    if (l5 == 0) psum[hh] = sacc;
  }
  // transposed write of chunk ch into slot, two keys per 32-bit store: channel 8 cg + e lives in Zt row 16 e + cg (with
  // 16 B-aligned rows a stride of 8 rows would put the 16 lanes of a store on one bank; consecutive rows spread them)
  auto scatter = [&](auto CH, int slot) {
    constexpr int ch = decltype(CH)::value;
    char* dst = zt + slot * CZ * OM_ZROW;
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int e = 0; e < 8; ++e)
        *(unsigned*)(dst + (16 * e + cg) * OM_ZROW + 4 * (jl0 + 16 * m)) =
            (unsigned)zr[4 * ch + 2 * m][e] | ((unsigned)zr[4 * ch + 2 * m + 1][e] << 16);
  };
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const char* arow = pb + (li < 8 ? li : 0) * prow + 16 * hi;  // A fragment source: row = head (lanes >= 8: zero)
#pragma unroll
  for (int pass = 0; pass < NH; ++pass) {
    if (pass > 0) {
      if (pass * (OM_NK / 4) >= nch) break;
      request(pass);
    }
    const int ch0 = pass * (OM_NK / 4);  // first chunk of the pass (the slots alternate on the chunk index inside the pass)
    scatter(std::integral_constant<int, 0>{}, 0);
    __syncthreads();
    auto chunk = [&](auto CH) {
      constexpr int ch = decltype(CH)::value;
      if (ch0 + ch < nch) {
        const char* zs = zt + (SB ? 0 : (ch & 1)) * CZ * OM_ZROW + (32 * wave + li) * OM_ZROW + 16 * hi;
#pragma unroll
        for (int s = 0; s < OM_JC / 16; ++s) {
          u16x8 af = {0, 0, 0, 0, 0, 0, 0, 0};
          if (li < 8) af = *(const u16x8*)(arow + 2 * ((ch0 + ch) * OM_JC + 16 * s));
          const u16x8 bfr = *(const u16x8*)(zs + 32 * s);
          acc = fd_mfma32(__builtin_bit_cast(hx8, af), __builtin_bit_cast(hx8, bfr), acc);
        }
        if (ch + 1 < OM_NK / 4 && ch0 + ch + 1 < nch) {
          if (SB) __syncthreads();  // (every wave is done with the buffer)
          scatter(std::integral_constant<int, (ch + 1 < OM_NK / 4 ? ch + 1 : 0)>{}, SB ? 0 : (ch + 1) & 1);
        }
        __syncthreads();
      }
    };
    om_for<OM_NK / 4>(chunk);
  }
  if (SB && wave == 0) wdz_load();  // (in flight across the exchange barrier)
  // D[h, Zt row]: lane (row 32 wave + li, hi) holds heads 4 hi + r in registers r < 4; row 16 e + cg = channel 8 cg + e
  {
    const int zrow_i = 32 * wave + li, ch_i = 8 * (zrow_i & 15) + (zrow_i >> 4);
#pragma unroll
    for (int r = 0; r < 4; ++r) red[(4 * hi + r) * RS + ch_i] = acc[r];
  }
  __syncthreads();
  // down_z (ipa_pytorch.py:317-322) on the matrix core: D[h, d] = sum_c az[h, c] Wdz[d, c], 8 k-steps, wave 0
  if (wave == 0) {
    f32x16 o2;
#pragma unroll
    for (int r = 0; r < 16; ++r) o2[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (li < 8) {
        const f32x4 x0 = *(const f32x4*)(red + li * RS + 16 * s + 8 * hi), x1 = *(const f32x4*)(red + li * RS + 16 * s + 8 * hi + 4);
        v[0] = x0[0]; v[1] = x0[1]; v[2] = x0[2]; v[3] = x0[3]; v[4] = x1[0]; v[5] = x1[1]; v[6] = x1[2]; v[7] = x1[3];
      }
      hx8 af, al;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        af[e] = (fd_h)v[e];
        al[e] = (fd_h)(v[e] - (float)af[e]);
      }
      if (dz_split) {
        o2 = fd_mfma32(af, wdl[s], o2);
        o2 = fd_mfma32(al, wdf[s], o2);
      }
      o2 = fd_mfma32(af, wdf[s], o2);
    }
    const float bd = a.bdz[li];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = o2[r] + bd * psum[4 * hi + r];
      if (a.out_h16) a.out_h16[(rb + i) * a.out_ld + a.off + (4 * hi + r) * CD + li] = f2h(v);
      else a.out[(rb + i) * a.out_ld + a.off + (4 * hi + r) * CD + li] = v;
    }
  }
}

template <class ZT>
static int launch_opair(const OPairArgs& a, hipStream_t st) {
  const size_t smem = (size_t)8 * a.N * 4 + (size_t)(4 * 8 * a.CZ + 8) * 4 + (size_t)a.CZ * a.CD * 4;
  if (smem > 64 * 1024) return FDIPT_ESIZE;
  if (a.CZ == 128 && a.H == 8) hipLaunchKernelGGL((opair_kernel<ZT, 128, 8>), dim3(a.N, a.B), dim3(FD_THREADS), smem, st, a);
  else if (a.CZ == 128) hipLaunchKernelGGL((opair_kernel<ZT, 128, 0>), dim3(a.N, a.B), dim3(FD_THREADS), smem, st, a);
  else if (a.CZ == 32) hipLaunchKernelGGL((opair_kernel<ZT, 32, 0>), dim3(a.N, a.B), dim3(FD_THREADS), smem, st, a);
  else return FDIPT_EINVAL;
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

int fd_opair_mfma_eligible(int precision, const OPairArgs& a) {
  return precision != FDIPT_PREC_F32 && a.CZ == 128 && a.H == 8 && a.CD == 32 && a.wdz_img && a.N <= 1024 && !FD_DEV_ENV("FDIPT_OPAIR_VALU");
}
int fd_opair(int precision, const OPairArgs& a, hipStream_t st) {
  if (a.H > 8) return FDIPT_ESIZE;
  if (fd_opair_mfma_eligible(precision, a)) {
    if (a.probs_h16 && (a.probs_np & 3)) return FDIPT_EINVAL;
    const int Np = (a.N + OM_JC - 1) / OM_JC * OM_JC;
#ifndef OM_PAD
#define OM_PAD 0  // (tools/micro/opair_bench.hip: extra dynamic LDS = fewer blocks per CU)
#endif
    const size_t smem = (size_t)2 * 128 * OM_ZROW + (size_t)8 * (Np * 2 + 16) + (size_t)(8 * 132 + 8) * 4 + OM_PAD;
#ifndef OM_SB
#define OM_SB 1   // N <= 320: the four-blocks-per-CU form (tools/micro/opair_bench.hip -DOM_SB=0: three blocks, double-buffered Zt)
#endif
    if (a.N <= 320 && OM_SB) {
      const size_t smem_sb = (size_t)128 * OM_ZROW + (size_t)8 * (Np * 2 + 16) + 64 + OM_PAD;
      hipLaunchKernelGGL((opair_mfma_kernel<20, OM_SB == 1 ? 4 : OM_SB, 1, true>), dim3(a.N, a.B), dim3(FD_THREADS), smem_sb, st, a, Np);
    } else if (a.N <= 320) hipLaunchKernelGGL((opair_mfma_kernel<20, 2>), dim3(a.N, a.B), dim3(FD_THREADS), smem, st, a, Np);
#ifndef OM_SB_LONG
#define OM_SB_LONG 1  // N > 320: the four-blocks-per-CU form in two to four passes of 320 keys (round 4, third session: 292 -> 258 us at N = 776, B = 8,
                      // against <32, 2, 2>: two blocks per CU, 226 registers)
#endif
    else if (OM_SB_LONG) {  // (N <= 1024: fd_opair_mfma_eligible)
      const size_t smem_sb = (size_t)128 * OM_ZROW + (size_t)8 * (Np * 2 + 16) + 64 + OM_PAD;
      if (a.N <= 640) hipLaunchKernelGGL((opair_mfma_kernel<20, 4, 2, true>), dim3(a.N, a.B), dim3(FD_THREADS), smem_sb, st, a, Np);
      else if (a.N <= 960) hipLaunchKernelGGL((opair_mfma_kernel<20, 4, 3, true>), dim3(a.N, a.B), dim3(FD_THREADS), smem_sb, st, a, Np);
      else hipLaunchKernelGGL((opair_mfma_kernel<20, 4, 4, true>), dim3(a.N, a.B), dim3(FD_THREADS), smem_sb, st, a, Np);
    }
#ifndef OM_MID
#define OM_MID 0
#endif
    else if (a.N <= 512 && OM_MID) hipLaunchKernelGGL((opair_mfma_kernel<32, 2, 1>), dim3(a.N, a.B), dim3(FD_THREADS), smem, st, a, Np);
    else if (a.N <= 640 && !OM_MID) hipLaunchKernelGGL((opair_mfma_kernel<40, 1>), dim3(a.N, a.B), dim3(FD_THREADS), smem, st, a, Np);
    else hipLaunchKernelGGL((opair_mfma_kernel<32, 2, 2>), dim3(a.N, a.B), dim3(FD_THREADS), smem, st, a, Np);
    FD_CHECK_LAUNCH();
    return FDIPT_OK;
  }
  if (a.probs_h16 || a.out_h16) return FDIPT_EINVAL;  // the VALU kernels read fp32 weights and write fp32 features
  return precision == FDIPT_PREC_F32 ? launch_opair<float>(a, st) : launch_opair<half_t>(a, st);
}

// ------------------------------------------------------------------ o_pair from the producer-emitted pair_z image (round 6)
// o_pair[i, h, :] = sum_j a[h, i, j] pair_z[i, j, :] (ipa_pytorch.py:317-322) where pair_z = down_z(z) + b was written by the kernel that
// produced z (edge embedder epilogue for block 0, EdgeTransition epilogue after: kernels.hpp fd_pz_bytes) — 64 B per pair instead of the
// 256 B of z that opair_mfma_kernel streams (184 -> 46 MB per call at N = 300, B = 8).  One wave per residue row (b, i): D[h, d] over
// k-steps of 16 keys; A = the attention weights (half-precision rows [b, i, h, probs_np] from attention3, head rows 0..7 of the 32-row tile),
// B = two 8 B pieces of the image per lane (key groups 4 s + 2 (lane >> 5) and + 1 of k-step s).  No LDS, no barrier; every load of a pass of
// OP_KS k-steps is requested before the first MFMA.
#define OP_KS 20
__global__ __launch_bounds__(FD_THREADS) void opair_pz_kernel(OPairArgs a, int n_rows, int NJ4) {
  const int lane = threadIdx.x & 63, li = lane & 31, hi = lane >> 5;
  const int row = blockIdx.x * (FD_THREADS / 64) + (threadIdx.x >> 6);
  if (row >= n_rows) return;
  const int ksteps = (a.N + 15) >> 4;
  const half_t* pr = a.probs_h16 + ((long)row * 8 + (li & 7)) * a.probs_np + 8 * hi;
  const char* pzr = (const char*)a.pz + (long)row * NJ4 * 256 + li * 8;
  typedef unsigned long long u64;
  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
  for (int s0 = 0; s0 < ksteps; s0 += OP_KS) {
    u16x8 af[OP_KS];
    u64 b0[OP_KS], b1[OP_KS];
#pragma unroll
    for (int u = 0; u < OP_KS; ++u) {
      const int s = s0 + u, g4 = 4 * s + 2 * hi;
      af[u] = u16x8{0, 0, 0, 0, 0, 0, 0, 0};
      b0[u] = b1[u] = 0ull;
      if (s < ksteps) {
        if (li < 8) af[u] = *(const u16x8*)(pr + 16 * s);
        if (g4 < NJ4) b0[u] = *(const u64*)(pzr + (long)g4 * 256);
        if (g4 + 1 < NJ4) b1[u] = *(const u64*)(pzr + (long)(g4 + 1) * 256);
      }
    }
#pragma unroll
    for (int u = 0; u < OP_KS; ++u) {
      if (s0 + u < ksteps) {
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        const hx8 bf = __builtin_bit_cast(hx8, u64x2{b0[u], b1[u]});
        if (u & 1) acc1 = fd_mfma32(__builtin_bit_cast(hx8, af[u]), bf, acc1);
        else acc0 = fd_mfma32(__builtin_bit_cast(hx8, af[u]), bf, acc0);
      }
    }
  }
  // D[h, d]: lane (d = li, hi) holds heads 4 hi + r in registers r < 4
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float v = acc0[r] + acc1[r];
    const long o = (long)row * a.out_ld + a.off + (4 * hi + r) * 32 + li;
    if (a.out_h16) a.out_h16[o] = f2h(v);
    else a.out[o] = v;
  }
}
int fd_opair_pz(const OPairArgs& a, hipStream_t st) {
  if (!a.pz || !a.probs_h16 || a.H != 8 || a.CD != 32 || (a.probs_np & 7) || a.probs_np < ((a.N + 15) & ~15)) return FDIPT_EINVAL;
  const int n_rows = a.B * a.N;
  hipLaunchKernelGGL(opair_pz_kernel, dim3(cdiv(n_rows, FD_THREADS / 64)), dim3(FD_THREADS), 0, st, a, n_rows, (a.N + 3) >> 2);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// ------------------------------------------------------------------ projected points -> global frame

__device__ __forceinline__ int pt_perm16(int pos) { return 4 * (pos >> 3) + (pos & 3) + 8 * ((pos & 7) >> 2); }

__global__ void points_kernel(PointsArgs a) {
  const long r = blockIdx.x;  // residue row b*N+i
  const int tid = threadIdx.x;
  const float* q = a.quat + r * 4;
  const float w = q[0], x = q[1], y = q[2], z = q[3];
  // quat_to_rot (openfold/utils/rigid_utils.py:173-205), no normalisation
  float R[9];
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * x * y - 2 * w * z; R[2] = 2 * x * z + 2 * w * y;
  R[3] = 2 * x * y + 2 * w * z; R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * y * z - 2 * w * x;
  R[6] = 2 * x * z - 2 * w * y; R[7] = 2 * y * z + 2 * w * x; R[8] = w * w - x * x - y * y + z * z;
  const float tx = a.trans[r * 3], ty = a.trans[r * 3 + 1], tz = a.trans[r * 3 + 2];
  if (tid < 9) a.rot[r * 9 + tid] = R[tid];
  const int HPq = a.H * a.Pq, Pkv = a.Pq + a.Pv, HPkv = a.H * Pkv;
  const float* row = a.proj + r * a.ld;
  for (int p = tid; p < HPq + HPkv; p += blockDim.x) {
    float px, py, pz;
    float* dst;
    if (p < HPq) {
      px = row[a.q_off + p]; py = row[a.q_off + HPq + p]; pz = row[a.q_off + 2 * HPq + p];
      dst = a.qp + (r * HPq + p) * 3;
    } else {
      const int pp = p - HPq;
      px = row[a.kv_off + pp]; py = row[a.kv_off + HPkv + pp]; pz = row[a.kv_off + 2 * HPkv + pp];
      const int hh = pp / Pkv, e = pp % Pkv;
      dst = e < a.Pq ? a.kp + ((r * a.H + hh) * a.Pq + e) * 3 : a.vp + ((r * a.H + hh) * a.Pv + (e - a.Pq)) * 3;
    }
    const float gx = R[0] * px + R[1] * py + R[2] * pz + tx;
    const float gy = R[3] * px + R[4] * py + R[5] * pz + ty;
    const float gz = R[6] * px + R[7] * py + R[8] * pz + tz;
    fd_store3(dst, gx, gy, gz);
    if (a.vpt && p >= HPq) {
      const int pp2 = p - HPq, hh = pp2 / Pkv, e = pp2 % Pkv;
      if (e >= a.Pq) {  // a value point: coordinates 3 (e - Pq) + {0,1,2} of head hh, key = residue index in its sample
        const long bidx = r / a.N;
        const int key = (int)(r - bidx * a.N), pos = (key & ~15) + pt_perm16(key & 15), ks = a.Np >> 4;
        const float g3[3] = {gx, gy, gz};
        for (int c = 0; c < 3; ++c) {
          const int row = 3 * (e - a.Pq) + c;
          const unsigned short vh = f2h(g3[c]);
          const unsigned short vl = f2h(g3[c] - h2f(vh));
          const long base = (bidx * a.H + hh) * 3;
          // rows < 36 (tile 0 and the first 4 rows of tile 1): high parts; rows 36..71: low parts
          const int rh = row, rl = 36 + row;
          a.vpt[((((base + (rh >> 5)) * ks + (pos >> 4)) * 64 + ((pos >> 3) & 1) * 32 + (rh & 31)) << 3) + (pos & 7)] = vh;
          a.vpt[((((base + (rl >> 5)) * ks + (pos >> 4)) * 64 + ((pos >> 3) & 1) * 32 + (rl & 31)) << 3) + (pos & 7)] = vl;
        }
      }
    }
  }
}

// The same for the attention3 path (value-point image wanted, Pv == 12, H % 2 == 0): one block = the 16 keys of one fragment
// key group of a sample x half of the heads, 16 threads per key.  The image leaves as whole 16 B units (8 key slots of one
// coordinate row) staged through LDS — the per-residue kernel above scatters it as single bf16 values (576 partial-sector
// stores per residue) — and every global operand of a thread (quaternion, translation, its 7 points) is requested up front:
// one memory round trip per block.
__global__ __launch_bounds__(256) void points16_kernel(PointsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  unsigned short* vs = (unsigned short*)smem;  // [H/2][72][16] bf16: hi rows 0..35, lo rows 36..71, permuted key slot
  float* kps = (float*)(smem + (size_t)(a.H >> 1) * 72 * 16 * 2);  // kpf: [H/2][16 keys][24] global-frame key points (fp32)
  const int tid = threadIdx.x, kk = tid >> 4, sub = tid & 15;
  const int ng = (a.N + 15) >> 4, HH = a.H >> 1;
  const int half = blockIdx.x & 1, bg = blockIdx.x >> 1, b = bg / ng, g = bg - b * ng;
  const int key = 16 * g + kk;
  const bool live = key < a.N;
  const long r = (long)b * a.N + (live ? key : a.N - 1);
  const int HPq = a.H * a.Pq, Pkv = a.Pq + a.Pv, HPkv = a.H * Pkv;
  const int nq = HH * a.Pq, nkv = HH * Pkv;          // this block's points per key: query points, key/value points
  constexpr int MAXI = 8;                             // (nq + nkv) / 16 <= 8 iterations per thread (H = 8: 2 + 5)
  const float* row = a.proj + r * a.ld;
  const f32x4 q4 = *(const f32x4*)(a.quat + r * 4);
  const float tx = a.trans[r * 3], ty = a.trans[r * 3 + 1], tz = a.trans[r * 3 + 2];
  float px[MAXI], py[MAXI], pz[MAXI];
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int p = sub + 16 * it;
    px[it] = py[it] = pz[it] = 0.f;
    if (p < nq) {
      const int c = half * nq + p;
      px[it] = row[a.q_off + c]; py[it] = row[a.q_off + HPq + c]; pz[it] = row[a.q_off + 2 * HPq + c];
    } else if (p < nq + nkv) {
      const int c = half * nkv + (p - nq);
      px[it] = row[a.kv_off + c]; py[it] = row[a.kv_off + HPkv + c]; pz[it] = row[a.kv_off + 2 * HPkv + c];
    }
  }
  // merged projection: this block's node-row pieces (block half 0: two 8-channel units of the K image rows, half 1: two V units = 8
  // keys of one channel each) are requested with the point operands — one memory round trip for the whole block
  constexpr int NC = 256;
  float nx[16];
  if (a.node) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int u = tid + 256 * q;
      if (half == 0) {
        const int cg = u & 31, rr = 16 * g + (u >> 5);
        const float* xp = a.node + ((long)b * a.N + (rr < a.N ? rr : a.N - 1)) * a.ld_node + 8 * cg;
        const float4 x0 = *(const float4*)xp, x1 = *(const float4*)(xp + 4);
        nx[8 * q] = x0.x; nx[8 * q + 1] = x0.y; nx[8 * q + 2] = x0.z; nx[8 * q + 3] = x0.w;
        nx[8 * q + 4] = x1.x; nx[8 * q + 5] = x1.y; nx[8 * q + 6] = x1.z; nx[8 * q + 7] = x1.w;
      } else {
        const int ln = u & 63, hf = ln >> 5, cc = 32 * (u >> 6) + (ln & 31);
#pragma unroll
        for (int e = 0; e < 8; ++e) {  // slot 8 hf + e of the 16-group -> key (inverse of pt_perm16)
          const int sl = 8 * hf + e, pos = 8 * ((sl >> 2) & 1) + 4 * (sl >> 3) + (sl & 3), ky = 16 * g + pos;
          nx[8 * q + e] = a.node[((long)b * a.N + (ky < a.N ? ky : a.N - 1)) * a.ld_node + cc];
        }
      }
    }
  }
  if (!live)  // keys beyond the sample: their slots of the image stay zero
    for (int v = sub; v < HH * 72; v += 16) vs[v * 16 + pt_perm16(kk)] = 0;
  const float w = q4[0], x = q4[1], y = q4[2], z = q4[3];
  float R[9];  // quat_to_rot (openfold/utils/rigid_utils.py:173-205), no normalisation
  R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * x * y - 2 * w * z; R[2] = 2 * x * z + 2 * w * y;
  R[3] = 2 * x * y + 2 * w * z; R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * y * z - 2 * w * x;
  R[6] = 2 * x * z - 2 * w * y; R[7] = 2 * y * z + 2 * w * x; R[8] = w * w - x * x - y * y + z * z;
  if (live && half == 0 && sub < 9) a.rot[r * 9 + sub] = R[sub];
  const int slot = pt_perm16(kk);
#pragma unroll
  for (int it = 0; it < MAXI; ++it) {
    const int p = sub + 16 * it;
    if (!live || p >= nq + nkv) continue;
    const float g3[3] = {R[0] * px[it] + R[1] * py[it] + R[2] * pz[it] + tx, R[3] * px[it] + R[4] * py[it] + R[5] * pz[it] + ty,
                         R[6] * px[it] + R[7] * py[it] + R[8] * pz[it] + tz};
    float* dst;
    if (p < nq) {
      dst = a.qp + (r * HPq + half * nq + p) * 3;
    } else {
      const int pp = p - nq, hl = pp / Pkv, e = pp - hl * Pkv, hh = half * HH + hl;
      dst = e < a.Pq ? a.kp + ((r * a.H + hh) * a.Pq + e) * 3 : a.vp + ((r * a.H + hh) * a.Pv + (e - a.Pq)) * 3;
      if (e >= a.Pq) {  // a value point: coordinate rows 3 (e - Pq) + {0,1,2} of head hh
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const int rw = 3 * (e - a.Pq) + c;
          const unsigned short vh = f2h(g3[c]);
          vs[(hl * 72 + rw) * 16 + slot] = vh;
          vs[(hl * 72 + 36 + rw) * 16 + slot] = f2h(g3[c] - h2f(vh));
        }
      } else if (a.kpf) {  // a key point: staged for the fragment image below
#pragma unroll
        for (int c = 0; c < 3; ++c) kps[(hl * 16 + kk) * 24 + 3 * e + c] = g3[c];
      }
    }
    fd_store3(dst, g3[0], g3[1], g3[2]);  // (not one dwordx3 store: common.hpp)
  }
  __syncthreads();
  const int ks = a.Np >> 4;
  for (int u = tid; u < HH * 144; u += 256) {
    const int hl = u / 144, rem = u - hl * 144, rw = rem >> 1, hf = rem & 1;
    const uint4 val = *(const uint4*)(vs + (hl * 72 + rw) * 16 + 8 * hf);
    *(uint4*)(a.vpt + (((((long)b * a.H + half * HH + hl) * 3 + (rw >> 5)) * ks + g) * 64 + hf * 32 + (rw & 31)) * 8) = val;
  }
  if (a.kpf) {
    // key-point fragment image (kernels.hpp: fd_kpf layout): 8 units of 16 B per (head, key) — fp16 hi / lo thirds of the 24 coordinates
    // and the mask / norm unit X; the last group of a sample also writes the all-padding group up to Np when there is one
    const int ntl = a.Np >> 5;
    const int g_end = g == ng - 1 ? (a.Np >> 4) : g + 1;
    for (int gg = g; gg < g_end; ++gg)
      for (int u = tid; u < HH * 128; u += 256) {
        const int k16 = u & 15, un = (u >> 4) & 7, hl = u >> 7, f = un >> 1, hf = un & 1, hh = half * HH + hl;  // (16 lanes = 256 B of one unit row)
        const int keyu = 16 * gg + k16;
        const bool lv = gg == g && keyu < a.N;
        const float* kv = kps + (hl * 16 + k16) * 24;
        unsigned short o[8];
        if (f == 3 && hf == 1) {
          float kn = 0.f;
          if (lv)
#pragma unroll
            for (int c = 0; c < 24; ++c) kn = fmaf(kv[c], kv[c], kn);
          const float m = lv ? a.res_mask[(long)b * a.N + keyu] : 0.f;
          // (saturated at the fp16 range: beyond |k| ~ 360 nm / sqrt(gamma) from the origin — 3.6 um in unscaled units, i.e. inputs that
          //  are not centred — the three-part norm would overflow to -inf / NaN; such keys keep a finite, still dominating penalty)
          const float t0 = fmaxf(-0.5f * a.gamma[hh] * kn, -65000.f);
          const unsigned short p0 = f2f16(t0);
          const float t1 = t0 - f162f(p0);
          const unsigned short p1 = f2f16(t1);
          const unsigned short p2 = f2f16(t1 - f162f(p1));
          o[0] = f2f16(2.f * m); o[1] = f2f16(m); o[2] = f2f16(lv ? 0.f : -60000.f); o[3] = p0; o[4] = p1; o[5] = p2; o[6] = o[7] = 0;
        } else {
          const int c0 = f < 2 ? 8 * hf : 16;
          const bool lo = f == 1 || (f == 2 && hf == 1);
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const float x = lv ? kv[c0 + e] : 0.f;
            const unsigned short xh = f2f16(x);
            o[e] = lo ? f2f16(x - f162f(xh)) : xh;
          }
        }
        uint4 val;
        val.x = o[0] | ((unsigned)o[1] << 16); val.y = o[2] | ((unsigned)o[3] << 16); val.z = o[4] | ((unsigned)o[5] << 16); val.w = o[6] | ((unsigned)o[7] << 16);
        *(uint4*)(a.kpf + ((((((long)b * a.H + hh) * ntl + (keyu >> 5)) * FD_KPF_FRAGS + f) * 64 + hf * 32 + (keyu & 31)) << 3)) = val;
      }
  }
  if (a.node) {
    // merged projection: this 16-group of keys of the node-row images (c_s = 256): block half 0 writes the K image rows, half 1 the
    // V_hi / V_lo units (operands requested at the top); the last group of a sample also covers the padded groups up to Np (zeros)
    const int ntl = a.Np >> 5;
    const int g_end = g == ng - 1 ? (a.Np >> 4) : g + 1;
    for (int gg = g; gg < g_end; ++gg) {
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        const int u = tid + 256 * q;
        if (half == 0) {
          const int cg = u & 31, rr = 16 * gg + (u >> 5);
          uint4 o = make_uint4(0, 0, 0, 0);
          if (rr < a.N) o = make_uint4(fd_cvt_pk(nx[8 * q], nx[8 * q + 1]), fd_cvt_pk(nx[8 * q + 2], nx[8 * q + 3]), fd_cvt_pk(nx[8 * q + 4], nx[8 * q + 5]),
                                       fd_cvt_pk(nx[8 * q + 6], nx[8 * q + 7]));
          *(uint4*)(a.nKb + ((((long)b * ntl + (rr >> 5)) * (NC >> 4) + (cg >> 1)) * 64 + (cg & 1) * 32 + (rr & 31)) * 8) = o;
        } else {
          const int ln = u & 63, hf = ln >> 5, dt = u >> 6;
          float xv[8];
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            const int sl = 8 * hf + e, pos = 8 * ((sl >> 2) & 1) + 4 * (sl >> 3) + (sl & 3);
            xv[e] = 16 * gg + pos < a.N ? nx[8 * q + e] : 0.f;
          }
          uint4 oh, ol;
          unsigned* ph = (unsigned*)&oh;
          unsigned* pl = (unsigned*)&ol;
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            ph[qq] = fd_cvt_pk(xv[2 * qq], xv[2 * qq + 1]);
            pl[qq] = fd_cvt_pk(xv[2 * qq] - h2f(f2h(xv[2 * qq])), xv[2 * qq + 1] - h2f(f2h(xv[2 * qq + 1])));
          }
          const long v = (((long)b * (NC / 32) + dt) * (2 * ntl) + gg) * 64 + ln;
          *(uint4*)(a.nVt + v * 8) = oh;
          if (a.nVt_lo) *(uint4*)(a.nVt_lo + v * 8) = ol;
        }
      }
    }
  }
}

int fd_points(const PointsArgs& a, hipStream_t st) {
  if (a.vpt && a.Pv == 12 && (a.H & 1) == 0 && (a.H / 2) * (2 * a.Pq + a.Pv) <= 128 && (a.ld & 0) == 0 && !FD_DEV_ENV("FDIPT_POINTS_V1")) {
    if (a.kpf && (a.Pq != 8 || !a.gamma || !a.res_mask)) return FDIPT_EINVAL;
    const size_t smem = (size_t)(a.H / 2) * 72 * 16 * 2 + (a.kpf ? (size_t)(a.H / 2) * 16 * 24 * 4 : 0);
    hipLaunchKernelGGL(points16_kernel, dim3(2 * a.B * ((a.N + 15) / 16)), dim3(256), smem, st, a);
    FD_CHECK_LAUNCH();
    return FDIPT_OK;
  }
  if (a.node || a.kpf) return FDIPT_EINVAL;  // (the node-row / key-point images ride on the 16-keys-per-block kernel only)
  hipLaunchKernelGGL(points_kernel, dim3(a.B * a.N), dim3(256), 0, st, a);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// ------------------------------------------------------------------ fp32 mode (round 5): IPA attention with the scores in registers
// attn_kernel<PrecF32, true> above stages every operand tile through LDS behind two barriers per 32-wide k-step (1.48 ms per call at
// N = 1000, B = 4; 272 us at N = 300, B = 8: 22 TF/s of the 157 TF/s fp32 matrix peak, 6.6 % of a config-5 step).  This kernel keeps
// the block = 32 queries of one (sample, head) and computes the same quantities in the formulation of ipa_attn3_kernel on
// v_mfma_f32_32x32x2_f32 (exact fp32 products):
//   * S^T[key, query]: key tiles of 32 dealt round-robin to the four waves, A = K rows straight from global memory (an fp32 MFMA takes
//     one k value per lane and WHICH channel a (lane half, step) pair stands for is free as long as A and B agree: lane half hi walks
//     channels 128 hi .. 128 hi + 127, i.e. consecutive floats of its row, 16 B loads, 32 channels per chunk, double-buffered);
//     B = the block's Q rows (pre-scaled) from LDS [channel][query];
//   * point term on the matrix cores: -gamma/2 |q - k|^2 = gamma q.k - gamma/2 |k|^2 - gamma/2 |q|^2; the last term is constant along a
//     softmax row and dropped, the first is 12 more MFMAs (24 coordinates), the second one MFMA against a column of ones;
//   * pair bias ([B,N,N,H]) and the reference's mask term 1e5 (m_i m_j - 1) are added to the accumulators; padded keys get -1e30;
//   * softmax: registers + lane^32 + 2 x 128 floats of LDS across the waves; the weights go to `probs` ([B,H,N,N], o_pair reads them) and
//     to LDS ([key][query]);
//   * O^T[d, query] = V^T P^T: ten row tiles of 32 (eight of the 256 channels, two of the 36 value-point coordinates), A = two rows of V
//     per MFMA (coalesced 128 B), B = two LDS rows of P; the point tiles order their rows so that a lane ends up with whole (x, y, z)
//     triples of five points and applies R_i^T (. - t_i) and the norm in registers.
// Reference widths only (C = Dv = 256, 8 query / 12 value points); everything else stays on attn_kernel.
#define AF_C 256
#define AF_HC 128   // channels per lane half
#define AF_CH 32    // channels per K chunk
template <int NTW, int LB>
__global__ __launch_bounds__(FD_THREADS, LB) void ipa_attn_f32_kernel(AttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.N, H = a.H, nt = (N + 31) >> 5;
  float* Qs = (float*)smem;            // [256 + 24 + 2][32]: channels, gamma * query points, a row of ones, a row of zeros
  float* Ps = (float*)smem;            // [32 nt][32]: the weights, once every wave is done with Qs (behind the softmax barriers)
  const int un = (AF_C + 26) * 32 > 32 * nt * 32 ? (AF_C + 26) * 32 : 32 * nt * 32;
  float* mxs = Qs + un;                // [4][32]
  float* sms = mxs + 128;              // [4][32]
  float* msk = sms + 128;              // [32 nt] res_mask of the sample's keys (0 beyond N)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  int b, h, qt;
  {  // the 8 heads of a (sample, query tile) on one XCD: they read the same [i, j, :] bias lines
    const int id = blockIdx.x, xcd = id & 7, local = id >> 3, groups = a.B * nt, per = (groups + 7) >> 3;
    const int g = xcd * per + local / H;
    h = local % H;
    if (local >= per * H || g >= groups) return;
    b = g / nt;
    qt = g - b * nt;
  }
  const long rb = (long)b * N;
  const int i0 = 32 * qt;
  const float gam = a.gamma[h];
  // ---- stage Q (scaled), gamma * q_pts, ones / zeros and the key mask
  for (int v = tid; v < 32 * (AF_C / 4); v += FD_THREADS) {
    const int r = v / (AF_C / 4), c4 = (v % (AF_C / 4)) * 4;
    const int row = i0 + r < N ? i0 + r : N - 1;
    const f32x4 x = *(const f32x4*)(a.q + (rb + row) * a.q_ld + (long)h * a.q_hs + c4);
#pragma unroll
    for (int q = 0; q < 4; ++q) Qs[(c4 + q) * 32 + r] = x[q] * a.scale;
  }
  for (int v = tid; v < 32 * 24; v += FD_THREADS) {
    const int r = v / 24, c = v % 24;
    const int row = i0 + r < N ? i0 + r : N - 1;
    Qs[(AF_C + c) * 32 + r] = gam * a.qp[((rb + row) * H + h) * 24 + c];
  }
  if (tid < 64) Qs[(AF_C + 24) * 32 + tid] = tid < 32 ? 1.f : 0.f;
  for (int v = tid; v < 32 * nt; v += FD_THREADS) msk[v] = v < N ? a.res_mask[rb + v] : 0.f;
  const int qi = i0 + li;
  const float mi = qi < N ? a.res_mask[rb + qi] : 0.f;
  __syncthreads();
  // ---- scores of this wave's key tiles
  f32x16 S[NTW];
  float mx = -3.0e38f;
  float Kc[2][AF_CH];
  auto kload = [&](auto BUF, int t, int c) {
    constexpr int bf = decltype(BUF)::value;
    const int key = 32 * t + li, krow = key < N ? key : N - 1;
    const float* kp = a.k + (rb + krow) * a.k_ld + (long)h * a.k_hs + AF_HC * hi + AF_CH * c;
#pragma unroll
    for (int s = 0; s < AF_CH; s += 4) {
      const f32x4 v = *(const f32x4*)(kp + s);
      Kc[bf][s] = v[0]; Kc[bf][s + 1] = v[1]; Kc[bf][s + 2] = v[2]; Kc[bf][s + 3] = v[3];
    }
  };
  constexpr std::integral_constant<int, 0> B0{};
  constexpr std::integral_constant<int, 1> B1{};
  if (wave < nt) kload(B0, wave, 0);
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int t = wave + 4 * u;
    if (t < nt) {
      const int key = 32 * t + li, krow = key < N ? key : N - 1;
      // the tile's small operands: key points (this lane half's 12 coordinates), pair bias of the 16 (key, query) elements
      float Kp[12];
      {
        const float* pp = a.kp + ((rb + krow) * H + h) * 24 + 12 * hi;
#pragma unroll
        for (int s = 0; s < 12; s += 4) {
          const f32x4 v = *(const f32x4*)(pp + s);
          Kp[s] = v[0]; Kp[s + 1] = v[1]; Kp[s + 2] = v[2]; Kp[s + 3] = v[3];
        }
      }
      float bs[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int j = 32 * t + c_row(r, lane);
        bs[r] = (j < N && qi < N) ? a.bias[((rb + qi) * N + j) * H + h] : 0.f;
      }
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int c = 0; c < AF_HC / AF_CH; ++c) {
        if (c + 1 < AF_HC / AF_CH) {
          if (c & 1) kload(B0, t, c + 1);
          else kload(B1, t, c + 1);
        } else if (t + 4 < nt && u + 1 < NTW) {
          kload(B0, t + 4, 0);  // (AF_HC / AF_CH is even: chunk 0 of the next tile lands in buffer 0)
        }
        const float* qrow = Qs + (AF_HC * hi + AF_CH * c) * 32 + li;
#pragma unroll
        for (int s = 0; s < AF_CH; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kc[c & 1][s], qrow[s * 32], acc, 0, 0, 0);
      }
      {  // gamma q_pt . k_pt - gamma / 2 |k_pt|^2
        float kn = 0.f;
#pragma unroll
        for (int s = 0; s < 12; ++s) kn = fmaf(Kp[s], Kp[s], kn);
        kn += __shfl_xor(kn, 32, 64);
        const float* qrow = Qs + (AF_C + 12 * hi) * 32 + li;
#pragma unroll
        for (int s = 0; s < 12; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Kp[s], qrow[s * 32], acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi == 0 ? -0.5f * gam * kn : 0.f, Qs[(AF_C + 24) * 32 + hi * 32 + li], acc, 0, 0, 0);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jl = 32 * t + c_row(r, lane);
        float s = acc[r] + bs[r];
        s += 1e5f * (mi * msk[jl] - 1.f);
        if (jl >= N) s = -1.0e30f;
        acc[r] = s;
        mx = fmaxf(mx, s);
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
  float* prow = a.probs + (((long)b * H + h) * N + (qi < N ? qi : 0)) * N;
  const bool vec4 = (N & 3) == 0;
#pragma unroll
  for (int u = 0; u < NTW; ++u) {
    const int t = wave + 4 * u;
    if (t < nt) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j0 = 32 * t + 8 * g + 4 * hi;
        f32x4 p;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          p[q] = S[u][4 * g + q] * inv;
          Ps[(j0 + q) * 32 + li] = p[q];
        }
        if (qi < N) {
          if (vec4 && j0 + 3 < N) *(f32x4*)(prow + j0) = p;
          else
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (j0 + q < N) prow[j0 + q] = p[q];
        }
      }
    }
  }
  __syncthreads();
  // ---- O^T = V^T P^T: row tiles 0..7 = channels, 8..9 = value-point coordinates; wave w owns tiles w, w + 4, w + 8
  const int nm = 16 * nt;
  constexpr int DEPTH = 16;
  for (int T = wave; T < 10; T += 4) {
    const float* base;
    long stride;
    bool alive = true;
    int pnt = 0;
    if (T < 8) {
      base = a.v + rb * a.v_ld + (long)h * a.v_hs + 32 * T + li;
      stride = a.v_ld;
    } else {
      // row li of a point tile: lane half hh = (li >> 2) & 1 of the OUTPUT fragment, slot = (li & 3) + 4 (li >> 3) = register index there;
      // slot -> (point, coordinate) = (slot / 3, slot % 3): tile 8 holds points 5 hh .. 5 hh + 4, tile 9 points 10 + hh
      const int hh = (li >> 2) & 1, slot = (li & 3) + 4 * (li >> 3);
      pnt = T == 8 ? 5 * hh + slot / 3 : 10 + hh;
      alive = T == 8 ? slot < 15 : slot < 3;
      base = a.vp + (rb * H + h) * 36 + 3 * (alive ? pnt : 0) + slot % 3;
      stride = (long)H * 36;
    }
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float Vr[DEPTH];
#pragma unroll
    for (int m = 0; m < DEPTH; ++m) {
      const int key = 2 * m + hi;
      Vr[m] = base[(long)(key < N ? key : N - 1) * stride];
    }
    for (int m0 = 0; m0 < nm; m0 += DEPTH) {
      float Vc[DEPTH];
#pragma unroll
      for (int m = 0; m < DEPTH; ++m) Vc[m] = Vr[m];
      if (m0 + DEPTH < nm) {
#pragma unroll
        for (int m = 0; m < DEPTH; ++m) {
          const int key = 2 * (m0 + DEPTH + m) + hi;
          Vr[m] = base[(long)(key < N ? key : N - 1) * stride];
        }
      }
#pragma unroll
      for (int m = 0; m < DEPTH; ++m)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(alive ? Vc[m] : 0.f, Ps[(2 * (m0 + m) + hi) * 32 + li], acc, 0, 0, 0);
    }
    if (qi >= N) continue;
    if (T < 8) {
      float* orow = a.out + (rb + qi) * a.out_ld + (long)h * AF_C + 32 * T + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *(f32x4*)(orow + 8 * g) = o;
      }
    } else {
      const float* R = a.rot + (rb + qi) * 9;
      const float* Tr = a.trans + (rb + qi) * 3;
      const int HP = H * 12, npts = T == 8 ? 5 : 1;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        if (k < npts) {
          const int pt = T == 8 ? 5 * hi + k : 10 + hi;
          const float x = acc[3 * k] - Tr[0], y = acc[3 * k + 1] - Tr[1], z = acc[3 * k + 2] - Tr[2];
          // invert_apply: R^T (p - t)   (openfold/utils/rigid_utils.py:1118-1130)
          const float ox = R[0] * x + R[3] * y + R[6] * z;
          const float oy = R[1] * x + R[4] * y + R[7] * z;
          const float oz = R[2] * x + R[5] * y + R[8] * z;
          float* o = a.out + (rb + qi) * a.out_ld + a.pt_off + h * 12 + pt;
          o[0] = ox; o[HP] = oy; o[2 * HP] = oz;
          o[3 * HP] = sqrtf(ox * ox + oy * oy + oz * oz + 1e-8f);
        }
      }
    }
  }
}

int fd_ipa_attention_f32_supported(const AttnArgs& a) {
  return a.C == AF_C && a.Dv == AF_C && a.Pq == 8 && a.Pv == 12 && a.N >= 1 && a.N <= 1024 && a.H >= 1 && !(a.q_ld & 3) && !(a.k_ld & 3) &&
         !(a.q_hs & 3) && !(a.k_hs & 3) && !(a.out_ld & 3) && a.bias && a.probs && a.qp && a.kp && a.vp && a.gamma && a.rot && a.trans &&
         a.res_mask;
}
int fd_ipa_attention_f32(const AttnArgs& a, hipStream_t st) {
  if (!fd_ipa_attention_f32_supported(a)) return FDIPT_EINVAL;
  const int nt = (a.N + 31) / 32, groups = a.B * nt, per = (groups + 7) / 8;
  const size_t smem = (size_t)(((AF_C + 26) * 32 > 32 * nt * 32 ? (AF_C + 26) * 32 : 32 * nt * 32) + 256 + 32 * nt) * 4;
  if (smem > 160 * 1024) return FDIPT_ESIZE;
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)ipa_attn_f32_kernel<3, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
        hipFuncSetAttribute((const void*)ipa_attn_f32_kernel<8, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  const dim3 grid(8 * per * a.H), block(FD_THREADS);
  if (a.N <= 3 * 4 * 32) hipLaunchKernelGGL((ipa_attn_f32_kernel<3, 2>), grid, block, smem, st, a);
  else hipLaunchKernelGGL((ipa_attn_f32_kernel<8, 1>), grid, block, smem, st, a);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
