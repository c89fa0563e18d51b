// attention2.hip — bf16 attention with scores held in REGISTERS (reference widths, N <= 480 keys).
//
//   attn2_kernel<IPA=true , C=256, DV=256> : InvariantPointAttention core, framedipt/model/ipa_pytorch.py:251-313
//   attn2_kernel<IPA=false, C=80 , DV=80 > : sequence-transformer attention, ipa_pytorch.py:433-443,536-538
//   pair_bias2_kernel                      : linear_b(z)/sqrt(3) (ipa_pytorch.py:247,256-257) streamed once over z
//
// Same transposed-MFMA scheme as edge_transition2.hip: a wave owns 32 QUERIES (lane = query), scores are computed as
// S^T[key, query] = K_tile[32 keys x C] * Q^T (K from LDS as the A operand, Q fragments in registers as B), so one
// lane holds the scores of its query against half of the keys (the other half lives in lane^32): softmax is pure
// register math + one cross-half shuffle, and the C/D fragments of S are directly the B fragments of the P.V
// product O^T[d, query] = V^T[d, keys] * P^T (V^T staged through LDS with the matching 16-wise key permutation).
// Point distances and the o_pt sum run on the VALU in fp32 against broadcast LDS reads of k_pts / v_pts.
#include "common.hpp"
#include "kernels.hpp"

#define A2_NT 15        // key tiles of 32 held in registers  -> N <= 480
#define A2_NMAX (32 * A2_NT)

__device__ __forceinline__ int a2_perm16(int pos) {  // involution: swaps the two middle groups of four
  const int hi = pos >> 3, e = pos & 7;
  return 4 * hi + (e & 3) + 8 * (e >> 2);
}
__device__ __forceinline__ bf16x8 a2_pack8(const float* v) {
  bf16x8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (__bf16)v[e];
  return o;
}
__device__ __forceinline__ bf16x8 a2_frag(const bf16_t* p) { return __builtin_bit_cast(bf16x8, *(const u16x8*)p); }

template <bool IPA, int C, int DV, int NT>
__global__ __launch_bounds__(FD_THREADS, 1) void attn2_kernel(AttnArgs a) {
  constexpr int NMAX = 32 * NT;
  constexpr int KS = C / 16;             // k-steps of the QK^T product
  constexpr int LDK = C + 8;             // K tile row stride (bf16 elements): 16-byte rows, conflict-free b128 reads
  constexpr int NDT = (DV + 31) / 32;    // 32-row d tiles of the output
  constexpr int P3 = 24, V3 = 36;        // Pq*3, Pv*3 of the reference IPA (8 / 12 points)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.N, nt = (N + 31) / 32, Np = nt * 32, LDV = Np + 8;
  // LDS carve: [K double buffer | k_pts | v_pts]; the V^T double buffer aliases the start after the score phase
  bf16_t* Ksm = (bf16_t*)smem;                               // [2][32][LDK]
  float* kps = (float*)(smem + 2 * 32 * LDK * 2);            // [Np][24]
  float* vps = kps + (IPA ? Np * P3 : 0);                    // [Np][36]
  bf16_t* Vsm = (bf16_t*)smem;                               // [2][32][LDV]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  const int h = blockIdx.y, b = blockIdx.z;
  const long rb = (long)b * N;
  const int i_raw = blockIdx.x * 128 + wave * 32 + li;
  const bool valid = i_raw < N;
  const int i = valid ? i_raw : N - 1;
  const float* qrow = a.q + (rb + i) * a.q_ld + (long)h * a.q_hs;
  const float* kbase = a.k + rb * a.k_ld + (long)h * a.k_hs;
  const float* vbase = a.v + rb * a.v_ld + (long)h * a.v_hs;

  float qp[P3];
  float gam = 0.f, mi = 1.f;
  if constexpr (IPA) {
#pragma unroll
    for (int c = 0; c < P3; ++c) qp[c] = a.qp[((rb + i) * a.H + h) * P3 + c];
    gam = -0.5f * a.gamma[h];
    mi = a.res_mask[rb + i];
    {  // k_pts / v_pts of this head -> LDS (batched 16-byte loads, then stores)
      constexpr int NK = (NMAX * (P3 / 4) + FD_THREADS - 1) / FD_THREADS, NVV = (NMAX * (V3 / 4) + FD_THREADS - 1) / FD_THREADS;
      f32x4 tk[NK], tv[NVV];
#pragma unroll
      for (int u = 0; u < NK; ++u) {
        const int v = tid + u * FD_THREADS, j = v / (P3 / 4), c = (v % (P3 / 4)) * 4;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (j < N) x = *(const f32x4*)(a.kp + ((rb + j) * a.H + h) * P3 + c);
        tk[u] = x;
      }
#pragma unroll
      for (int u = 0; u < NVV; ++u) {
        const int v = tid + u * FD_THREADS, j = v / (V3 / 4), c = (v % (V3 / 4)) * 4;
        f32x4 x = {0.f, 0.f, 0.f, 0.f};
        if (j < N) x = *(const f32x4*)(a.vp + ((rb + j) * a.H + h) * V3 + c);
        tv[u] = x;
      }
#pragma unroll
      for (int u = 0; u < NK; ++u) {
        const int v = tid + u * FD_THREADS;
        if (v < Np * (P3 / 4)) *(f32x4*)(kps + v * 4) = tk[u];
      }
#pragma unroll
      for (int u = 0; u < NVV; ++u) {
        const int v = tid + u * FD_THREADS;
        if (v < Np * (V3 / 4)) *(f32x4*)(vps + v * 4) = tv[u];
      }
    }
  }
  // ---- Q fragments (B operand), pre-scaled
  bf16x8 Qf[KS];
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const f32x4 u0 = *(const f32x4*)(qrow + 16 * s + 8 * hi), u1 = *(const f32x4*)(qrow + 16 * s + 8 * hi + 4);
    const float v[8] = {u0[0] * a.scale, u0[1] * a.scale, u0[2] * a.scale, u0[3] * a.scale,
                        u1[0] * a.scale, u1[1] * a.scale, u1[2] * a.scale, u1[3] * a.scale};
    Qf[s] = a2_pack8(v);
  }
  // ---- K tile staging: 32 keys x C fp32 -> bf16 [32][LDK]; register prefetch of the next tile
  constexpr int KV = (32 * C / 4 + FD_THREADS - 1) / FD_THREADS;  // float4 per thread per tile
  f32x4 kreg[KV];
  auto k_load = [&](int t) {
#pragma unroll
    for (int u = 0; u < KV; ++u) {
      const int v = tid + u * FD_THREADS, r = v / (C / 4), c4 = (v % (C / 4)) * 4, j = 32 * t + r;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (v < 32 * C / 4 && j < N) x = *(const f32x4*)(kbase + (long)j * a.k_ld + c4);
      kreg[u] = x;
    }
  };
  auto k_store = [&](int bufi) {
    bf16_t* dst = Ksm + bufi * 32 * LDK;
#pragma unroll
    for (int u = 0; u < KV; ++u) {
      const int v = tid + u * FD_THREADS, r = v / (C / 4), c4 = (v % (C / 4)) * 4;
      u16x4 hh = {f2bf(kreg[u][0]), f2bf(kreg[u][1]), f2bf(kreg[u][2]), f2bf(kreg[u][3])};
      if (v < 32 * C / 4) *(u16x4*)(dst + r * LDK + c4) = hh;
    }
  };
  k_load(0);
  k_store(0);
  __syncthreads();

  // ---- phase 1: scores S[t][r] for key j = 32t + (r&3) + 8(r>>2) + 4hi
  f32x16 S[NT];
  const float* brow = IPA ? a.bias + (((long)b * a.H + h) * N + i) * N : nullptr;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (t < nt) {
      if (t + 1 < nt) k_load(t + 1);
      const bf16_t* Kt = Ksm + (t & 1) * 32 * LDK + li * LDK + 8 * hi;
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < KS; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2_frag(Kt + 16 * s), Qf[s], acc, 0, 0, 0);
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int j0 = 32 * t + 8 * g + 4 * hi;
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if constexpr (IPA) {
          if (j0 + 3 < N && (N & 3) == 0) bv = *(const f32x4*)(brow + j0);
          else {
#pragma unroll
            for (int q = 0; q < 4; ++q) bv[q] = j0 + q < N ? brow[j0 + q] : 0.f;
          }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int j = j0 + q;
          float sv = acc[4 * g + q];
          if constexpr (IPA) {
            sv += bv[q];
            const float* kpj = kps + j * P3;
            float d2 = 0.f;
#pragma unroll
            for (int c = 0; c < P3; ++c) {
              const float d = qp[c] - kpj[c];
              d2 += d * d;
            }
            sv += gam * d2;
            const float mj = j < N ? a.res_mask[rb + j] : 0.f;
            sv += 1e5f * (mi * mj - 1.f);
          } else {
            const float mj = j < N ? a.res_mask[rb + j] : 0.f;
            if (mj == 0.f) sv = -1e30f;
          }
          if (j >= N) sv = -1e30f;
          acc[4 * g + q] = sv;
        }
        __builtin_amdgcn_sched_barrier(0);  // bound the live range of the broadcast k_pts reads (register pressure)
      }
      S[t] = acc;
      if (t + 1 < nt) k_store((t + 1) & 1);
    }
    __syncthreads();  // unconditional: keeps the tile loop fully unrollable (S[t] must stay in registers)
  }
  // ---- phase 2: softmax over keys (this lane's half + lane^32)
  float mx = -3.0e38f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[t][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = expf(S[t][r] - mx);
        S[t][r] = e;
        sum += e;
      }
  sum += __shfl_xor(sum, 32, 64);
  const float inv = 1.0f / sum;
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) S[t][r] *= inv;

  if constexpr (IPA) {
    // attention weights for the o_pair kernel: probs[b,h,i,j]
    if (valid) {
      float* prow = a.probs + (((long)b * a.H + h) * N + i) * N;
#pragma unroll
      for (int t = 0; t < NT; ++t)
        if (t < nt)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int j0 = 32 * t + 8 * g + 4 * hi;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (j0 + q < N) prow[j0 + q] = S[t][4 * g + q];
          }
    }
    // ---- phase 3: o_pt = R_i^T (sum_j p v_pts_j - t_i)  (fp32; v_pts rows broadcast from LDS)
    float op[V3];
#pragma unroll
    for (int c = 0; c < V3; ++c) op[c] = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (t < nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int j = 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi;
          const float p = S[t][r];
          const float* vj = vps + j * V3;
#pragma unroll
          for (int c = 0; c < V3; ++c) op[c] += p * vj[c];
          if ((r & 1) == 1) __builtin_amdgcn_sched_barrier(0);  // keep at most two v_pts rows in flight
        }
#pragma unroll
    for (int c = 0; c < V3; ++c) op[c] += __shfl_xor(op[c], 32, 64);
    if (valid) {
      const float* R = a.rot + (rb + i) * 9;
      const float* T = a.trans + (rb + i) * 3;
      const int HP = a.H * 12;
      float* o = a.out + (rb + i) * a.out_ld + a.pt_off + h * 12;
#pragma unroll
      for (int pt = 0; pt < 12; ++pt) {
        if ((pt < 6) == (hi == 0)) {  // the two half-waves split the 12 points
          const float x = op[pt * 3] - T[0], y = op[pt * 3 + 1] - T[1], z = op[pt * 3 + 2] - T[2];
          const float ox = R[0] * x + R[3] * y + R[6] * z;
          const float oy = R[1] * x + R[4] * y + R[7] * z;
          const float oz = R[2] * x + R[5] * y + R[8] * z;
          o[pt] = ox; o[HP + pt] = oy; o[2 * HP + pt] = oz;
          o[3 * HP + pt] = sqrtf(ox * ox + oy * oy + oz * oz + 1e-8f);
        }
      }
    }
  }
  // ---- phase 4: P fragments (B operand, permuted key order inside every 16-group)
  bf16x8 Pf[2 * NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
    if (t < nt) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = S[t][r];
      Pf[2 * t] = a2_pack8(v);
      Pf[2 * t + 1] = a2_pack8(v + 8);
    }
  __syncthreads();  // every wave is done with Ksm / kps / vps: the V^T buffers may overwrite them
  // ---- phase 5: O^T[d, query] = V^T[d, keys] P^T, one 32-row d tile at a time
  // staging: V[key][d0 .. d0+31] fp32 -> Vsm[d][(key & ~15) + perm16(key & 15)] bf16
  constexpr int NVS = NMAX * 8 / FD_THREADS;  // float4 per thread per d tile (upper bound)
  f32x4 vreg[NVS];
  auto v_load = [&](int dt) {
#pragma unroll
    for (int u = 0; u < NVS; ++u) {
      const int v = tid + u * FD_THREADS, jj = v >> 3, d = 32 * dt + (v & 7) * 4;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (jj < N && d < DV) x = *(const f32x4*)(vbase + (long)jj * a.v_ld + d);
      vreg[u] = x;
    }
  };
  auto v_store = [&](int bufi) {
    bf16_t* dst = Vsm + bufi * 32 * LDV;
#pragma unroll
    for (int u = 0; u < NVS; ++u) {
      const int v = tid + u * FD_THREADS, jj = v >> 3, d4 = (v & 7) * 4;
      if (jj < Np) {
        const int col = (jj & ~15) + a2_perm16(jj & 15);
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[(d4 + q) * LDV + col] = f2bf(vreg[u][q]);
      }
    }
  };
  v_load(0);
  v_store(0);
  __syncthreads();
#pragma unroll 1
  for (int dt = 0; dt < NDT; ++dt) {
    if (dt + 1 < NDT) v_load(dt + 1);
    const bf16_t* Vt = Vsm + (dt & 1) * 32 * LDV + li * LDV + 8 * hi;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 2 * NT; ++s)
      if (s < 2 * nt) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2_frag(Vt + 16 * s), Pf[s], acc, 0, 0, 0);
    if (valid) {
      float* orow = a.out + (rb + i) * a.out_ld + (long)h * DV + 32 * dt + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (32 * dt + 8 * g + 4 * hi < DV) {
          f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
          *(f32x4*)(orow + 8 * g) = o;
        }
      }
    }
    if (dt + 1 < NDT) v_store((dt + 1) & 1);
    __syncthreads();
  }
}

template <bool IPA, int C, int DV, int NT>
static int launch_attn2(const AttnArgs& a, hipStream_t st) {
  const int nt = (a.N + 31) / 32, Np = nt * 32;
  const size_t s1 = (size_t)2 * 32 * (C + 8) * 2 + (IPA ? (size_t)Np * (24 + 36) * 4 : 0);
  const size_t s2 = (size_t)2 * 32 * (Np + 8) * 2;
  const size_t smem = (s1 > s2 ? s1 : s2) + 16;
  if (smem > 160 * 1024) return FDIPT_ESIZE;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)attn2_kernel<IPA, C, DV, NT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL((attn2_kernel<IPA, C, DV, NT>), dim3(cdiv(a.N, 128), a.H, a.B), dim3(FD_THREADS), smem, st, a);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// returns 1 if the register kernel covers this shape (caller falls back to attention.hip otherwise)
int fd_attention2_supported(int ipa, const AttnArgs& a) {
  if (a.N > A2_NMAX || a.N < 1) return 0;
  if (ipa) return a.C == 256 && a.Dv == 256 && a.Pq == 8 && a.Pv == 12;
  return a.C == 80 && a.Dv == 80;
}
int fd_attention2(int ipa, const AttnArgs& a, hipStream_t st) {
  if (a.N <= 320) return ipa ? launch_attn2<true, 256, 256, 10>(a, st) : launch_attn2<false, 80, 80, 10>(a, st);
  return ipa ? launch_attn2<true, 256, 256, A2_NT>(a, st) : launch_attn2<false, 80, 80, A2_NT>(a, st);
}

// ------------------------------------------------------------------ pair bias
// bias[b,h,i,j] = sqrt(1/3) (Wb z[b,i,j,:] + bb)_h : lane = pair, heads = MFMA rows (8 of 32 used), z fragments loaded
// straight from HBM in B-operand layout (no LDS), Wb fragments in registers.  HBM-bound: one pass over z.
__global__ __launch_bounds__(FD_THREADS) void pair_bias2_kernel(int B, int N, int H, const bf16_t* __restrict__ z,
                                                                const bf16_t* __restrict__ wb /* [H,128] pre-scaled */,
                                                                const float* __restrict__ bb, float* __restrict__ out,
                                                                int frag /* 1: fd_bias_frag_off order (attention3) */) {
  const int lane = threadIdx.x & 63, hi = lane >> 5, li = lane & 31;
  const long NN = (long)N * N, n_pairs = (long)B * NN;
  bf16x8 Wf[8];
#pragma unroll
  for (int s = 0; s < 8; ++s) {
    u16x8 w = {0, 0, 0, 0, 0, 0, 0, 0};
    if (li < H) w = *(const u16x8*)(wb + li * 128 + 16 * s + 8 * hi);
    Wf[s] = __builtin_bit_cast(bf16x8, w);
  }
  const long n_tiles = (n_pairs + 31) / 32;
  for (long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6); tile < n_tiles; tile += (long)gridDim.x * 4) {
    const long p_raw = tile * 32 + li;
    const long p = p_raw < n_pairs ? p_raw : n_pairs - 1;
    const bf16_t* zr = z + p * 128 + 8 * hi;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    bf16x8 zf[8];
#pragma unroll
    for (int s = 0; s < 8; ++s) zf[s] = __builtin_bit_cast(bf16x8, *(const u16x8*)(zr + 16 * s));
#pragma unroll
    for (int s = 0; s < 8; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Wf[s], zf[s], acc, 0, 0, 0);
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
  hipLaunchKernelGGL(pair_bias2_kernel, dim3(grid), dim3(FD_THREADS), 0, st, B, N, H, (const bf16_t*)z, (const bf16_t*)wb, bb,
                     out, frag);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
