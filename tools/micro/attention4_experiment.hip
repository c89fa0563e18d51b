// EXPERIMENT (round 3), NOT part of the library: measured 66 - 74 us per call at N = 300, B = 8 against 45 us of ipa_attn3_kernel
// (tools/micro/attn4_bench.hip; phase profile: per key tile pass 1 = 4.4 k cycles for 1.4 k matrix cycles, pass 2 = 8.1 k for 3.3 k).
// One wave per SIMD (512 registers: 176 accumulators + the Q fragments) leaves nothing to overlap the DMA issue (60 - 70 cycles per
// instruction), the softmax VALU work and the LDS waits with; the traffic saved (96 MB instead of 230 MB per call) never was the
// bound.  Results are bit-identical in structure to the library kernel's split P V (110 GPU tests passed with it in the forward).
//
// attention4.hip — invariant point attention core (framedipt/model/ipa_pytorch.py:251-313) for N <= 384, half-precision mode,
// reference widths (C = 256, 8 q/k points, 12 v points, H <= 8).  Third generation, built around two measurements of round 2:
// ipa_attn3_kernel (32 queries x one (sample, head) per block, keys dealt to the four waves) pulls every K and V fragment of its
// (sample, head) through the CU's 64 B/clk L2 port once per 32 queries — 230 MB per call at N = 300, B = 8, MFMA utilisation 9.7 %;
// and the rounding of the attention weights and of V to half precision is the largest error source of the half mode that is left
// once the per-residue products run on split operands (tests/err_budget.py at bb_gain 0.3).
//
//   * A WAVE owns a query tile (32 queries) and walks ALL key tiles; a block is up to four query tiles of one (sample, head), so
//     every K / V fragment is fetched once per 128 queries, by LDS-DMA (global_load_lds_dwordx4) of the fragment-order images
//     ipa_proj2_kernel writes.  EVERY per-tile operand (K fragments, key points, pair-bias tile, key mask, V_hi, V_lo, value
//     points) travels by DMA, so the only vector-memory loads inside the loops are the kernel's own and vmcnt is waited for by count.
//   * Two passes over the key tiles, both ROLLED loops (a first, fully unrolled two-phase version that kept all logits in
//     registers was 100 KB of straight-line code executed once per wave: instruction fetch at ~1 B / cycle made it 58 us per call).
//     Pass 1: logits of a key tile -> running row maximum and sum only.  Pass 2: the same logits again -> normalised weights ->
//     (a) half rows for the o_pair kernel, (b) the B fragments of O^T[d, query] += V^T P for all eight 32-channel tiles and the
//     three value-point tiles.  The weights never leave the registers: the C/D layout of S^T[key, query] IS the B-operand layout
//     for the key-permuted V^T image (perm16), a lane's own 16 weights of a key tile are its two B fragments.
//     Logits: mask product and point term as fp32 MFMAs (exact fp32 FMA chains: q.k - |k|^2/2 - |q|^2/2 has 3-4 digits of
//     cancellation), Q K^T as 16 fp16 MFMAs, pair bias from its tiled layout — bit-identical in both passes.
//   * SPLIT: P = hi + lo and V = hi + lo (each part one half-precision value; V_lo image written by the projection):
//     V_hi P_hi + V_hi P_lo + V_lo P_hi with fp32 accumulation — the accuracy of an fp32 product; the value points already
//     are a hi / lo image, they get the P_lo term.
// Per call at N = 300, B = 8: 192 blocks; ~47 k matrix cycles per wave (pass 1 14 k, pass 2 33 k).
#include <type_traits>

#include "../../framedipt_amd/csrc/common.hpp"
#include "../../framedipt_amd/csrc/kernels.hpp"

#define A4_C 256
// key part of a slot : K fragments 16 KB | key points [32 keys][24] f32 (4 KB with padding) | pair-bias tile of each wave's query
//                      tile 4 x 4 KB | key mask 4 x 256 B
// value part         : V_hi 16 KB | V_lo 16 KB | value-point image 6 KB
#define A4_K_KP 16384
#define A4_K_BIAS 20480
#define A4_K_MASK 36864
#define A4_KPART 37888
#define A4_V_LO 16384
#define A4_V_PT 32768
#define A4_VPART 38912
#define A4_SLOT2 (A4_KPART + A4_VPART)   // pass 2: key part + value part, ring of 2 (one tile ahead: a step is ~3.3 k matrix cycles)
#define A4_RING1 4                       // pass 1: key parts only, ring of 4 (three tiles ahead: a step is ~1.4 k matrix cycles)
#define A4_LDS (2 * A4_SLOT2)            // 153600 >= A4_RING1 * A4_KPART = 151552
#define A4_KREQ 10                       // DMA instructions per wave and key part

typedef const __attribute__((address_space(3))) u16x8* a4_lds_u16x8;
typedef const __attribute__((address_space(3))) f32x4* a4_lds_f32x4;
typedef const __attribute__((address_space(3))) float* a4_lds_f32;
__device__ __forceinline__ hx8 a4_frag(unsigned off) { return __builtin_bit_cast(hx8, *(a4_lds_u16x8)(unsigned long)off); }
__device__ __forceinline__ f32x4 a4_ldsf4(unsigned off) { return *(a4_lds_f32x4)(unsigned long)off; }
// 64 lanes x 16 B -> 1 KB of LDS at `lds_dst` (wave-uniform), lane l at + 16 l
__device__ __forceinline__ void a4_dma16(const void* gsrc, unsigned lds_dst) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(gsrc) : "memory", "m0");
}
__device__ __forceinline__ void a4_dma4(const void* gsrc, unsigned lds_dst) {  // 64 lanes x 4 B
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dword %1, off" : : "s"(m0v), "v"(gsrc) : "memory", "m0");
}
// s_waitcnt vmcnt(n) for the counts the pipeline needs (the immediate must be a constant; n is wave-uniform)
__device__ __forceinline__ void a4_vm_wait(int n) {
  __builtin_amdgcn_sched_barrier(0);
  if (n >= 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
  else if (n >= 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ void a4_split8(const float* v, hx8& hi, hx8& lo) {
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    hi[e] = (fd_h)v[e];
    lo[e] = (fd_h)(v[e] - (float)hi[e]);
  }
}

// SPLIT: see above (needs Attn3Args.Vt_lo).
template <bool SPLIT>
__global__ __launch_bounds__(FD_THREADS, 1) void ipa_attn4_kernel(Attn3Args a, int nblk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  const int N = a.N, H = a.H, nt = (N + 31) / 32, ks = 2 * nt;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), hi = lane >> 5, li = lane & 31;
  // XCD-aware block -> (sample, head, block of query tiles): consecutive workgroup ids go round-robin over the 8 XCDs; all blocks
  // of one (sample, head) run on the SAME XCD, whose L2 then holds its K / V images once
  const int BH = a.B * H;
  int bhq, jb;
  {
    const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
    const int per = (BH + 7) >> 3;
    bhq = xcd * per + local / nblk;
    jb = local % nblk;
    if (local >= per * nblk || bhq >= BH) return;
  }
  // query tiles of this block: nt tiles dealt to nblk blocks as evenly as possible, one per wave
  const int base = nt / nblk, rem = nt % nblk;
  const int q0 = jb * base + (jb < rem ? jb : rem), nq = base + (jb < rem ? 1 : 0);
  const bool active = wave < nq;                 // waves without a query tile only move data
  const int qt = q0 + (active ? wave : 0);
  const int h = bhq % H, b = bhq / H;
  const long rb = (long)b * N, bh = bhq;
  const int i_raw = qt * 32 + li;
  const bool valid = active && i_raw < N;
  const int i = i_raw < N ? i_raw : N - 1;

  // ---- data movement
  const half_t* Kimg = a.Kb + (bh * nt) * (16 * 512);
  const float* bias_q = a.bias + ((bh * nt + qt) * (long)nt) * 1024;  // fd_bias_frag_off: tiles [key tile][32 queries][32 keys] of this query tile
  auto request_k = [&](int t, unsigned dst) {  // A4_KREQ instructions per wave
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = wave + 4 * u;
      a4_dma16(Kimg + ((long)t * 16 + f) * 512 + lane * 8, dst + f * 1024);
    }
    {  // key points: 16 B chunk c = 6 key + part of the tile's [32][24] floats (chunks 192..255: padding, clamped source)
      const int c = wave * 64 + lane, key = c < 192 ? c / 6 : 31, part = c < 192 ? c - 6 * key : 5;
      int j = 32 * t + key;
      if (j > N - 1) j = N - 1;
      a4_dma16(a.kp + ((rb + j) * H + h) * 24 + 4 * part, dst + A4_K_KP + wave * 1024);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)  // this wave's query tile: 4 KB, row i = 32 floats
      a4_dma16(bias_q + (long)t * 1024 + u * 256 + lane * 4, dst + A4_K_BIAS + wave * 4096 + u * 1024);
    {
      int j = 32 * t + li;
      if (j > N - 1) j = N - 1;
      a4_dma4(a.res_mask + rb + j, dst + A4_K_MASK + wave * 256);
    }
  };
  const half_t* Vimg = a.Vt + (bh * (A4_C / 32)) * ((long)ks * 512);
  const half_t* Vlimg = SPLIT ? a.Vt_lo + (bh * (A4_C / 32)) * ((long)ks * 512) : nullptr;
  const half_t* Pimg = a.vpt + (bh * 3) * ((long)ks * 512);
  auto request_v = [&](int t, unsigned dst) {  // V_hi 16 (| V_lo 16) | points 6 fragments (k-steps 2t, 2t + 1)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f = wave + 4 * u, dt = f >> 1, s = f & 1;
      a4_dma16(Vimg + ((long)dt * ks + 2 * t + s) * 512 + lane * 8, dst + f * 1024);
      if constexpr (SPLIT) a4_dma16(Vlimg + ((long)dt * ks + 2 * t + s) * 512 + lane * 8, dst + A4_V_LO + f * 1024);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      const int f = wave + 4 * u;
      if (f < 6) a4_dma16(Pimg + ((long)(f >> 1) * ks + 2 * t + (f & 1)) * 512 + lane * 8, dst + A4_V_PT + f * 1024);
    }
  };

  FD_STAMP(0);
  // ---- query-side registers (plain loads: the compiler waits for them, and for everything older, before the first step)
  hx8 Qf[16];
  {
    const half_t* qr = a.Qb + ((bh * nt + qt) * 16 * 64 + lane) * 8;  // fragment order: 1 KB per k-step
#pragma unroll
    for (int s = 0; s < 16; ++s) Qf[s] = __builtin_bit_cast(hx8, *(const u16x8*)(qr + s * 512));
  }
  float mi = a.res_mask[rb + i];
  const float gam = a.gamma[h];
  float qB[13];  // B operand of the point product (k index 2s + hi), pre-multiplied by the head's point weight gamma
  {
    const float* qpr = a.qp + ((rb + i) * H + h) * 24;
    float qv[24];
#pragma unroll
    for (int c4 = 0; c4 < 6; ++c4) {
      const f32x4 x = *(const f32x4*)(qpr + 4 * c4);
      qv[4 * c4] = x[0]; qv[4 * c4 + 1] = x[1]; qv[4 * c4 + 2] = x[2]; qv[4 * c4 + 3] = x[3];
    }
    float qn = 0.f;
#pragma unroll
    for (int c = 0; c < 24; ++c) qn += qv[c] * qv[c];
#pragma unroll
    for (int s = 0; s < 12; ++s) qB[s] = gam * qv[2 * s + hi];
    qB[12] = gam * (hi ? -0.5f * qn : 1.0f);
  }
  // pin the query-side values here: their (compiler-placed) waits must not land behind the requests below
#pragma unroll
  for (int s = 0; s < 16; ++s) asm volatile("" : "+v"(Qf[s]));
#pragma unroll
  for (int s = 0; s < 13; ++s) asm volatile("" : "+v"(qB[s]));
  asm volatile("" : "+v"(mi));

  // logits of key tile t from the key part at LDS offset `sl` (identical arithmetic in both passes)
  auto logits = [&](int t, unsigned sl) {
    // ONE accumulator for the three terms: it starts at -1e5, the mask product 1e5 m_i m_j brings an unmasked pair back
    // to exactly 0 (a masked one stays at -1e5, a padded key goes to -1e30) BEFORE the small terms are added, so nothing
    // is lost to the large constant; then gamma * (q.k - |k|^2/2 - |q|^2/2) as a 26-deep fp32 MFMA chain, then Q K^T
    // every LDS operand of the tile is requested up front (one wave per SIMD: nothing else hides the LDS latency); the 16 K
    // fragments land under the fp32 chain
    const unsigned kb = sl + lane * 16;
    hx8 kf[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) kf[s] = a4_frag(kb + s * 1024);
    float mA = *(a4_lds_f32)(unsigned long)(sl + A4_K_MASK + wave * 256 + li * 4);
    float kpv[24];
#pragma unroll
    for (int c4 = 0; c4 < 6; ++c4) {
      const f32x4 x = a4_ldsf4(sl + A4_K_KP + li * 96 + 16 * c4);
      kpv[4 * c4] = x[0]; kpv[4 * c4 + 1] = x[1]; kpv[4 * c4 + 2] = x[2]; kpv[4 * c4 + 3] = x[3];
    }
    f32x4 bv[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) bv[g] = a4_ldsf4(sl + A4_K_BIAS + wave * 4096 + li * 128 + (8 * g + 4 * hi) * 4);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = -1e5f;
    if (32 * t + li >= N) mA = -1e25f;  // padded keys: marker so that the logit becomes -1e30
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? 0.f : mA, hi ? 0.f : 1e5f * mi, acc, 0, 0, 0);
    float kn = 0.f;
#pragma unroll
    for (int c = 0; c < 24; ++c) kn += kpv[c] * kpv[c];
#pragma unroll
    for (int s = 0; s < 12; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? kpv[2 * s + 1] : kpv[2 * s], qB[s], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(hi ? 1.0f : -0.5f * kn, qB[12], acc, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = fd_mfma32(kf[s], Qf[s], acc);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j0 = 32 * t + 8 * g + 4 * hi;
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[4 * g + q] += (j0 + q < N) ? bv[g][q] : 0.f;  // (padded keys: the slot is never written)
    }
    return acc;
  };

  // ---- pass 1: row maximum and sum of exp over all keys (online: the sum is rescaled when the maximum grows)
  FD_STAMP(1);
  float mx = -3.0e38f, sum = 0.f;
  for (int r = 0; r < 3 && r < nt; ++r) request_k(r, lds0 + r * A4_KPART);
  for (int t = 0; t < nt; ++t) {
    const int younger = nt - 1 - t;
    if (t == 4) FD_STAMP(5);
    a4_vm_wait((younger < 2 ? younger : 2) * A4_KREQ);  // tile t has landed (in-order counter: only the two younger requests may be out)
    if (t == 4) FD_STAMP(6);
    __syncthreads();
    if (t == 4) FD_STAMP(7);
    if (t + 3 < nt) request_k(t + 3, lds0 + ((t + 3) & 3) * A4_KPART);  // into the slot of tile t - 1, which every wave has left
    if (t == 4) FD_STAMP(8);
    {  // (waves without a query tile redo wave 0's, unstored: uniform control flow keeps the accumulators where they are)
      const f32x16 S = logits(t, lds0 + (t & 3) * A4_KPART);
      if (t == 4) FD_STAMP(9);
      float tm = S[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) tm = fmaxf(tm, S[r]);
      tm = fmaxf(tm, __shfl_xor(tm, 32, 64));
      const float mn = fmaxf(mx, tm);
      float ts = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) ts += __builtin_amdgcn_exp2f((S[r] - mn) * 1.4426950408889634f);
      ts += __shfl_xor(ts, 32, 64);
      sum = sum * __builtin_amdgcn_exp2f((mx - mn) * 1.4426950408889634f) + ts;
      mx = mn;
      if (t == 4) FD_STAMP(10);
    }
  }
  const float inv = 1.0f / sum;
  FD_STAMP(2);
  __syncthreads();  // every wave has left pass 1's ring: pass 2's slots overlay it
  request_k(0, lds0);
  request_v(0, lds0 + A4_KPART);

  // ---- pass 2: weights of a key tile -> HBM (half rows for o_pair) and -> B fragments of O^T[d, query] += V^T P
  f32x16 O[8], OP[3];
#pragma unroll
  for (int d = 0; d < 8; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) O[d][r] = 0.f;
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int r = 0; r < 16; ++r) OP[d][r] = 0.f;
  const long bq = bh / H;
  half_t* prow16 = a.probs_h16 + ((bq * N + i) * H + (bh - bq * H)) * (long)a.Np;
  for (int t = 0; t < nt; ++t) {
    if (t == 4) FD_STAMP(11);
    a4_vm_wait(0);  // tile t (requested a step ago) has landed; the weight rows stored in the last step have left
    __syncthreads();
    if (t == 4) FD_STAMP(12);
    if (t + 1 < nt) {
      request_k(t + 1, lds0 + ((t + 1) & 1) * A4_SLOT2);
      request_v(t + 1, lds0 + ((t + 1) & 1) * A4_SLOT2 + A4_KPART);
    }
    if (t == 4) FD_STAMP(13);
    {
      const unsigned sl = lds0 + (t & 1) * A4_SLOT2;
      const f32x16 S = logits(t, sl);
      if (t == 4) FD_STAMP(14);
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = __builtin_amdgcn_exp2f((S[r] - mx) * 1.4426950408889634f) * inv;
      hx8 ph[2], pl[2];
      if constexpr (SPLIT) {
        a4_split8(v, ph[0], pl[0]);
        a4_split8(v + 8, ph[1], pl[1]);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          ph[0][e] = (fd_h)v[e];
          ph[1][e] = (fd_h)v[8 + e];
        }
      }
      if (valid) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // (masked / padded keys carry exact zeros); the half values the o_pair kernel multiplies with
          const hx4 o = {ph[g >> 1][4 * (g & 1)], ph[g >> 1][4 * (g & 1) + 1], ph[g >> 1][4 * (g & 1) + 2], ph[g >> 1][4 * (g & 1) + 3]};
          *(hx4*)(prow16 + 32 * t + 8 * g + 4 * hi) = o;
        }
      }
      const unsigned vb = sl + A4_KPART + lane * 16;
      // the fragments of a whole k-step (16 keys: 8 + 8 + 3 fragments) are requested at once, the second k-step's under the
      // first one's MFMAs
      hx8 vh[2][8], vl[SPLIT ? 2 : 1][8], pf[2][3];
      auto v_load = [&](auto SC) {
        constexpr int s = decltype(SC)::value;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          vh[s][d] = a4_frag(vb + (2 * d + s) * 1024);
          if constexpr (SPLIT) vl[s][d] = a4_frag(vb + A4_V_LO + (2 * d + s) * 1024);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) pf[s][d] = a4_frag(vb + A4_V_PT + (2 * d + s) * 1024);
      };
      v_load(std::integral_constant<int, 0>{});
      v_load(std::integral_constant<int, 1>{});
#pragma unroll
      for (int s = 0; s < 2; ++s) {
#pragma unroll
        for (int d = 0; d < 8; ++d) {
          if constexpr (SPLIT) {
            O[d] = fd_mfma32(vl[s][d], ph[s], O[d]);
            O[d] = fd_mfma32(vh[s][d], pl[s], O[d]);
          }
          O[d] = fd_mfma32(vh[s][d], ph[s], O[d]);
        }
#pragma unroll
        for (int d = 0; d < 3; ++d) {  // value points: rows 0..35 high parts, 36..71 low parts (points16_kernel), 72..95 zero
          if constexpr (SPLIT)
            if (d < 2) OP[d] = fd_mfma32(pf[s][d], pl[s], OP[d]);  // (tile 2 holds low parts only: P_lo v_lo is below fp32 resolution)
          OP[d] = fd_mfma32(pf[s][d], ph[s], OP[d]);
        }
      }
      if (t == 4) FD_STAMP(15);
    }
  }
  FD_STAMP(3);
  // ---- o rows: lane = query, 4-runs of channels
#pragma unroll
  for (int d = 0; d < 8; ++d) {
    if (valid && a.out_h16) {
      half_t* orow = a.out_h16 + (rb + i) * a.out_ld + (long)h * A4_C + 32 * d + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const u16x4 o = {f2h(O[d][4 * g]), f2h(O[d][4 * g + 1]), f2h(O[d][4 * g + 2]), f2h(O[d][4 * g + 3])};
        *(u16x4*)(orow + 8 * g) = o;
      }
    } else if (valid) {
      float* orow = a.out + (rb + i) * a.out_ld + (long)h * A4_C + 32 * d + 4 * hi;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 o = {O[d][4 * g], O[d][4 * g + 1], O[d][4 * g + 2], O[d][4 * g + 3]};
        *(f32x4*)(orow + 8 * g) = o;
      }
    }
  }
  // ---- o_pt = R_i^T (sum - t_i) and its norm (ipa_pytorch.py:296-308): the wave's 32 x 96 sums through its own LDS tile
  __syncthreads();  // every wave is done with the ring
  float* opr = (float*)smem + wave * (32 * 96);  // [32 queries][96]: 12 KB per wave
#pragma unroll
  for (int d = 0; d < 3; ++d)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4 o = {OP[d][4 * g], OP[d][4 * g + 1], OP[d][4 * g + 2], OP[d][4 * g + 3]};
      *(f32x4*)(opr + li * 96 + 32 * d + 8 * g + 4 * hi) = o;
    }
  __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave reads back only what it wrote itself
  if (active && i_raw < N) {  // lane = (query li, points 6 hi .. 6 hi + 5): the frame is read once per lane
    float R[9], Tr[3];
#pragma unroll
    for (int k = 0; k < 9; ++k) R[k] = a.rot[(rb + i) * 9 + k];
#pragma unroll
    for (int k = 0; k < 3; ++k) Tr[k] = a.trans[(rb + i) * 3 + k];
    const int HP = H * 12;
#pragma unroll
    for (int pp = 0; pp < 6; ++pp) {
      const int pt = 6 * hi + pp;
      const float* o = opr + li * 96 + pt * 3;
      const float sx = o[0] + o[36], sy = o[1] + o[37], sz = o[2] + o[38];
      const float x = sx - Tr[0], y = sy - Tr[1], z = sz - Tr[2];
      const float ox = R[0] * x + R[3] * y + R[6] * z;
      const float oy = R[1] * x + R[4] * y + R[7] * z;
      const float oz = R[2] * x + R[5] * y + R[8] * z;
      const float on = sqrtf(ox * ox + oy * oy + oz * oz + 1e-8f);
      if (a.out_h16) {
        half_t* oo = a.out_h16 + (rb + i) * a.out_ld + a.pt_off + h * 12 + pt;
        oo[0] = f2h(ox); oo[HP] = f2h(oy); oo[2 * HP] = f2h(oz); oo[3 * HP] = f2h(on);
      } else {
        float* oo = a.out + (rb + i) * a.out_ld + a.pt_off + h * 12 + pt;
        oo[0] = ox; oo[HP] = oy; oo[2 * HP] = oz;
        oo[3 * HP] = on;
      }
    }
  }
  FD_STAMP(4);
}

int fd_attention4_supported(const Attn3Args& a) {
  return a.N >= 1 && a.N <= 384 && a.H <= 8 && a.Np == ((a.N + 31) / 32) * 32 && a.vpt != nullptr && a.probs_h16 != nullptr;
}

int fd_attention4(const Attn3Args& a, hipStream_t st) {
  if (!fd_attention4_supported(a)) return FDIPT_ESIZE;
  const int nt = (a.N + 31) / 32, nblk = (nt + 3) / 4;
  const int per = (a.B * a.H + 7) / 8;
  const dim3 grid(8 * per * nblk), block(FD_THREADS);
  if (a.Vt_lo) {
    if (hipFuncSetAttribute((const void*)ipa_attn4_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, A4_LDS) != hipSuccess) return FDIPT_ELAUNCH;
    hipLaunchKernelGGL((ipa_attn4_kernel<true>), grid, block, A4_LDS, st, a, nblk);
  } else {
    if (hipFuncSetAttribute((const void*)ipa_attn4_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, A4_LDS) != hipSuccess) return FDIPT_ELAUNCH;
    hipLaunchKernelGGL((ipa_attn4_kernel<false>), grid, block, A4_LDS, st, a, nblk);
  }
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
