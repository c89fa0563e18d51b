// attention3.hip — invariant point attention core (framedipt/model/ipa_pytorch.py:251-313), bf16, reference widths
// (C = 256, 8 q/k points, 12 v points, H <= 8), N <= 512.
//
// One block = 32 queries of one (batch, head); the KEYS are dealt round-robin to the block's 4 waves in tiles of 32,
// so every lane holds only N/8 scores.  Operands come pre-formatted from ipa_proj_kernel (gemm.hip):
//   scalar logits    S^T[key, query] = Kb_tile * Qb^T            bf16 MFMA, A (K fragments, 1 KB linear loads) straight
//                    from HBM/L2, B = Q regs
//   point logits     -1/2 |q_pt - k_pt|^2 = q.k - |k|^2/2 - |q|^2/2; the last term is constant along a softmax row and dropped, the
//                    rest runs on fp16 hi / lo parts as five fp16 MFMAs (2^-22 relative): A = the key-point fragment image
//                    points16_kernel writes (kernels.hpp: fd_kpf layout; linear 1 KB loads), B = query points in registers
//   mask             1e5 m_i m_j as one more fp16 MFMA (exact: 2 m_j * 32768 m_i + m_j * 34464 m_i; padded keys carry a marker)
//   softmax          registers + lane^32 shuffle, cross-wave max / sum through 2 x 128 floats of LDS
//   o = a v          P fragments are exchanged through LDS once; wave w then owns d tiles {2w, 2w+1} over ALL keys,
//                    A = Vt rows (V transposed, key-permuted) straight from HBM/L2 — no LDS staging, no reduction
//   o_pt             one more 32-row tile of the same P.V product: v_pts as bf16 high + low parts (points_kernel writes the
//                    fragment image), fp32 accumulate; rotated into the local frame from a 32 x 96 LDS tile
// LDS holds only the P fragments (N x 64 bf16), the o_pt tile and the small reduction buffers.
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

#define A3_C 256
// (dev) ablations for the concurrency bisection of tools/hazard_lib_repro.py: FDIPT_VARIANT builds with -DA3_ABL=bits —
//   1: softmax without the transcendental unit (no v_exp_f32)   2: no logit MFMAs   4: no P V / value-point MFMAs (phase 4 skipped)
//   8: no LDS Q fragments (zeros)   16: s_waitcnt vmcnt(0) lgkmcnt(0) + s_nop at the very end of the kernel
#ifndef A3_ABL
#define A3_ABL 0
#endif
#define A3_NTW_MAX 8  // key tiles per wave -> N <= 8 * 4 * 32 = 1024

__device__ __forceinline__ hx8 a3_pack8(const float* v) {
  hx8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (fd_h)v[e];
  return o;
}
// bytes of the LDS region shared by the P fragments (64 B per key) and, after them, the o_pt tile (32 x 96 floats)
__host__ __device__ __forceinline__ int a3_pf_region(int nt) { return 2 * nt * 64 * 16 > 32 * 96 * 4 ? 2 * nt * 64 * 16 : 32 * 96 * 4; }
__device__ __forceinline__ hx8 a3_ld(const half_t* p) { return __builtin_bit_cast(hx8, *(const u16x8*)p); }

// NTW: key tiles per wave (N <= 128 NTW).  NTW = 3 (N <= 384): 234 registers -> launch bound 2 -> TWO blocks per CU hide each
// other's memory latency, so the operands of a tile are simply fetched when needed (DB = false).  NTW = 4: one block per
// CU with all 512 registers, next tile's operands / second V tile in flight under the current one (DB = true).
// SPLIT (Attn3Args.Vt_lo): P V on split operands — the attention weights P = hi + lo and V = hi + lo, each part one half-precision
// value, V_hi P_hi + V_hi P_lo + V_lo P_hi with fp32 accumulation (the accuracy of an fp32 product; the rounding of P and V to half
// precision is the largest error source of the half mode once the per-residue products are split: tests/err_budget.py at bb_gain
// 0.3).  The P_lo fragments take the LDS of the Q fragments (dead after the logits) where those live in LDS, their own region
// otherwise; the V_lo fragments of a d tile are fetched into the registers of its V_hi fragments once those are consumed.  The
// value-point image already is hi / lo: it gets the P_lo term.
template <int A3_NTW, int LB, bool DB, bool SPLIT>
__global__ __launch_bounds__(FD_THREADS, LB) void ipa_attn3_kernel(Attn3Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int N = a.N, H = a.H, nt = (N + 31) / 32, Np = nt * 32;
  float* mxs = (float*)smem;                                 // [4][32]
  float* sms = mxs + 128;                                    // [4][32]
  u16x8* Pfs = (u16x8*)(sms + 128);                          // [2*nt][64] (the region holds at least 12 KB: a3_pf_region)
  // Round 6: the o_pt tile OVERLAYS the P fragments (a block-wide barrier separates the last P read from the first o_pt write).  With its
  // own 12 KB the N = 300 launch asked for 54,288 B per block: three blocks are 162,864 of the CU's 163,840 B, the occupancy API says
  // "3" — and the hardware ran TWO (LDS is handed out in granules: tools/micro/attn3_bench.hip's dispatch timeline showed 128 of the
  // 640 blocks of a B = 8 launch starting only when the first ones had finished: 40 us per call for blocks that take 20).
  float* opr = (float*)Pfs;                                  // [32 queries][96]: o_pt sums, high parts then low parts
  u16x8* Qs = Pfs + a3_pf_region(nt) / 16;                   // LB == 3: the Q fragments of the query tile [16][64] (16 KB)
  // SPLIT: P_lo fragments [2*nt][64].  They overlay the Q fragments (and extend behind them: the launcher sizes the region as the
  // larger of the two), which nobody reads after phase 1 — two block-wide barriers (softmax) lie between
  u16x8* Pls = Qs;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, hi = lane >> 5, li = lane & 31;
  // XCD-aware block -> (sample, head, query tile): consecutive workgroup ids go round-robin over the 8 XCDs (one L2 each),
  // so the id is split as xcd = id % 8 and all nt query tiles of one (sample, head) are given to the SAME XCD: its K, V
  // and Q images are then fetched into one L2 instead of eight.
  const int BH = a.B * H;
  int bhq, qt;
  {
    const int id = blockIdx.x, xcd = id & 7, local = id >> 3;
    const int per = (BH + 7) >> 3;           // (sample, head) pairs per XCD (last ones may be short)
    bhq = xcd * per + local / nt;
    qt = local % nt;
    if (local >= per * nt || bhq >= BH) return;
  }
  const int h = bhq % H, b = bhq / H;
  const long rb = (long)b * N, bh = bhq;
  const long kvh = a.kv_per_sample ? b : bh;  // index of the K / V images: per (sample, head), or per sample (merged projection: the
                                              // node rows are keys and values of every head)
  const int i_raw = qt * 32 + li;
  const bool valid = i_raw < N;
  const int i = valid ? i_raw : N - 1;

  FD_STAMP(0);
  // ---- query-side registers
  // LB == 2: Q fragments in registers.  LB == 3 (three blocks per CU: B H nt blocks then fit one round of the 256 CUs): the 64
  // registers are not affordable; the four waves need the same fragments, so they go to LDS once and are read per k-step
  constexpr bool QLDS = LB == 3 || (LB == 2 && A3_NTW >= 6);  // Q fragments through LDS (frees 64 registers)
  hx8 Qf[QLDS ? 1 : 16];
  {
    const half_t* qr = a.Qb + ((bh * nt + qt) * 16 * 64 + lane) * 8;  // fragment order: 1 KB per k-step
    if constexpr (QLDS) {
#pragma unroll
      for (int s = 0; s < 4; ++s) Qs[(4 * s + wave) * 64 + lane] = *(const u16x8*)(qr + (4 * s + wave) * 512);
      __syncthreads();
    } else {
#pragma unroll
      for (int s = 0; s < 16; ++s) Qf[s] = a3_ld(qr + s * 512);
    }
  }
  const float mi = a.res_mask[rb + i];
  const float gam = a.gamma[h];
  // B operands of the mask / point products (IEEE fp16; pairing with the key-point fragment image: kernels.hpp, fd_kpf layout).  With
  // q = gamma * (the query's 24 global-frame point coordinates), thirds q0 | q1 | q2, h / l = fp16 hi / lo parts, lane half hi:
  //   Bq01h = h(q0) | h(q1)     Bq01l = l(q0) | l(q1)     Bq2h = h(q2) | h(q2)     Bq2x = l(q2) | [0,0,0,1,1,1,0,0]
  //   Bm    = 0 | [32768 m_i, 34464 m_i, 60000, 0, ...]   (2 m_j * 32768 m_i + m_j * 34464 m_i = 1e5 m_i m_j exactly)
  f16x8 Bq01h, Bq01l, Bq2h, Bq2x, Bm;
  {
    const float* qpr = a.qp + ((rb + i) * H + h) * 24;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float v = gam * qpr[8 * hi + e], v2 = gam * qpr[16 + e];
      const _Float16 vh = (_Float16)v, v2h = (_Float16)v2;
      Bq01h[e] = vh;
      Bq01l[e] = (_Float16)(v - (float)vh);
      Bq2h[e] = v2h;
      Bq2x[e] = hi ? (_Float16)((e >= 3 && e < 6) ? 1.f : 0.f) : (_Float16)(v2 - (float)v2h);
      Bm[e] = (_Float16)0.f;
    }
    if (hi) { Bm[0] = (_Float16)(32768.f * mi); Bm[1] = (_Float16)(34464.f * mi); Bm[2] = (_Float16)60000.f; }
  }
  FD_STAMP(1);
  // ---- phase 1: logits of this wave's key tiles t = wave, wave+4, ...  The global operands of tile u+1 (K rows as A
  // fragments, key points, bias row pieces, mask) are fetched while tile u runs through the matrix cores.
  struct TileIn {
    hx8 k[16];
    f16x8 kf[FD_KPF_FRAGS];
    f32x4 bv[4];
  };
  auto tile_load = [&](TileIn& ti, int t) {
    const half_t* kr = a.Kb + ((kvh * nt + t) * 16 * 64 + lane) * 8;  // fragment order (padded keys are zero rows)
#pragma unroll
    for (int s = 0; s < 16; ++s) ti.k[s] = a3_ld(kr + s * 512);
    const half_t* pr = a.kpf + ((bh * nt + t) * (FD_KPF_FRAGS * 64) + lane) * 8;  // key-point fragments (padded keys: marker in X)
#pragma unroll
    for (int f = 0; f < FD_KPF_FRAGS; ++f) ti.kf[f] = __builtin_bit_cast(f16x8, *(const u16x8*)(pr + f * 512));
    const float* bt = a.bias + (((bh * nt + qt) * nt + t) * 32 + li) * 32 + 4 * hi;  // fd_bias_frag_off: this query's row of the tile
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int j0 = 32 * t + 8 * g + 4 * hi;
      f32x4 bv = *(const f32x4*)(bt + 8 * g);
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (j0 + q >= N) bv[q] = 0.f;  // padded keys: the slot is never written
      ti.bv[g] = bv;
    }
  };
  f32x16 S[A3_NTW];
  TileIn tin[DB ? 2 : 1];
  if (DB && wave < nt) tile_load(tin[0], wave);
#pragma unroll
  for (int u = 0; u < A3_NTW; ++u) {
    const int t = wave + 4 * u;
    if (DB && u + 1 < A3_NTW && t + 4 < nt) tile_load(tin[(u + 1) & 1], t + 4);
    if (t < nt) {
      if (!DB) tile_load(tin[0], t);
      const TileIn& ti = tin[DB ? (u & 1) : 0];
      // ONE accumulator for the three terms: it starts at -1e5, the mask product 1e5 m_i m_j brings an unmasked pair back
      // to exactly 0 (a masked one stays at -1e5, a padded key goes to -3.6e9) BEFORE the small terms are added, so nothing
      // is lost to the large constant (the mask product is its own MFMA: every partial sum of it is exactly representable);
      // then gamma (q.k - |k|^2 / 2) on fp16 hi / lo parts (h.h + l.h + h.l: 2^-22 relative, five MFMAs of 32 cycles where the
      // 26-deep fp32 chain of rounds 1 - 3 took thirteen of 64), then Q K^T
      f32x16 acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = -1e5f;
      if (!(A3_ABL & 2)) {
      acc = fd_mfma32_f16(ti.kf[3], Bm, acc);
      acc = fd_mfma32_f16(ti.kf[0], Bq01h, acc);
      acc = fd_mfma32_f16(ti.kf[1], Bq01h, acc);
      acc = fd_mfma32_f16(ti.kf[0], Bq01l, acc);
      acc = fd_mfma32_f16(ti.kf[2], Bq2h, acc);
      acc = fd_mfma32_f16(ti.kf[3], Bq2x, acc);
#pragma unroll
      for (int s = 0; s < 16; ++s)
        acc = fd_mfma32(ti.k[s], QLDS ? __builtin_bit_cast(hx8, Qs[s * 64 + lane]) : Qf[QLDS ? 0 : s], acc);
      }
#pragma unroll
      for (int g = 0; g < 4; ++g)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[4 * g + q] += ti.bv[g][q];
      S[u] = acc;
    }
  }
  FD_STAMP(2);
  // ---- phase 2: softmax over all keys (own registers -> lane^32 -> the other 3 waves through LDS)
  float mx = -3.0e38f;
#pragma unroll
  for (int u = 0; u < A3_NTW; ++u)
    if (wave + 4 * u < nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, S[u][r]);
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  if (hi == 0) mxs[wave * 32 + li] = mx;
  __syncthreads();
  mx = fmaxf(fmaxf(mxs[li], mxs[32 + li]), fmaxf(mxs[64 + li], mxs[96 + li]));
  float sum = 0.f;
#pragma unroll
  for (int u = 0; u < A3_NTW; ++u)
    if (wave + 4 * u < nt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float e = (A3_ABL & 1) ? fmaxf(1.0f + (S[u][r] - mx) * 0.01f, 0.f)
                                     : __builtin_amdgcn_exp2f((S[u][r] - mx) * 1.4426950408889634f);  // exp(x): one multiply + v_exp_f32 (expf adds range fix-ups)
        S[u][r] = e;
        sum += e;
      }
  sum += __shfl_xor(sum, 32, 64);
  if (hi == 0) sms[wave * 32 + li] = sum;
  __syncthreads();
  const float inv = 1.0f / (sms[li] + sms[32 + li] + sms[64 + li] + sms[96 + li]);
  FD_STAMP(3);
  // ---- phase 3: normalise; attention weights -> HBM (for o_pair), P fragments -> LDS
  float* prow = a.probs + (bh * N + i) * N;
  // bf16 hand-over to the MFMA o_pair kernel: row (b, i) = the 8 heads' weights back to back, Np keys each (zero beyond N)
  const long bq = bh / a.H;
  half_t* prow16 = a.probs_h16 ? a.probs_h16 + ((bq * N + i) * a.H + (bh - bq * a.H)) * (long)a.Np : nullptr;
#pragma unroll
  for (int u = 0; u < A3_NTW; ++u) {
    const int t = wave + 4 * u;
    if (t < nt) {
      float v[16];
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = S[u][r] * inv;
      if (valid && prow16) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u16x4 o = {f2h(v[4 * g]), f2h(v[4 * g + 1]), f2h(v[4 * g + 2]), f2h(v[4 * g + 3])};
          *(u16x4*)(prow16 + 32 * t + 8 * g + 4 * hi) = o;  // (masked / padded keys carry exact zeros)
        }
      } else if (valid) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int j0 = 32 * t + 8 * g + 4 * hi;
          if (j0 + 3 < N && (N & 3) == 0) {
            f32x4 o = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
            *(f32x4*)(prow + j0) = o;
          } else {
#pragma unroll
            for (int q = 0; q < 4; ++q)
              if (j0 + q < N) prow[j0 + q] = v[4 * g + q];
          }
        }
      }
      const hx8 p0 = a3_pack8(v), p1 = a3_pack8(v + 8);
      Pfs[(2 * t) * 64 + lane] = __builtin_bit_cast(u16x8, p0);
      Pfs[(2 * t + 1) * 64 + lane] = __builtin_bit_cast(u16x8, p1);
      if constexpr (SPLIT) {
        float w[16];
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          w[r] = v[r] - (float)p0[r];
          w[8 + r] = v[8 + r] - (float)p1[r];
        }
        Pls[(2 * t) * 64 + lane] = __builtin_bit_cast(u16x8, a3_pack8(w));
        Pls[(2 * t + 1) * 64 + lane] = __builtin_bit_cast(u16x8, a3_pack8(w + 8));
      }
    }
  }
  FD_STAMP(4);
  __syncthreads();
  FD_STAMP(5);
  FD_STAMP(6);
  // ---- phase 4: O^T[d, query] for d tiles {2w, 2w+1} over all keys (A = Vt fragments from HBM/L2, B = P from LDS); waves
  // 0..2 also take one 32-row tile of the v_pts image (rows = point coordinates as bf16 high + low parts): o_pt on the
  // matrix core, sum_j a v_pts_j to ~2^-17 relative in v_pts (the weights are the same bf16 P as for o)
  if (!(A3_ABL & 4)) {
    constexpr int KSM = 2 * 4 * A3_NTW;  // k-steps of 16 keys at the maximum N
    const int ks = 2 * nt;
    hx8 Va[DB ? 2 : 1][KSM];
    auto v_load = [&](auto BUF, const half_t* base) {
      constexpr int bf = decltype(BUF)::value;
#pragma unroll
      for (int s = 0; s < KSM; ++s)
        if (s < ks) Va[bf][s] = a3_ld(base + ((size_t)s * 64 + lane) * 8);
    };
    auto v_mma = [&](auto BUF, f32x16& acc) {
      constexpr int bf = decltype(BUF)::value;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int s = 0; s < KSM; ++s)
        if (s < ks) acc = fd_mfma32(Va[bf][s], __builtin_bit_cast(hx8, Pfs[s * 64 + lane]), acc);
    };
    auto v_mma_add = [&](auto BUF, f32x16& acc, const u16x8* P) {  // acc += (fragments in buffer BUF) x (weight fragments P)
      constexpr int bf = decltype(BUF)::value;
#pragma unroll
      for (int s = 0; s < KSM; ++s)
        if (s < ks) acc = fd_mfma32(Va[bf][s], __builtin_bit_cast(hx8, P[s * 64 + lane]), acc);
    };
    auto o_store = [&](const f32x16& acc, int dt) {
      if (valid && a.out_h16) {  // bf16 features: exactly what the output projection's bf16 GEMM would round them to
        half_t* orow = a.out_h16 + (rb + i) * a.out_ld + (long)h * A3_C + 32 * dt + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const u16x4 o = {f2h(acc[4 * g]), f2h(acc[4 * g + 1]), f2h(acc[4 * g + 2]), f2h(acc[4 * g + 3])};
          *(u16x4*)(orow + 8 * g) = o;
        }
      } else if (valid) {
        float* orow = a.out + (rb + i) * a.out_ld + (long)h * A3_C + 32 * dt + 4 * hi;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
          *(f32x4*)(orow + 8 * g) = o;
        }
      }
    };
    f32x16 acc;
    if constexpr (SPLIT) {
      constexpr std::integral_constant<int, 0> B0{};
      const half_t* vlo = a.Vt_lo;
#pragma unroll
      for (int dd = 0; dd < 2; ++dd) {
        const long off = ((kvh * (A3_C / 32) + 2 * wave + dd) * ks) * 512;
        if (dd == 0) v_load(B0, a.Vt + off);   // (d tile 1's V_hi was requested under d tile 0's last products)
        v_mma(B0, acc);                        // V_hi P_hi
        v_mma_add(B0, acc, Pls);               // V_hi P_lo
        v_load(B0, vlo + off);
        v_mma_add(B0, acc, Pfs);               // V_lo P_hi
        if (dd == 0) v_load(B0, a.Vt + ((kvh * (A3_C / 32) + 2 * wave + 1) * ks) * 512);
        else if (wave < 3) v_load(B0, a.vpt + ((bh * 3 + wave) * ks) * 512);
        o_store(acc, 2 * wave + dd);
      }
    } else if constexpr (DB) {
      v_load(std::integral_constant<int, 0>{}, a.Vt + ((kvh * (A3_C / 32) + 2 * wave) * ks) * 512);
      v_load(std::integral_constant<int, 1>{}, a.Vt + ((kvh * (A3_C / 32) + 2 * wave + 1) * ks) * 512);
      v_mma(std::integral_constant<int, 0>{}, acc);
      if (wave < 3) v_load(std::integral_constant<int, 0>{}, a.vpt + ((bh * 3 + wave) * ks) * 512);  // points tile, under tile 1
      o_store(acc, 2 * wave);
      v_mma(std::integral_constant<int, 1>{}, acc);
      o_store(acc, 2 * wave + 1);
    } else {
      v_load(std::integral_constant<int, 0>{}, a.Vt + ((kvh * (A3_C / 32) + 2 * wave) * ks) * 512);
      v_mma(std::integral_constant<int, 0>{}, acc);
      v_load(std::integral_constant<int, 0>{}, a.Vt + ((kvh * (A3_C / 32) + 2 * wave + 1) * ks) * 512);
      o_store(acc, 2 * wave);
      v_mma(std::integral_constant<int, 0>{}, acc);
      if (wave < 3) v_load(std::integral_constant<int, 0>{}, a.vpt + ((bh * 3 + wave) * ks) * 512);
      o_store(acc, 2 * wave + 1);
    }
    if (wave < 3) {
      v_mma(std::integral_constant<int, 0>{}, acc);
      if constexpr (SPLIT)
        if (wave < 2) v_mma_add(std::integral_constant<int, 0>{}, acc, Pls);  // (tile 2 holds low parts only: P_lo v_lo is below fp32 resolution)
    }
    __syncthreads();  // every wave has read its last P fragment: the o_pt tile may overwrite them
    if (wave < 3) {
#pragma unroll
      for (int g = 0; g < 4; ++g) {  // D rows 32 wave + 8g + 4hi + q of query li
        f32x4 o = {acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]};
        *(f32x4*)(opr + li * 96 + 32 * wave + 8 * g + 4 * hi) = o;
      }
    }
  }
  __syncthreads();
  // ---- o_pt = R_i^T (sum - t_i) and its norm (ipa_pytorch.py:296-308), 32 queries x 12 points
  for (int it = tid; it < 32 * 12; it += FD_THREADS) {
    const int q = it / 12, pt = it % 12, iq = qt * 32 + q;
    if (iq < N) {
      const float* o = opr + q * 96 + pt * 3;
      const float sx = o[0] + o[36], sy = o[1] + o[37], sz = o[2] + o[38];
      const float* R = a.rot + (rb + iq) * 9;
      const float* T = a.trans + (rb + iq) * 3;
      const float x = sx - T[0], y = sy - T[1], z = sz - T[2];
      const float ox = R[0] * x + R[3] * y + R[6] * z;
      const float oy = R[1] * x + R[4] * y + R[7] * z;
      const float oz = R[2] * x + R[5] * y + R[8] * z;
      const int HP = H * 12;
      const float on = sqrtf(ox * ox + oy * oy + oz * oz + 1e-8f);
      if (a.out_h16) {
        half_t* oo = a.out_h16 + (rb + iq) * a.out_ld + a.pt_off + h * 12 + pt;
        oo[0] = f2h(ox); oo[HP] = f2h(oy); oo[2 * HP] = f2h(oz); oo[3 * HP] = f2h(on);
      } else {
        float* oo = a.out + (rb + iq) * a.out_ld + a.pt_off + h * 12 + pt;
        oo[0] = ox; oo[HP] = oy; oo[2 * HP] = oz;
        oo[3 * HP] = on;
      }
    }
  }
  FD_STAMP(7);
  if (A3_ABL & 16) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
}

int fd_attention3_supported(const Attn3Args& a) {
  return a.N >= 1 && a.N <= A3_NTW_MAX * 4 * 32 && a.H <= 8 && (a.H & 1) == 0 && a.Np == ((a.N + 31) / 32) * 32 && a.vpt != nullptr;
}

template <int NTW, int LB, bool DB>
static int a3_launch(const Attn3Args& a, dim3 grid, size_t smem, hipStream_t st) {
  // (the attribute is per device: set on every launch rather than cached per process)
  if (a.Vt_lo) {
    if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)ipa_attn3_kernel<NTW, LB, DB, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return FDIPT_ELAUNCH;
    hipLaunchKernelGGL((ipa_attn3_kernel<NTW, LB, DB, true>), grid, dim3(FD_THREADS), smem, st, a);
  } else {
    if (smem > 64 * 1024 && hipFuncSetAttribute((const void*)ipa_attn3_kernel<NTW, LB, DB, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem) != hipSuccess)
      return FDIPT_ELAUNCH;
    hipLaunchKernelGGL((ipa_attn3_kernel<NTW, LB, DB, false>), grid, dim3(FD_THREADS), smem, st, a);
  }
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

int fd_attention3(const Attn3Args& a, hipStream_t st) {
  const int nt = (a.N + 31) / 32;
  if (!fd_attention3_supported(a)) return FDIPT_ESIZE;
  if (!a.kpf) return FDIPT_EINVAL;
  // reduction buffers 1 KB + P fragments 64 B per key (>= the 12 KB of the o_pt tile that later overlays them), then ONE region for the Q fragments (16 KB, the variants
  // that keep them in LDS) and / or the P_lo fragments (64 B per key, split P V): they are never live together
  // (round 6: the o_pt tile overlays the P fragments: a3_pf_region)
  const size_t pf = (size_t)2 * nt * 64 * 16, base = 2 * 128 * 4 + (size_t)a3_pf_region(nt) + 16;
  const size_t plo = a.Vt_lo ? pf : 0;
  auto with = [&](bool qlds) { const size_t q = qlds ? 16384 : 0; return base + (q > plo ? q : plo); };
  const int per = (a.B * a.H + 7) / 8;  // see the block mapping in the kernel
  const dim3 grid(8 * per * nt);
  if (a.N <= 3 * 4 * 32 && !FD_DEV_ENV("FDIPT_A3_LB2")) return a3_launch<3, 3, false>(a, grid, with(true), st);
  if (a.N <= 3 * 4 * 32) return a3_launch<3, 2, false>(a, grid, with(false), st);
  if (a.N <= 4 * 4 * 32) return a3_launch<4, 1, true>(a, grid, with(false), st);
  // 512 < N <= 1024 (TCR-pMHC complexes, long chains): 6 / 8 key tiles per wave, operands fetched per tile
#ifndef A3_MID2
#define A3_MID2 1
#endif
  if (a.N <= 6 * 4 * 32 && A3_MID2) return a3_launch<6, 2, false>(a, grid, with(true), st);
  if (a.N <= 6 * 4 * 32) return a3_launch<6, 1, false>(a, grid, with(false), st);
  return a3_launch<8, 1, false>(a, grid, with(false), st);
}
