// ipa_proj2.hip — the fused IPA input projection (q | k,v | q_pts | kv_pts: framedipt/model/ipa_pytorch.py:218-245 as ONE
// Linear c_s -> 3 H C + PT, bf16 operands) whose epilogue writes the attention operand images, second generation.
//
// ipa_proj_kernel (gemm.hip) is a generic 128 x 128-tile GEMM: every one of its 54 column blocks re-reads and re-converts
// the 128 x 256 fp32 activation tile and synchronises four times over K = 256; 44 us for 8.4 GFLOP.  Here K = 256 is small
// enough for a wave to keep its activation fragments in registers for the whole kernel:
//   * block = 128 rows; the rows are converted to bf16 once (coalesced fp32 loads -> LDS -> fragments), each wave (wr, wc)
//     keeps the 2 x 16 fragments of its 64 rows (128 registers);
//   * the weights are a fragment image [column tile][k-step][lane][8] (fd_chain_build_image); the 64 KB of a 128-column block
//     reach LDS by LDS-DMA, double-buffered against the MFMAs of the previous column block (the activation staging buffer is
//     reused); a wave reads the 2 x 16 fragments of its two column tiles with linear ds_read_b128;
//   * a block walks the column blocks blockIdx.y, blockIdx.y + gridDim.y, ...: 64 MFMAs per wave per column block, then the
//     epilogue of the block's kind, same layouts as ipa_proj_kernel (gemm.hip:165-172): Q / K blocks run with the MFMA
//     operands exchanged (lane = row, registers = 4-runs of channels -> 8 B pieces of the fragment images), V blocks
//     untransposed (lane = channel, registers = 4-runs of keys -> 8 B pieces of V^T), point columns as fp32 rows.
#include "common.hpp"
#include "kernels.hpp"

#define P2_K 256
#define P2_KS (P2_K / 16)
#define P2_WBLK (4 * P2_KS * 1024)          // one 128-column block of the weight image: 64 KB
#define P2_XROW (P2_K * 2)                  // bf16 activation row in LDS: 128 rows = exactly one weight buffer (buffer 1)
#define P2_LDS (2 * P2_WBLK)

typedef fd_h p2_hx2 __attribute__((ext_vector_type(2)));
typedef float p2_f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned p2_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned p2_u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ int p2_perm16(int pos) { return 4 * (pos >> 3) + (pos & 3) + 8 * ((pos & 7) >> 2); }
__device__ __forceinline__ void p2_dma16(const void* gsrc, unsigned lds_dst) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(gsrc) : "memory", "m0");
}
typedef const __attribute__((address_space(3))) u16x8* p2_lds_u16x8;
__device__ __forceinline__ hx8 p2_frag(unsigned off) { return __builtin_bit_cast(hx8, *(p2_lds_u16x8)(unsigned long)off); }

// QK: this block walks the Q / K column blocks (n_walk_qk walkers, blockIdx.y < n_walk_qk) or the V / point blocks
// SPLIT: split operands (x = hi + lo, W = hi + lo, each part one half-precision value; W_hi x_hi + W_hi x_lo + W_lo x_hi with fp32
// accumulation: the accuracy of an fp32 product).  The q / k / v / point projections are per-residue quantities whose rounding
// errors are coherent over all keys of the attention (tests/err_budget.py at bb_gain 0.3: the largest single group of the half mode).
// Buffer 0 holds the hi block of the weight image, buffer 1 the lo block (no double buffering: the next block is requested once
// every wave is done with the current one, and the epilogue's stores run under that DMA); both activation parts stay in registers.
template <bool QK, bool SPLIT>
__device__ __forceinline__ void ipa_proj2_body(const ProjArgs& a, int n_cblk, int walker, int n_walkers, char* smem) {
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
  const int wr = wave >> 1, wc = wave & 1;
  const int M = a.B * a.N, HC = a.H * a.C, NOUT = 3 * HC + a.PT, ntl = a.Np >> 5;
  const int m0 = blockIdx.x * 128;
  const char* wimg = (const char*)a.W_img;
  // ---- column blocks of this block's class, in class order: k-th Q/K block / k-th V-or-point block -> column block index
  const int nq = HC / 128, per_head = (2 * a.C) / 128, n_class = QK ? 2 * nq : n_cblk - 2 * nq;
  auto cblk_of = [&](int k) {
    if (QK) return k < nq ? k : nq + ((k - nq) / (per_head / 2)) * per_head + (k - nq) % (per_head / 2);
    const int nv = nq;  // V blocks, then the point blocks
    return k < nv ? nq + (k / (per_head / 2)) * per_head + per_head / 2 + k % (per_head / 2) : 3 * nq + (k - nv);
  };
  int kb = walker;
  const char* wimg_lo = (const char*)a.W_img_lo;
  auto request = [&](int c, int buf) {
    const char* src = (SPLIT && buf ? wimg_lo : wimg) + (size_t)c * P2_WBLK;
#pragma unroll
    for (int u = 0; u < P2_WBLK / (FD_THREADS * 16); ++u)
      p2_dma16(src + (size_t)(u * FD_THREADS + tid) * 16, lds0 + buf * P2_WBLK + (unsigned)(u * FD_THREADS + (tid & ~63)) * 16);
  };
  FD_STAMP(0);
  if (kb < n_class) request(cblk_of(kb), 0);  // the first weight block is on its way while the activations are staged
  // ---- activation rows: fp32 -> bf16 -> LDS (buffer 1: 128 rows x 512 B, 16 B chunk c of row r at c ^ (r & 15)), once
  hx8 Af[2][P2_KS];
  hx8 Al[SPLIT ? 2 : 1][SPLIT ? P2_KS : 1];
#pragma unroll
  for (int pass = 0; pass < (SPLIT ? 2 : 1); ++pass) {
    {
      char* xs = smem + P2_WBLK;
#pragma unroll
      for (int it = 0; it < 32; ++it) {  // 128 rows x 64 float4 (all 32 requests of a thread in flight: one round trip)
        const int idx = tid + it * FD_THREADS, r = idx >> 6, c4 = idx & 63;
        const int gr = m0 + r < M ? m0 + r : M - 1;
        f32x4 x = *(const f32x4*)(a.A + (long)gr * a.lda + 4 * c4);
        if (pass) {  // lo part: x - half(x), exact in fp32
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] -= h2f(f2h(x[q]));
        }
        const p2_u32x2 h = {fd_cvt_pk(x[0], x[1]), fd_cvt_pk(x[2], x[3])};
        *(p2_u32x2*)(xs + r * P2_XROW + (((c4 >> 1) ^ (r & 15)) << 4) + 8 * (c4 & 1)) = h;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < P2_KS; ++s) {
        const int r = (wr * 2 + i) * 32 + li;
        const hx8 f = p2_frag(lds0 + P2_WBLK + r * P2_XROW + (((2 * s + hi) ^ (r & 15)) << 4));
        if (pass == 0) Af[i][s] = f;
        else if constexpr (SPLIT) Al[i][s] = f;
      }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the fragments are in registers before the buffer is overwritten
    if (SPLIT) __syncthreads();
  }
  if (SPLIT && kb < n_class) request(cblk_of(kb), 1);
  FD_STAMP(1);
  // ---- store addressing, split into a part that depends on the row(s) of a register group (computed once) and a part that
  // depends on the column block (once per block): an epilogue unit adds the two and a compile-time constant
  //   Q / K images (lane = row): element ((((b H + h) ntl + (r >> 5)) (C >> 4) + (cc >> 4)) 64 + ((cc >> 3) & 1) 32 + (r & 31)) 8 + (cc & 7)
  //   V^T image (lane = channel, registers = 4 consecutive keys): ((((b H + h) (C >> 5) + (cc >> 5)) 2 ntl + (pp >> 4)) 64 +
  //                                                               ((pp >> 3) & 1) 32 + (cc & 31)) 8 + (pp & 7)
  //   points (fp32 rows): row * PT + column
  int qk_row[2];      // QK: row part of tile i (this lane's row), -1 = row beyond M
  int v_row[2][4];    // V / points: row part of register group (i, g) (rows mg .. mg + 3), -1 = beyond M
  int p_row[2][4];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = m0 + (wr * 2 + i) * 32 + li;
    const int b = m < M ? m / a.N : 0, r = m - b * a.N;
    qk_row[i] = m < M ? ((b * a.H * ntl + (r >> 5)) * (a.C >> 4) * 64 + (r & 31)) * 8 : -1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int mg = m0 + (wr * 2 + i) * 32 + 8 * g + 4 * hi;
      const int bb = mg < M ? mg / a.N : 0, key = mg - bb * a.N, pp = (key & ~15) + p2_perm16(key & 15);
      v_row[i][g] = mg + 3 < M ? (bb * a.H * (a.C >> 5) * (2 * ntl) + (pp >> 4)) * 512 + ((pp >> 3) & 1) * 256 + (pp & 7) : -1;
      p_row[i][g] = mg + 3 < M ? mg * a.PT : -1;
    }
  }
  // One 4-run (registers 4 g .. 4 g + 3) of tile (i, j) of a column block: the epilogue is cut into these 16 units so that the
  // units of the PREVIOUS column block ride under the 16 k-steps of the current one (with one wave per SIMD nothing else
  // would overlap the stores; and stores issued a whole column block before the next wait never stall it).
  // cpart: column part of the block (QK: head; V: per j (head, channel of this lane); points: column of this lane, or -1)
  auto epi_unit = [&](const f32x16 (&acc)[2][2], const f32x4 (&bq)[2][4], const int (&cpart)[2], int kind, int i, int j, int g) {
    if constexpr (QK) {
      // the rows of the Q / K weight tiles are permuted in the image (fd_ipa_proj2_permute_image) so that register r of a lane
      // is channel 16 (r >> 3) + 8 hi + (r & 7) of the tile: 8 consecutive channels = one whole 16 B fragment unit per lane and
      // 16-group G = g >> 1; the odd g of a pair has nothing left to do
      if ((g & 1) || qk_row[i] < 0) return;
      const f32x4 b0 = bq[j][g], b1 = bq[j][g + 1];
      const float sc = kind == 0 ? a.qscale : 1.f;
      const p2_u32x4 o = {fd_cvt_pk((acc[i][j][4 * g] + b0[0]) * sc, (acc[i][j][4 * g + 1] + b0[1]) * sc),
                          fd_cvt_pk((acc[i][j][4 * g + 2] + b0[2]) * sc, (acc[i][j][4 * g + 3] + b0[3]) * sc),
                          fd_cvt_pk((acc[i][j][4 * g + 4] + b1[0]) * sc, (acc[i][j][4 * g + 5] + b1[1]) * sc),
                          fd_cvt_pk((acc[i][j][4 * g + 6] + b1[2]) * sc, (acc[i][j][4 * g + 7] + b1[3]) * sc)};
      // cc = cbase + (2 wc + j) 32 + 16 G + 8 hi: cc >> 4 = (cbase >> 4) + 2 (2 wc + j) + G, (cc >> 3) & 1 = hi, cc & 7 = 0
      half_t* dst = (kind == 0 ? a.Qb : a.Kb) + (long)qk_row[i] + cpart[0] + ((2 * j + (g >> 1)) * 64) * 8;
      *(p2_u32x4*)dst = o;
    } else {
      if (cpart[j] < 0) return;
      const float bv = bq[j][0][0];
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q] + bv;
      if (kind == 2) {
        if (v_row[i][g] < 0) return;
        const p2_u32x2 o = {fd_cvt_pk(v[0], v[1]), fd_cvt_pk(v[2], v[3])};
        *(p2_u32x2*)(a.Vt + (long)v_row[i][g] + cpart[j]) = o;
        if (SPLIT && a.Vt_lo) {  // V - half(V): the attention multiplies P with V_hi + V_lo (attention4.hip)
          float w[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) w[q] = v[q] - h2f(f2h(v[q]));
          const p2_u32x2 ol = {fd_cvt_pk(w[0], w[1]), fd_cvt_pk(w[2], w[3])};
          *(p2_u32x2*)(a.Vt_lo + (long)v_row[i][g] + cpart[j]) = ol;
        }
      } else {
        if (p_row[i][g] < 0) return;
        float* dst = a.pts + (long)p_row[i][g] + cpart[j];
#pragma unroll
        for (int q = 0; q < 4; ++q) dst[q * a.PT] = v[q];
      }
    }
  };
  // kind (0 Q, 1 K, 2 V, 3 points) and column part of column block c
  auto col_part = [&](int c, int (&cpart)[2]) {
    const int n0 = c * 128;
    if constexpr (QK) {
      const bool isq = n0 < HC;
      const int nn0 = isq ? n0 : n0 - HC, hh = isq ? nn0 / a.C : nn0 / (2 * a.C), cbase = isq ? nn0 % a.C : nn0 % (2 * a.C);
      cpart[0] = (hh * ntl * (a.C >> 4) * 64 + ((cbase >> 4) + 4 * wc) * 64 + hi * 32) * 8;
      cpart[1] = 0;
      return isq ? 0 : 1;
    } else {
      const bool isv = n0 < 3 * HC;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + (wc * 2 + j) * 32 + li;
        if (isv) {
          const int nn = n - HC, hh = nn / (2 * a.C), cc = nn % (2 * a.C) - a.C;
          cpart[j] = (hh * (a.C >> 5) + (cc >> 5)) * (2 * ntl) * 512 + (cc & 31) * 8;
        } else cpart[j] = n < NOUT ? n - 3 * HC : -1;
      }
      return isv ? 2 : 3;
    }
  };
  // bias of a column block in the layout its epilogue wants (requested when the block's MFMAs start, used a block later:
  // a bias load inside an epilogue unit would wait for itself AND, vmcnt being in order, for the whole weight DMA before it)
  auto load_bias = [&](f32x4 (&bq)[2][4], int n0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (QK) bq[j][g] = *(const f32x4*)(a.bias + n0 + (wc * 2 + j) * 32 + 16 * (g >> 1) + 8 * hi + 4 * (g & 1));  // (permuted rows)
        else if (g == 0) {
          const int n = n0 + (wc * 2 + j) * 32 + li;
          bq[j][0][0] = n < NOUT ? a.bias[n] : 0.f;
        }
      }
  };
  auto step = [&](f32x16 (&accN)[2][2], f32x4 (&bqN)[2][4], int (&cpN)[2], int& kindN, const f32x16 (&accP)[2][2], const f32x4 (&bqP)[2][4],
                  const int (&cpP)[2], int kindP, int k, bool has_prev, int bufc) {
    __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this thread's share of the block has landed (issued a whole block ago)
    __syncthreads();
    if (k + n_walkers < n_class) request(cblk_of(k + n_walkers), bufc ^ 1);
    load_bias(bqN, cblk_of(k) * 128);
    kindN = col_part(cblk_of(k), cpN);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) accN[i][j][r] = 0.f;
    (void)k;
    const unsigned wb = lds0 + bufc * P2_WBLK + (wc * 2) * (P2_KS * 1024) + lane * 16;
    // weight fragments two k-steps ahead of their MFMAs (one wave per SIMD: nothing else hides the LDS latency); pinned
    constexpr int DEPTH = 3;
    hx8 w0[DEPTH], w1[DEPTH];
#pragma unroll
    for (int s = 0; s < DEPTH - 1; ++s) {
      w0[s] = p2_frag(wb + s * 1024);
      w1[s] = p2_frag(wb + (P2_KS + s) * 1024);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < P2_KS; ++s) {
      if (s + DEPTH - 1 < P2_KS) {
        w0[(s + DEPTH - 1) % DEPTH] = p2_frag(wb + (s + DEPTH - 1) * 1024);
        w1[(s + DEPTH - 1) % DEPTH] = p2_frag(wb + (P2_KS + s + DEPTH - 1) * 1024);
      }
      if constexpr (QK) {  // operands exchanged: lane = row
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          accN[i][0] = fd_mfma32(w0[s % DEPTH], Af[i][s], accN[i][0]);
          accN[i][1] = fd_mfma32(w1[s % DEPTH], Af[i][s], accN[i][1]);
        }
      } else {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          accN[i][0] = fd_mfma32(Af[i][s], w0[s % DEPTH], accN[i][0]);
          accN[i][1] = fd_mfma32(Af[i][s], w1[s % DEPTH], accN[i][1]);
        }
      }
      // (two units per k-step in the first half of the block: their stores then have the second half to retire before the
      //  vmcnt(0) at the top of the next block, which cannot tell them from the weight DMA it is waiting for)
      if (has_prev && s < 8) {
        epi_unit(accP, bqP, cpP, kindP, (2 * s) >> 3, ((2 * s) >> 2) & 1, (2 * s) & 3);
        epi_unit(accP, bqP, cpP, kindP, (2 * s + 1) >> 3, ((2 * s + 1) >> 2) & 1, (2 * s + 1) & 3);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  if constexpr (SPLIT) {
    f32x16 acc[2][2];
    f32x4 bq[2][4];
    int cp[2] = {0, 0};
    for (; kb < n_class; kb += n_walkers) {
      __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this thread's share of both parts of the block has landed
      __syncthreads();
      load_bias(bq, cblk_of(kb) * 128);
      const int kind = col_part(cblk_of(kb), cp);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      const unsigned wbh = lds0 + (wc * 2) * (P2_KS * 1024) + lane * 16, wbl = wbh + P2_WBLK;
      constexpr int DEPTH = 2;
      hx8 h0[DEPTH], h1[DEPTH], l0[DEPTH], l1[DEPTH];
      h0[0] = p2_frag(wbh); h1[0] = p2_frag(wbh + P2_KS * 1024); l0[0] = p2_frag(wbl); l1[0] = p2_frag(wbl + P2_KS * 1024);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < P2_KS; ++s) {
        if (s + 1 < P2_KS) {
          h0[(s + 1) % DEPTH] = p2_frag(wbh + (s + 1) * 1024); h1[(s + 1) % DEPTH] = p2_frag(wbh + (P2_KS + s + 1) * 1024);
          l0[(s + 1) % DEPTH] = p2_frag(wbl + (s + 1) * 1024); l1[(s + 1) % DEPTH] = p2_frag(wbl + (P2_KS + s + 1) * 1024);
        }
        const int c = s % DEPTH;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (QK) {  // operands exchanged: lane = row.  Small terms first, then the hi x hi product
            acc[i][0] = fd_mfma32(l0[c], Af[i][s], acc[i][0]); acc[i][1] = fd_mfma32(l1[c], Af[i][s], acc[i][1]);
            acc[i][0] = fd_mfma32(h0[c], Al[i][s], acc[i][0]); acc[i][1] = fd_mfma32(h1[c], Al[i][s], acc[i][1]);
            acc[i][0] = fd_mfma32(h0[c], Af[i][s], acc[i][0]); acc[i][1] = fd_mfma32(h1[c], Af[i][s], acc[i][1]);
          } else {
            acc[i][0] = fd_mfma32(Af[i][s], l0[c], acc[i][0]); acc[i][1] = fd_mfma32(Af[i][s], l1[c], acc[i][1]);
            acc[i][0] = fd_mfma32(Al[i][s], h0[c], acc[i][0]); acc[i][1] = fd_mfma32(Al[i][s], h1[c], acc[i][1]);
            acc[i][0] = fd_mfma32(Af[i][s], h0[c], acc[i][0]); acc[i][1] = fd_mfma32(Af[i][s], h1[c], acc[i][1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      // the bias must have landed BEFORE the next block is requested: a compiler-placed vmcnt wait behind those (invisible)
      // DMAs would wait for all of them, vmcnt being in order
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) asm volatile("" : "+v"(bq[j][g]));
      __syncthreads();  // every wave is done with both buffers
      if (kb + n_walkers < n_class) { request(cblk_of(kb + n_walkers), 0); request(cblk_of(kb + n_walkers), 1); }
#pragma unroll
      for (int u = 0; u < 16; ++u) epi_unit(acc, bq, cp, kind, u >> 3, (u >> 2) & 1, u & 3);
    }
    return;
  }
  f32x16 acc0[2][2], acc1[2][2];
  f32x4 bq0[2][4], bq1[2][4];
  int cp0[2] = {0, 0}, cp1[2] = {0, 0}, kind0 = 0, kind1 = 0;
  bool has_prev = false;
  int n_step = 0;
  (void)n_step;
  for (;;) {
    if (kb >= n_class) break;
    step(acc0, bq0, cp0, kind0, acc1, bq1, cp1, kind1, kb, has_prev, 0);
    FD_STAMP(2 + n_step); ++n_step;
    has_prev = true;
    kb += n_walkers;
    if (kb >= n_class) {
#pragma unroll
      for (int s = 0; s < 16; ++s) epi_unit(acc0, bq0, cp0, kind0, s >> 3, (s >> 2) & 1, s & 3);
      break;
    }
    step(acc1, bq1, cp1, kind1, acc0, bq0, cp0, kind0, kb, true, 1);
    FD_STAMP(2 + n_step); ++n_step;
    kb += n_walkers;
    if (kb >= n_class) {
#pragma unroll
      for (int s = 0; s < 16; ++s) epi_unit(acc1, bq1, cp1, kind1, s >> 3, (s >> 2) & 1, s & 3);
      break;
    }
  }
}
template <bool SPLIT>
__global__ __launch_bounds__(FD_THREADS, 1) void ipa_proj2_kernel(ProjArgs a, int n_cblk, int n_walk_qk) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if ((int)blockIdx.y < n_walk_qk) ipa_proj2_body<true, SPLIT>(a, n_cblk, blockIdx.y, n_walk_qk, smem);
  else ipa_proj2_body<false, SPLIT>(a, n_cblk, blockIdx.y - n_walk_qk, gridDim.y - n_walk_qk, smem);
}

// Row permutation of the Q / K tiles of the weight image (see the Q / K epilogue unit): image row rho of a 32-row tile takes the
// weight row of channel 16 (r >> 3) + 8 hi + (r & 7) with hi = (rho >> 2) & 1, r = (rho & 3) + 4 (rho >> 3).  Tiles: the q part
// (H C / 32 tiles), then per head 2 C / 32 tiles of which the first C / 32 are K.  In place, one block per (tile, k-step).
__global__ void p2_permute_rows_kernel(half_t* img, int ks, int n_q_tiles, int per_head) {
  const int tile = blockIdx.x / ks, s = blockIdx.x % ks;
  const bool qk = tile < n_q_tiles || ((tile - n_q_tiles) % per_head) < per_head / 2;
  if (!qk) return;
  __shared__ u16x8 buf[64];
  u16x8* frag = (u16x8*)img + ((size_t)tile * ks + s) * 64;
  const int lane = threadIdx.x, rho = lane & 31, half = lane >> 5;
  buf[lane] = frag[lane];
  __syncthreads();
  const int hi = (rho >> 2) & 1, r = (rho & 3) + 4 * (rho >> 3), c = 16 * (r >> 3) + 8 * hi + (r & 7);
  frag[lane] = buf[c + 32 * half];
}
int fd_ipa_proj2_permute_image(void* img, int H, int C, int K, hipStream_t st) {
  if ((C & 31) || (K & 15)) return FDIPT_EINVAL;
  const int n_q = H * C / 32, n_kv = 2 * H * C / 32, ks = K / 16;
  hipLaunchKernelGGL(p2_permute_rows_kernel, dim3((n_q + n_kv) * ks), dim3(64), 0, st, (half_t*)img, ks, n_q, 2 * C / 32);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
int fd_ipa_proj2_supported(const ProjArgs& a) {
  const int HC = a.H * a.C;
  return a.K == P2_K && a.W_img && (a.N & 3) == 0 && (a.C % 128) == 0 && (HC % 128) == 0 && (a.lda & 3) == 0 && (a.Np & 31) == 0 && (a.PT & 3) == 0;
}
int fd_ipa_proj2(const ProjArgs& a, hipStream_t st) {
  if (!fd_ipa_proj2_supported(a)) return FDIPT_EINVAL;
  const int M = a.B * a.N, NOUT = 3 * a.H * a.C + a.PT;
  const int n_cblk = cdiv(NOUT, 128), n_rblk = cdiv(M, 128);
  // (the attribute is per device: set on every launch — a host-side table lookup — rather than cached per process)
  if (hipFuncSetAttribute(a.W_img_lo ? (const void*)ipa_proj2_kernel<true> : (const void*)ipa_proj2_kernel<false>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, P2_LDS) != hipSuccess)
    return FDIPT_ELAUNCH;
  // one block per CU: about 256 / row-blocks column walkers per row block, split between the two column classes in
  // proportion to their column blocks
  int ncg = 256 / n_rblk;
  if (ncg < 2) ncg = 2;
  const int n_qk = 2 * (a.H * a.C / 128), n_other = n_cblk - n_qk;
  // (a Q / K column block costs about 6 / 7 of a V / point block: 16 B stores against 8 B / 4 B ones)
  int wq = (ncg * n_qk * 6 + (n_qk * 6 + n_other * 7) / 2) / (n_qk * 6 + n_other * 7);
  if (wq < 1) wq = 1;
  if (wq > ncg - 1) wq = ncg - 1;
  if (wq > n_qk) wq = n_qk;
  int wo = ncg - wq;
  if (wo > n_other) wo = n_other;
  if (a.W_img_lo) hipLaunchKernelGGL(ipa_proj2_kernel<true>, dim3(n_rblk, wq + wo), dim3(FD_THREADS), P2_LDS, st, a, n_cblk, wq);
  else hipLaunchKernelGGL(ipa_proj2_kernel<false>, dim3(n_rblk, wq + wo), dim3(FD_THREADS), P2_LDS, st, a, n_cblk, wq);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
