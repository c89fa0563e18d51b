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
#include <type_traits>

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
typedef const __attribute__((address_space(3))) f32x4* p2_lds_f32x4;
typedef const __attribute__((address_space(3))) float* p2_lds_f32;
__device__ __forceinline__ hx8 p2_frag(unsigned off) { return __builtin_bit_cast(hx8, *(p2_lds_u16x8)(unsigned long)off); }

// QK: this block walks the Q / K column blocks (n_walk_qk walkers, blockIdx.y < n_walk_qk) or the V / point blocks
//
// SPLIT: split operands (x = hi + lo, W = hi + lo, each part one half-precision value; W_hi x_hi + W_hi x_lo + W_lo x_hi with fp32
// accumulation: the accuracy of an fp32 product).  The q / k / v / point projections are per-residue quantities whose rounding
// errors are coherent over all keys of the attention (tests/err_budget.py at bb_gain 0.3: the largest single group of the half mode).
// Twice the weight bytes and three times the MFMAs, so the shape changes (first split version: the 128-row blocks of the plain
// kernel with the hi / lo blocks of a column block in the two LDS buffers, no double buffering: 50 us against 25 — every row block
// streams all 7 MB of hi + lo images, 133 MB per call, at the ~6.4 TB/s the L2 fabric delivers to 256 CUs' LDS-DMA, exposed):
//   * a block owns 256 rows and has EIGHT waves (wave w: rows 32 w .. 32 w + 31, both parts of its activation fragments in
//     registers: 128 of its 256 registers; four waves with 64 rows each need 256 registers for the fragments alone and the
//     compiler shuffles them through the accumulation registers) -> 10 row blocks stream 70 MB, two waves per SIMD;
//   * a step is HALF a column block (64 columns: hi 32 KB + lo 32 KB of fragments), double-buffered, 96 MFMAs per wave and step
//     (6.1 k matrix cycles per SIMD against ~3-5 k cycles of DMA for the next 64 KB);
//   * the bias vector of the whole projection sits in LDS (27 KB) instead of registers.
#define P2S_HALF 32768                      // one part (hi or lo) of a half column block: 2 tiles x 16 k-steps x 1 KB
#define P2S_STAGE (2 * P2S_HALF)            // hi | lo
#define P2S_BIAS_OFF (2 * P2S_STAGE)        // fp32 bias of every column (zero-padded to whole column blocks)
template <bool QK, bool SPLIT>
__device__ __forceinline__ void ipa_proj2_body(const ProjArgs& a, int n_cblk, int walker, int n_walkers, char* smem) {
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, hi = lane >> 5;
  constexpr int RB = SPLIT ? 256 : 128;     // rows per block
  constexpr int TM = SPLIT ? 1 : 2;         // 32-row tiles per wave
  constexpr int NTH = SPLIT ? 512 : FD_THREADS;
  const int wr = SPLIT ? wave : wave >> 1;  // row part of this wave: rows 32 TM wr .. of the block
  int wc = SPLIT ? 0 : (wave & 1);          // column half of a 128-column block: per wave (plain), per step (SPLIT)
  const bool mrg = a.merged != 0;  // columns [q' | points] instead of [q | k, v per head | points]
  const int M = a.B * a.N, HC = a.H * a.C, NOUT = (mrg ? HC : 3 * HC) + a.PT, ntl = a.Np >> 5;
  const int m0 = blockIdx.x * RB;
  const char* wimg = (const char*)a.W_img;
  const char* wimg_lo = (const char*)a.W_img_lo;
  // ---- column blocks of this block's class, in class order: k-th Q/K block / k-th V-or-point block -> column block index
  const int nq = HC / 128, per_head = (2 * a.C) / 128, n_class = mrg ? (QK ? nq : n_cblk - nq) : (QK ? 2 * nq : n_cblk - 2 * nq);
  auto cblk_of = [&](int k) {
    if (mrg) return QK ? k : nq + k;
    if (QK) return k < nq ? k : nq + ((k - nq) / (per_head / 2)) * per_head + (k - nq) % (per_head / 2);
    const int nv = nq;  // V blocks, then the point blocks
    return k < nv ? nq + (k / (per_head / 2)) * per_head + per_head / 2 + k % (per_head / 2) : 3 * nq + (k - nv);
  };
  int kb = walker;
  auto request = [&](int c, int buf) {  // plain: one 128-column block (64 KB)
    const char* src = wimg + (size_t)c * P2_WBLK;
#pragma unroll
    for (int u = 0; u < P2_WBLK / (FD_THREADS * 16); ++u)
      p2_dma16(src + (size_t)(u * FD_THREADS + tid) * 16, lds0 + buf * P2_WBLK + (unsigned)(u * FD_THREADS + (tid & ~63)) * 16);
  };
  auto request_half = [&](int c, int half, int buf) {  // SPLIT: tiles 2 half, 2 half + 1 of column block c, hi then lo part
    const size_t so = (size_t)c * P2_WBLK + (size_t)half * P2S_HALF;
#pragma unroll
    for (int u = 0; u < P2S_HALF / (NTH * 16); ++u) {
      const unsigned d = lds0 + buf * P2S_STAGE + (unsigned)(u * NTH + (tid & ~63)) * 16;
      p2_dma16(wimg + so + (size_t)(u * NTH + tid) * 16, d);
      p2_dma16(wimg_lo + so + (size_t)(u * NTH + tid) * 16, d + P2S_HALF);
    }
  };
  FD_STAMP(0);
  if (!SPLIT && kb < n_class) request(cblk_of(kb), 0);  // the first weight block is on its way while the activations are staged
  // ---- activation rows: fp32 -> half -> LDS (rows of 512 B, 16 B chunk c of row r at c ^ (r & 15)) -> fragments in registers, once.
  // plain: 128 rows in buffer 1.  SPLIT: the waves' private tiles fill both buffers (the weight requests start afterwards)
  hx8 Af[TM][P2_KS];
  hx8 Al[1][SPLIT ? P2_KS : 1];
  if constexpr (SPLIT) {
    // every wave stages its OWN 32 rows (read once: 32 float4 per lane, all in flight) through a private 16 KB LDS tile, the hi
    // parts and then — out of the same registers — the lo parts: no block barrier until the tiles are free for the weights, and
    // half the activation bytes of a two-pass staging (the fabric between L2 and the CUs is what bounds this kernel)
    f32x4 xv[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) {
      const int gr = m0 + wave * 32 + r < M ? m0 + wave * 32 + r : M - 1;
      xv[r] = *(const f32x4*)(a.A + (long)gr * a.lda + 4 * lane);
    }
    {  // the bias vector -> LDS, zero beyond NOUT: all requests of a thread in flight together (a rolled copy loop is one
       // dependent L2 round trip per iteration: 14 of them were 20 k cycles of prologue)
      float* bls = (float*)(smem + P2S_BIAS_OFF);
      constexpr int NBV = 16;  // 16 x 512 threads = 8192 >= 160 KB limit of columns checked by the launcher
      float bv[NBV];
#pragma unroll
      for (int k = 0; k < NBV; ++k) {
        const int v = tid + k * NTH;
        bv[k] = v < NOUT ? a.bias[v] : 0.f;
      }
#pragma unroll
      for (int k = 0; k < NBV; ++k)
        if (tid + k * NTH < n_cblk * 128) bls[tid + k * NTH] = bv[k];
    }
    char* xs = smem + wave * (32 * P2_XROW);
    const unsigned xl = lds0 + wave * (32 * P2_XROW);
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
#pragma unroll
      for (int r = 0; r < 32; ++r) {
        f32x4 x = xv[r];
        if (pass) {  // lo part: x - half(x), exact in fp32
#pragma unroll
          for (int q = 0; q < 4; ++q) x[q] -= h2f(f2h(x[q]));
        }
        const p2_u32x2 h = {fd_cvt_pk(x[0], x[1]), fd_cvt_pk(x[2], x[3])};
        *(p2_u32x2*)(xs + r * P2_XROW + (((lane >> 1) ^ (r & 15)) << 4) + 8 * (lane & 1)) = h;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the wave's own writes have landed (wave-private tile: no barrier)
#pragma unroll
      for (int s = 0; s < P2_KS; ++s) {
        const hx8 f = p2_frag(xl + li * P2_XROW + (((2 * s + hi) ^ (li & 15)) << 4));
        if (pass == 0) Af[0][s] = f;
        else Al[0][s] = f;
      }
      __builtin_amdgcn_s_waitcnt(0xC07F);  // the fragments are in registers before the tile is overwritten
    }
    __syncthreads();  // the tiles become the weight buffers
  } else {
    char* xs = smem + P2_WBLK;
#pragma unroll
    for (int it = 0; it < 32; ++it) {  // 128 rows x 64 float4 (all 32 requests of a thread in flight: one round trip)
      const int idx = tid + it * FD_THREADS, r = idx >> 6, c4 = idx & 63;
      const int gr = m0 + r < M ? m0 + r : M - 1;
      const f32x4 x = *(const f32x4*)(a.A + (long)gr * a.lda + 4 * c4);
      const p2_u32x2 h = {fd_cvt_pk(x[0], x[1]), fd_cvt_pk(x[2], x[3])};
      *(p2_u32x2*)(xs + r * P2_XROW + (((c4 >> 1) ^ (r & 15)) << 4) + 8 * (c4 & 1)) = h;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int s = 0; s < P2_KS; ++s) {
        const int r = (wr * 2 + i) * 32 + li;
        Af[i][s] = p2_frag(lds0 + P2_WBLK + r * P2_XROW + (((2 * s + hi) ^ (r & 15)) << 4));
      }
    __builtin_amdgcn_s_waitcnt(0xC07F);  // lgkmcnt(0): the fragments are in registers before the buffer is overwritten
  }
  FD_STAMP(1);
  // ---- store addressing, split into a part that depends on the row(s) of a register group (computed once) and a part that
  // depends on the column block (once per block): an epilogue unit adds the two and a compile-time constant
  //   Q / K images (lane = row): element ((((b H + h) ntl + (r >> 5)) (C >> 4) + (cc >> 4)) 64 + ((cc >> 3) & 1) 32 + (r & 31)) 8 + (cc & 7)
  //   V^T image (lane = channel, registers = 4 consecutive keys): ((((b H + h) (C >> 5) + (cc >> 5)) 2 ntl + (pp >> 4)) 64 +
  //                                                               ((pp >> 3) & 1) 32 + (cc & 31)) 8 + (pp & 7)
  //   points (fp32 rows): row * PT + column
  int qk_row[TM];      // QK: row part of tile i (this lane's row), -1 = row beyond M
  int v_row[TM][4];    // V / points: row part of register group (i, g) (rows mg .. mg + 3), -1 = beyond M
  int p_row[TM][4];
#pragma unroll
  for (int i = 0; i < TM; ++i) {
    const int m = m0 + (wr * TM + i) * 32 + li;
    const int b = m < M ? m / a.N : 0, r = m - b * a.N;
    qk_row[i] = m < M ? ((b * a.H * ntl + (r >> 5)) * (a.C >> 4) * 64 + (r & 31)) * 8 : -1;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int mg = m0 + (wr * TM + i) * 32 + 8 * g + 4 * hi;
      const int bb = mg < M ? mg / a.N : 0, key = mg - bb * a.N, pp = (key & ~15) + p2_perm16(key & 15);
      v_row[i][g] = mg + 3 < M ? (bb * a.H * (a.C >> 5) * (2 * ntl) + (pp >> 4)) * 512 + ((pp >> 3) & 1) * 256 + (pp & 7) : -1;
      p_row[i][g] = mg + 3 < M ? mg * a.PT : -1;
    }
  }
  // One 4-run (registers 4 g .. 4 g + 3) of tile (i, j) of a column block: the epilogue is cut into these 16 units so that the
  // units of the PREVIOUS column block ride under the 16 k-steps of the current one (with one wave per SIMD nothing else
  // would overlap the stores; and stores issued a whole column block before the next wait never stall it).
  // cpart: column part of the block (QK: head; V: per j (head, channel of this lane); points: column of this lane, or -1)
  // BQ: bias of the column block in the layout the unit wants: q(j, g) f32x4 (Q / K blocks: permuted rows), s(j) (others)
  auto epi_unit = [&](const f32x16 (&acc)[TM][2], const auto& BQ, const int (&cpart)[2], int kind, int i, int j, int g) {
    if constexpr (QK) {
      // the rows of the Q / K weight tiles are permuted in the image (fd_ipa_proj2_permute_image) so that register r of a lane
      // is channel 16 (r >> 3) + 8 hi + (r & 7) of the tile: 8 consecutive channels = one whole 16 B fragment unit per lane and
      // 16-group G = g >> 1; the odd g of a pair has nothing left to do
      if ((g & 1) || qk_row[i] < 0) return;
      const f32x4 b0 = BQ.q(j, g), b1 = BQ.q(j, g + 1);
      const float sc = kind == 0 ? a.qscale : 1.f;
      const p2_u32x4 o = {fd_cvt_pk((acc[i][j][4 * g] + b0[0]) * sc, (acc[i][j][4 * g + 1] + b0[1]) * sc),
                          fd_cvt_pk((acc[i][j][4 * g + 2] + b0[2]) * sc, (acc[i][j][4 * g + 3] + b0[3]) * sc),
                          fd_cvt_pk((acc[i][j][4 * g + 4] + b1[0]) * sc, (acc[i][j][4 * g + 5] + b1[1]) * sc),
                          fd_cvt_pk((acc[i][j][4 * g + 6] + b1[2]) * sc, (acc[i][j][4 * g + 7] + b1[3]) * sc)};
      // cc = cbase + (2 wc + j) 32 + 16 G + 8 hi: cc >> 4 = (cbase >> 4) + 2 (2 wc + j) + G, (cc >> 3) & 1 = hi, cc & 7 = 0
      half_t* dst = (kind == 0 ? a.Qb : a.Kb) + (long)qk_row[i] + cpart[0] + ((2 * j + (g >> 1)) * 64) * 8;
#ifdef P2_NARROW_STORES
      if constexpr (SPLIT) {
        asm volatile("global_store_dword %0, %1, off\n\tglobal_store_dword %0, %2, off offset:4\n\tglobal_store_dword %0, %3, off offset:8\n\tglobal_store_dword %0, %4, off offset:12"
                     : : "v"(dst), "v"(o[0]), "v"(o[1]), "v"(o[2]), "v"(o[3]) : "memory");
      } else
#endif
      *(p2_u32x4*)dst = o;
    } else {
      if (cpart[j] < 0) return;
      const float bv = BQ.s(j);
      float v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = acc[i][j][4 * g + q] + bv;
      if (kind == 2) {
        if (v_row[i][g] < 0) return;
        const p2_u32x2 o = {fd_cvt_pk(v[0], v[1]), fd_cvt_pk(v[2], v[3])};
        *(p2_u32x2*)(a.Vt + (long)v_row[i][g] + cpart[j]) = o;
        if (SPLIT && a.Vt_lo) {  // V - half(V): the attention multiplies P with V_hi + V_lo (attention3.hip)
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
  // kind (0 Q, 1 K, 2 V, 3 points) and column part of column block c (of its half `wc`)
  auto col_part = [&](int c, int (&cpart)[2]) {
    const int n0 = c * 128;
    if constexpr (QK) {
      const bool isq = n0 < HC;
      const int nn0 = isq ? n0 : n0 - HC, hh = isq ? nn0 / a.C : nn0 / (2 * a.C), cbase = isq ? nn0 % a.C : nn0 % (2 * a.C);
      cpart[0] = (hh * ntl * (a.C >> 4) * 64 + ((cbase >> 4) + 4 * wc) * 64 + hi * 32) * 8;
      cpart[1] = 0;
      return isq ? 0 : 1;
    } else {
      const bool isv = !mrg && n0 < 3 * HC;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int n = n0 + (wc * 2 + j) * 32 + li;
        if (isv) {
          const int nn = n - HC, hh = nn / (2 * a.C), cc = nn % (2 * a.C) - a.C;
          cpart[j] = (hh * (a.C >> 5) + (cc >> 5)) * (2 * ntl) * 512 + (cc & 31) * 8;
        } else cpart[j] = n < NOUT ? n - (mrg ? HC : 3 * HC) : -1;
      }
      return isv ? 2 : 3;
    }
  };
  if constexpr (SPLIT) {
    // bias from LDS: nb = first column of the half block (c * 128 + 64 wc)
    struct BiasLds {
      unsigned base; int hi, li;
      __device__ __forceinline__ f32x4 q(int j, int g) const { return *(p2_lds_f32x4)(unsigned long)(base + 4 * (j * 32 + 16 * (g >> 1) + 8 * hi + 4 * (g & 1))); }
      __device__ __forceinline__ float s(int j) const { return *(p2_lds_f32)(unsigned long)(base + 4 * (j * 32 + li)); }
    };
    const int n_hs = kb < n_class ? 2 * ((n_class - 1 - kb) / n_walkers + 1) : 0;  // half-steps of this walker
    auto cb_of_hs = [&](int hs) { return cblk_of(kb + (hs >> 1) * n_walkers); };
    if (n_hs > 0) request_half(cb_of_hs(0), 0, 0);
    // Work unit = one 32-column tile of a half block (a "quarter step": 48 MFMAs per wave).  The two tiles of a half block are the
    // two accumulators; the 4 epilogue units of the previous tile ride under the k-steps of the current one (also across the step
    // barrier), and the fragments run two k-steps ahead — so every wave always has matrix work in flight and the two waves of a SIMD
    // keep its core busy (first version: products, then the epilogue, in lockstep on all waves: 10 k cycles per half block for
    // 6.1 k of matrix work; the two waves of a SIMD in opposite phase: the same, the lone wave's products waited for its LDS reads).
    struct HalfBlock { int cp[2]; int kind; unsigned nb; };
    HalfBlock cur = {{0, 0}, 0, 0u}, prev = {{0, 0}, 0, 0u};
    f32x16 acc[1][2];
    auto quarter = [&](int hs, auto J) {
      constexpr int j = decltype(J)::value;
      if (j == 0) {
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this thread's share of the half block (requested a step ago) has landed
        __syncthreads();
        if (hs + 1 < n_hs) request_half(cb_of_hs(hs + 1), (hs + 1) & 1, (hs + 1) & 1);
        prev = cur;
        const int c = cb_of_hs(hs);
        wc = hs & 1;
        cur.kind = col_part(c, cur.cp);
        cur.nb = lds0 + P2S_BIAS_OFF + 4u * (unsigned)(c * 128 + 64 * wc);
      }
      const HalfBlock& ep = j == 0 ? prev : cur;  // the half block of the previous tile
      const bool has_prev = j == 1 || hs > 0;
      const BiasLds bq = {ep.nb, hi, li};
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[0][j][r] = 0.f;
      const unsigned wbh = lds0 + (hs & 1) * P2S_STAGE + j * (P2_KS * 1024) + lane * 16, wbl = wbh + P2S_HALF;
      constexpr int DEPTH = 3;
      hx8 wh[DEPTH], wl[DEPTH];
#pragma unroll
      for (int s = 0; s < DEPTH - 1; ++s) { wh[s] = p2_frag(wbh + s * 1024); wl[s] = p2_frag(wbl + s * 1024); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < P2_KS; ++s) {
        if (s + DEPTH - 1 < P2_KS) {
          wh[(s + DEPTH - 1) % DEPTH] = p2_frag(wbh + (s + DEPTH - 1) * 1024);
          wl[(s + DEPTH - 1) % DEPTH] = p2_frag(wbl + (s + DEPTH - 1) * 1024);
        }
        const hx8 h = wh[s % DEPTH], l = wl[s % DEPTH];
        if constexpr (QK) {  // operands exchanged: lane = row.  Small terms first, then the hi x hi product
          acc[0][j] = fd_mfma32(l, Af[0][s], acc[0][j]);
          acc[0][j] = fd_mfma32(h, Al[0][s], acc[0][j]);
          acc[0][j] = fd_mfma32(h, Af[0][s], acc[0][j]);
        } else {
          acc[0][j] = fd_mfma32(Af[0][s], l, acc[0][j]);
          acc[0][j] = fd_mfma32(Al[0][s], h, acc[0][j]);
          acc[0][j] = fd_mfma32(Af[0][s], h, acc[0][j]);
        }
        if (has_prev && (s & 3) == 1) epi_unit(acc, bq, ep.cp, ep.kind, 0, j ^ 1, s >> 2);
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    for (int hs = 0; hs < n_hs; ++hs) {
      quarter(hs, std::integral_constant<int, 0>{});
      FD_STAMP(2 + 2 * hs);
      quarter(hs, std::integral_constant<int, 1>{});
      FD_STAMP(3 + 2 * hs);
    }
    if (n_hs > 0) {
      const BiasLds bq = {cur.nb, hi, li};
#pragma unroll
      for (int g = 0; g < 4; ++g) epi_unit(acc, bq, cur.cp, cur.kind, 0, 1, g);
    }
    FD_STAMP(15);
    return;
  }
  if constexpr (!SPLIT) {
  // bias of a column block in the layout its epilogue wants (requested when the block's MFMAs start, used a block later:
  // a bias load inside an epilogue unit would wait for itself AND, vmcnt being in order, for the whole weight DMA before it)
  struct BiasReg {
    f32x4 v[2][4];
    __device__ __forceinline__ f32x4 q(int j, int g) const { return v[j][g]; }
    __device__ __forceinline__ float s(int j) const { return v[j][0][0]; }
  };
  auto load_bias = [&](BiasReg& bq, int n0) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (QK) bq.v[j][g] = *(const f32x4*)(a.bias + n0 + (wc * 2 + j) * 32 + 16 * (g >> 1) + 8 * hi + 4 * (g & 1));  // (permuted rows)
        else if (g == 0) {
          const int n = n0 + (wc * 2 + j) * 32 + li;
          bq.v[j][0][0] = n < NOUT ? a.bias[n] : 0.f;
        }
      }
  };
  auto step = [&](f32x16 (&accN)[2][2], BiasReg& bqN, int (&cpN)[2], int& kindN, const f32x16 (&accP)[2][2], const BiasReg& bqP,
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
  f32x16 acc0[2][2], acc1[2][2];
  BiasReg bq0, bq1;
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
}
template <bool SPLIT>
__global__ __launch_bounds__(SPLIT ? 512 : FD_THREADS, 1) void ipa_proj2_kernel(ProjArgs a, int n_cblk, int n_walk_qk) {
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
int fd_ipa_proj2_permute_image_q(void* img, int H, int C, int K, hipStream_t st) {
  if ((C & 31) || (K & 15)) return FDIPT_EINVAL;
  const int n_q = H * C / 32, ks = K / 16;
  hipLaunchKernelGGL(p2_permute_rows_kernel, dim3(n_q * ks), dim3(64), 0, st, (half_t*)img, ks, n_q, 2 * C / 32);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// ------------------------------------------------------------------ node rows -> attention operand images (merged projection)
// One thread = one 16 B unit of an image.  Units [0, nk): Kb (key r, channels 8 cg .. 8 cg + 7: contiguous in the row);
// units [nk, nk + nv): Vt and Vt_lo (channel cc, the 8 key slots of a half 16-group: eight coalesced row reads, lanes = channels).
__global__ void node_images_kernel(int B, int N, int Np, const float* __restrict__ node, int ld, half_t* __restrict__ Kb,
                                   half_t* __restrict__ Vt, half_t* __restrict__ Vt_lo) {
  constexpr int C = 256;
  const int ntl = Np >> 5;
  const long nk = (long)B * Np * (C / 8), nv = (long)B * (C / 32) * (2 * ntl) * 64;
  for (long u = blockIdx.x * (long)blockDim.x + threadIdx.x; u < nk + nv; u += (long)gridDim.x * blockDim.x) {
    if (u < nk) {
      const int cg = (int)(u % (C / 8));
      const long br = u / (C / 8);
      const int r = (int)(br % Np), b = (int)(br / Np);
      p2_u32x4 o = {0u, 0u, 0u, 0u};
      if (r < N) {
        const float* x = node + ((long)b * N + r) * ld + 8 * cg;
        const f32x4 x0 = *(const f32x4*)x, x1 = *(const f32x4*)(x + 4);
        o = p2_u32x4{fd_cvt_pk(x0[0], x0[1]), fd_cvt_pk(x0[2], x0[3]), fd_cvt_pk(x1[0], x1[1]), fd_cvt_pk(x1[2], x1[3])};
      }
      // element ((((b ntl + (r >> 5)) (C >> 4) + (cc >> 4)) 64 + ((cc >> 3) & 1) 32 + (r & 31)) 8 + (cc & 7), cc = 8 cg
      *(p2_u32x4*)(Kb + ((((long)b * ntl + (r >> 5)) * (C >> 4) + (cg >> 1)) * 64 + (cg & 1) * 32 + (r & 31)) * 8) = o;
    } else {
      const long v = u - nk;
      const int lane = (int)(v & 63), half = lane >> 5, c5 = lane & 31;
      const long f = v >> 6;
      const int s16 = (int)(f % (2 * ntl));
      const long bd = f / (2 * ntl);
      const int dt = (int)(bd % (C / 32)), b = (int)(bd / (C / 32));
      const int cc = 32 * dt + c5;
      float xv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {  // slot 8 half + e of the 16-group -> key 16 s16 + pos, slot = 4 (pos >> 3) + (pos & 3) + 8 ((pos & 7) >> 2)
        const int slot = 8 * half + e, pos = 8 * ((slot >> 2) & 1) + 4 * (slot >> 3) + (slot & 3);
        const int key = 16 * s16 + pos;
        xv[e] = key < N ? node[((long)b * N + key) * ld + cc] : 0.f;
      }
      p2_u32x4 oh, ol;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        oh[q] = fd_cvt_pk(xv[2 * q], xv[2 * q + 1]);
        ol[q] = fd_cvt_pk(xv[2 * q] - h2f(f2h(xv[2 * q])), xv[2 * q + 1] - h2f(f2h(xv[2 * q + 1])));
      }
      *(p2_u32x4*)(Vt + v * 8) = oh;
      if (Vt_lo) *(p2_u32x4*)(Vt_lo + v * 8) = ol;
    }
  }
}
int fd_node_images(int B, int N, int Np, const float* node, int ld, half_t* Kb, half_t* Vt, half_t* Vt_lo, hipStream_t st) {
  if ((Np & 31) || Np < N || (ld & 3) || !Kb || !Vt) return FDIPT_EINVAL;
  const long units = (long)B * Np * 32 + (long)B * 8 * (Np / 16) * 64;
  hipLaunchKernelGGL(node_images_kernel, dim3((unsigned)((units + 255) / 256)), dim3(256), 0, st, B, N, Np, node, ld, Kb, Vt, Vt_lo);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

int fd_ipa_proj2_supported(const ProjArgs& a) {
  const int HC = a.H * a.C;
  return a.K == P2_K && a.W_img && (a.N & 3) == 0 && (a.C % 128) == 0 && (HC % 128) == 0 && (a.lda & 3) == 0 && (a.Np & 31) == 0 && (a.PT & 3) == 0;
}
int fd_ipa_proj2(const ProjArgs& a, hipStream_t st) {
  if (!fd_ipa_proj2_supported(a)) return FDIPT_EINVAL;
  const bool split = a.W_img_lo != nullptr;
  const int M = a.B * a.N, NOUT = (a.merged ? 1 : 3) * a.H * a.C + a.PT;
  const int n_cblk = cdiv(NOUT, 128), n_rblk = cdiv(M, split ? 256 : 128);
  const int lds = split ? P2S_BIAS_OFF + n_cblk * 128 * 4 : P2_LDS;
  if (lds > 160 * 1024 || (split && n_cblk * 128 > 8192)) return FDIPT_ESIZE;
  // (the attribute is per device: set on every launch — a host-side table lookup — rather than cached per process)
  if (hipFuncSetAttribute(split ? (const void*)ipa_proj2_kernel<true> : (const void*)ipa_proj2_kernel<false>,
                          hipFuncAttributeMaxDynamicSharedMemorySize, lds) != hipSuccess)
    return FDIPT_ELAUNCH;
  // one block per CU: about 256 / row-blocks column walkers per row block, split between the two column classes in
  // proportion to their column blocks
  int ncg = 256 / n_rblk;
  if (ncg < 2) ncg = 2;
  const int n_qk = (a.merged ? 1 : 2) * (a.H * a.C / 128), n_other = n_cblk - n_qk;
  // (a Q / K column block costs about 6 / 7 of a V / point block: 16 B stores against 8 B / 4 B ones)
  int wq = (ncg * n_qk * 6 + (n_qk * 6 + n_other * 7) / 2) / (n_qk * 6 + n_other * 7);
  if (wq < 1) wq = 1;
  if (wq > ncg - 1) wq = ncg - 1;
  if (wq > n_qk) wq = n_qk;
  int wo = ncg - wq;
  if (wo > n_other) wo = n_other;
  if (split) hipLaunchKernelGGL(ipa_proj2_kernel<true>, dim3(n_rblk, wq + wo), dim3(512), lds, st, a, n_cblk, wq);
  else hipLaunchKernelGGL(ipa_proj2_kernel<false>, dim3(n_rblk, wq + wo), dim3(FD_THREADS), lds, st, a, n_cblk, wq);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
