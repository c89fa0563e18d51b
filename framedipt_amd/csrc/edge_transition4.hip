// edge_transition4.hip — bf16 EdgeTransition (framedipt/model/ipa_pytorch.py:84-102), third generation.
//
// edge_transition3 (16-pair waves, v_mfma_f32_16x16x32_bf16) needs a fresh 1 KB weight fragment from LDS for every 16-cycle
// MFMA: at full matrix rate the four SIMDs ask for exactly the 256 B/clk the LDS delivers, and the kernel sits on that
// limit (42 % MFMA utilisation).  This generation halves the LDS bytes per FLOP and removes a fifth of the FLOPs:
//   * a wave owns 32 pairs and works with v_mfma_f32_32x32x16_bf16 (32 cycles per 1 KB fragment), still TWO waves per SIMD:
//     the activations fit 256 registers because x is no longer held — the z fragments are re-read from the wave's LDS rows
//     for the final layer, and e_j is gone from the matrix products altogether (next point);
//   * a wave's 32 pairs are a patch of 8 rows i x 4 columns j of one sample.  Everything in the first and final layers that
//     depends on one residue only — W[:, e_i cols] e_i + b (rows A1[i], Af[i]) AND W[:, e_j cols] e_j (rows B1[j], Bf[j])
//     — enters the accumulator through ONE extra k-step per output tile: A = [A1 rows of the 8 i | B1 rows of the 4 j] as
//     a bf16 fragment straight from L2, B = a constant 0/1 selection matrix (pair p picks i = p >> 2 and j = p & 3).
//     Layer 1 shrinks from K = 256 to 128 + 16, the final layer from 640 to 512 + 16: 536 MFMA x 32 cycles per 32 pairs
//     instead of 644 x 16 per 16 pairs (-17 % matrix cycles, -60 % LDS fragment bytes);
//   * same transposed scheme and register hand-off: D^T[feature, pair]; the C/D registers 8u .. 8u+7 of a 32-feature tile
//     are the B fragment of k-step 2T + u of the next layer, k position (half, e) = feature 32 T + e4_chain_feat(u, half, e),
//     folded into the weight stream;
//   * weight stream 512 KB per 256 pairs (edge_transition3: 640 KB per 128), 20 chunks through a 2 x 32 KB LDS ring.
// Needs N % 4 == 0; other sizes run edge_transition3.
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "kernels.hpp"

#ifndef E4_ABL
#define E4_ABL 0  // timing ablations (tools/micro/et4_bench.hip): 1 no epilogue, 2 no MFMA, 4 no weight DMA, 8 no z' / bias stores, 16 (flat) no LDS fragment reads,
                  // 64 layer 1's chunks are not streamed, 32 no LayerNorm arithmetic in the epilogue (statistics and normalisation skipped: conversions, staging, stores, products stay)
#endif
#ifndef E4_PZ_ABL
#define E4_PZ_ABL 0  // timing ablations of the pair_z emission (wrong results): 1 no lo part, 2 lo part from the hi image in LDS (no L2 loads), 4 no pair_z stores, 8 no hi part
#endif
#ifndef E4_D1
#define E4_D1 3   // weight-fragment ring depths: layer 1, layer 2, final layer
#endif
#ifndef E4_D2
#define E4_D2 3
#endif
#ifndef E4_DF
#define E4_DF 6
#endif
#define E4_CZ 128
#define E4_H 384
#ifndef E4_WAVES
#define E4_WAVES 8  // waves per block (4: experiment builds of tools/micro/et4_bench.hip)
#endif
#define E4_THREADS (64 * E4_WAVES)
#ifdef E4_FAKE2  // TIMING ONLY (wrong results): the LDS footprint of a two-blocks-per-CU design with the present chunk sizes —
#define E4_BUF 16384   // the chunks overlap each other and the z rows
#else
#define E4_BUF 32768
#endif
#define E4_WBI (2048 + 64)  // linear_b image, compact (8 head rows) | 16 B of zeros (+ pad)   (down_z: the stream's last chunk, ring slot 3)
#define E4_L1_FR (12 * 8)    // fragments (1 KB): layer 1, 12 tiles x 8 k-steps (K = 128: z)
#define E4_L2_FR (12 * 24)   // layer 2, 12 tiles x 24 k-steps
// Round 6: the reference is final_layer(trunk(x) + x) (ipa_pytorch.py:99): the z part of x meets the SAME weight columns as the first 128
// features of h2, so z is added to those features when they are handed over (e4_add_z) and the final layer runs 24 k-steps instead of
// 8 (z) + 24 (h2): 504 products per 32-pair tile instead of 536.  The stream keeps its 512 fragments = 32 chunks (the ring of four chunk
// slots needs a multiple of four): fragments 480 .. 495 are unused, 496 .. 511 (chunk 31 = ring slot 3, resident through the epilogue)
// hold down_z of the next block, hi and lo images (fd_et4_set_dz) — the fragments the pair_z emission reads.
#define E4_LF_FR (24 * 4)    // final layer, k-major: 24 k-steps of (h2 + x) x 4 tiles
#define E4_DZ_FR0 496        // down_z hi [8] | lo [8]
#define E4_STREAM_FR 512
#define E4_STREAM_BYTES (E4_STREAM_FR * 1024)
static_assert(E4_L1_FR + E4_L2_FR + E4_LF_FR == 480, "480 weight fragments, then one unused chunk and the down_z chunk");
#define E4_ZOFF (2 * E4_BUF)                 // per-wave z rows [8][32 rows x 256 B]
#define E4_VOFF (E4_ZOFF + E4_WAVES * 8192)         // b2[384] | gamma[128] | beta[128] | down_z bias [32] f32, then the epilogue's images (E4_WBI)
#define E4_VEC_BYTES (1536 + 1024 + 128)
#define E4_SOFF (E4_VOFF + E4_VEC_BYTES + E4_WBI)  // per-wave output staging [8][32 rows x 64 B]
#define E4_MOFF (E4_SOFF + E4_WAVES * 2048)          // per-lane pair masks of the current tile [512] f32
#define E4_BOFF (E4_MOFF + E4_THREADS * 4)              // bias of linear_b [8] f32
#define E4_LDS (E4_BOFF + 32)

// phase profile (-DE4_PROF, tools/micro/et4_bench.hip): cycle differences accumulate in scalar registers over all tiles of a
// block and are written once at the end (FD_STAMP's per-stamp vector store costs registers this kernel does not have)
#ifdef E4_PROF
__device__ unsigned e4_prof[256 * 8];
__device__ unsigned long long e4_span[1024 * 3];  // per block: start, end (s_memrealtime), HW_ID
#define E4_STAMP(k)                                              \
  do {                                                           \
    const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); \
    ph[k] += t_ - tlast;                                         \
    tlast = t_;                                                  \
  } while (0)
#else
#define E4_STAMP(k) \
  do {              \
  } while (0)
#endif
// Shader clock actually sustained inside the kernel (the matrix peak scales with it: under dense MFMA load the chip runs well below
// its 2.4 GHz nominal clock, and how far below depends on the operand bits — tools/micro/et4_bench.hip): common.hpp FD_CLK_*,
// into the caller's FdiptForwardArgs.clock_out (ET2Args.clock) when that is set; no atomics and no state otherwise.
#define E4_CLK_BEGIN FD_CLK_BEGIN
#define E4_CLK_END FD_CLK_END(a.clock)
typedef fd_h e4_hx4 __attribute__((ext_vector_type(4)));
typedef unsigned int e4_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int e4_u32x2 __attribute__((ext_vector_type(2)));
typedef short e4_s16x2 __attribute__((ext_vector_type(2)));

// C/D register 8u + e of lane half `half` of a 32-feature tile holds feature (offset in the tile):
__host__ __device__ __forceinline__ int e4_chain_feat(int u, int half, int e) { return (e & 3) + 8 * (2 * u + (e >> 2)) + 4 * half; }

// ------------------------------------------------------------------ prepare: weight stream image
// w1 [384,384], w2 [384,384], wf [128,384] fp32 row-major (out, in); in = [z(0:128) | e_i(128:256) | e_j(256:384)]
__global__ void et4_build_stream_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                        const float* __restrict__ wf, half_t* __restrict__ stream) {
  const int n_units = E4_STREAM_BYTES / 16;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n_units; g += gridDim.x * blockDim.x) {
    int frag = g >> 6;
    const int lane = g & 63, f = lane & 31, half = lane >> 5;
    const float* src;
    int n, s;
    bool chained;
    if (frag < E4_L1_FR) { src = w1; n = 32 * (frag / 8) + f; s = frag % 8; chained = false; }
    else if (frag < E4_L1_FR + E4_L2_FR) { frag -= E4_L1_FR; src = w2; n = 32 * (frag / 24) + f; s = frag % 24; chained = true; }
    else if (frag < E4_L1_FR + E4_L2_FR + E4_LF_FR) {
      frag -= E4_L1_FR + E4_L2_FR;
      src = wf; n = 32 * (frag & 3) + f; s = frag >> 2;
      chained = true;
    } else {  // unused chunk / down_z chunk (fd_et4_set_dz): zeros
      for (int e = 0; e < 8; ++e) stream[(long)g * 8 + e] = 0;
      continue;
    }
    half_t out[8];
    for (int e = 0; e < 8; ++e) {
      const int col = chained ? 32 * (s >> 1) + e4_chain_feat(s & 1, half, e) : 16 * s + 8 * half + e;  // z columns are 0..127
      out[e] = f2h(src[(long)n * E4_H + col]);
    }
    for (int e = 0; e < 8; ++e) stream[(long)g * 8 + e] = out[e];
  }
}
int fd_et4_build_stream(const float* w1, const float* w2, const float* wf, void* stream, hipStream_t st) {
  hipLaunchKernelGGL(et4_build_stream_kernel, dim3(128), dim3(256), 0, st, w1, w2, wf, (half_t*)stream);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
size_t fd_et4_stream_bytes() { return E4_STREAM_BYTES; }
// down_z of the NEXT block's IPA into the stream's last chunk: img_hi / img_lo = fd_chain_build_image_ex(Wdz, 32, 128, permuted = 1, lo = 0 / 1), 8 KB each
int fd_et4_set_dz(void* stream, const void* img_hi, const void* img_lo, hipStream_t st) {
  char* dst = (char*)stream + (size_t)E4_DZ_FR0 * 1024;
  if (hipMemcpyAsync(dst, img_hi, 8192, hipMemcpyDeviceToDevice, st) != hipSuccess || hipMemcpyAsync(dst + 8192, img_lo, 8192, hipMemcpyDeviceToDevice, st) != hipSuccess)
    return FDIPT_ELAUNCH;
  return FDIPT_OK;
}

// linear_b of the NEXT block's attention (ipa_pytorch.py:247,256-257) as 8 A fragments of a 32-row tile of which only the H <= 8 head
// rows exist: [k-step][lane half][8 rows][8] = 2 KB (round 6: the other 24 rows were 6 KB of zeros in LDS; lanes f >= 8 read one shared
// 16 B zero unit instead), k in the hand-off order of the LayerNorm output tiles; `scale` = sqrt(1/3)
__global__ void et4_bias_image_kernel(const float* __restrict__ wb, int H, float scale, half_t* __restrict__ img) {
  for (int g = threadIdx.x; g < 8 * 16; g += blockDim.x) {
    const int s = g >> 4, half = (g >> 3) & 1, f = g & 7;
    for (int e = 0; e < 8; ++e)
      img[g * 8 + e] = f < H ? f2h(wb[f * E4_CZ + 32 * (s >> 1) + e4_chain_feat(s & 1, half, e)] * scale) : (half_t)0;
  }
}
int fd_et4_build_bias_image(const float* wb, int H, float scale, void* img, hipStream_t st) {
  if (H > 8) return FDIPT_ESIZE;
  hipLaunchKernelGGL(et4_bias_image_kernel, dim3(1), dim3(256), 0, st, wb, H, scale, (half_t*)img);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// ------------------------------------------------------------------ per-forward: residue rows -> fold fragments
// rows [B*N][1024] f32 = [A1 (384) | Af (128) | B1 (384) | Bf (128)]   (A*: e_i columns + bias, B*: e_j columns)
// a_img [ceil(B*N/8)][16 feature tiles][32 f][8]: element e = flattened residue row 8 rt + e (0 beyond B*N)
// b_img [B][N/4][16][32 f][8]: element e < 4 = row j = 4 jt + e of sample b, e >= 4 = row 4 jt + e - 4 of sample b + 1
//                             (for the rows of a patch that straddles two samples; 0 for the last sample)
__global__ void et4_row_images_kernel(const float* __restrict__ rows, int B, int N, half_t* __restrict__ a_img,
                                      half_t* __restrict__ b_img) {
  const int M = B * N, MT8 = (M + 7) >> 3, NJ4 = N >> 2;
  const long na = (long)MT8 * 512, nb = (long)B * NJ4 * 512;
  for (long g = (long)blockIdx.x * blockDim.x + threadIdx.x; g < na + nb; g += (long)gridDim.x * blockDim.x) {
    const bool is_b = g >= na;
    const long u = is_b ? g - na : g;
    const int c = (int)(u & 511);                 // column 32 ft + f of the 512-wide half
    const long t = u >> 9;
    const int b = is_b ? (int)(t / NJ4) : 0, jt = is_b ? (int)(t - (long)b * NJ4) : 0;
    u16x8 o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      long r;
      bool ok;
      if (is_b) { const int bb = b + (e >> 2); ok = bb < B; r = (long)bb * N + 4 * jt + (e & 3); }
      else { r = 8 * t + e; ok = r < M; }
      o[e] = ok ? f2h(rows[r * 1024 + (is_b ? 512 : 0) + c]) : (half_t)0;
    }
    *(u16x8*)((is_b ? b_img : a_img) + u * 8) = o;
  }
}
size_t fd_et4_a_image_bytes(int B, int N) { return (size_t)((B * N + 7) / 8) * 8192; }
size_t fd_et4_b_image_bytes(int B, int N) { return (size_t)B * (N / 4) * 8192; }
int fd_et4_row_images(const float* rows, int B, int N, void* a_img, void* b_img, hipStream_t st) {
  const long units = ((long)(B * N + 7) / 8 + (long)B * (N / 4)) * 512;
  hipLaunchKernelGGL(et4_row_images_kernel, dim3((unsigned)cdiv(units, 256)), dim3(256), 0, st, rows, B, N, (half_t*)a_img, (half_t*)b_img);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// ------------------------------------------------------------------ device helpers
// 16 B-per-lane LDS-DMA as inline asm (see edge_transition2.hip: the builtin makes hipcc force lgkmcnt(0) everywhere)
// LDS is addressed by 32-bit byte offsets into the dynamic segment (the only LDS of the kernel, so it starts at 0): no
// generic pointers, no address-space casts with their null checks
typedef const __attribute__((address_space(3))) u16x8* e4_lds_u16x8;
typedef const __attribute__((address_space(3))) f32x4* e4_lds_f32x4;
__device__ __forceinline__ void e4_dma16(const void* gsrc, unsigned lds_dst) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(gsrc) : "memory", "m0");
}
// Cache policy of the pair stream (z rows in, z' rows out: 2 x 184 MB per launch at N = 300, B = 8, read / written exactly once),
// E4_ZPOL bit 0: z loads `nt`, bit 1: z' stores `sc1` (write-through: the line does not stay in the XCD's L2), bit 2: z' stores `nt`.
// The 512 KB weight stream every block re-reads 11 times per launch should stay L2-resident next to it (round 2: re-fetched ~38
// times per launch from the Infinity Cache: profiles/r02_pmc_bench_c4_fp16.md).
#ifndef E4_ZPOL
#define E4_ZPOL 0
#endif
__device__ __forceinline__ void e4_dma16_z(const void* gsrc, unsigned lds_dst) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane(lds_dst);
  if (E4_ZPOL & 1) asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt" : : "s"(m0v), "v"(gsrc) : "memory", "m0");
  else asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(gsrc) : "memory", "m0");
}
__device__ __forceinline__ void e4_store_z(half_t* dst, u16x8 v) {
  if (E4_ZPOL & 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" : : "v"(dst), "v"(v) : "memory");
  else if (E4_ZPOL & 4) __builtin_nontemporal_store(v, (u16x8*)dst);
  else *(u16x8*)dst = v;
}
__device__ __forceinline__ void e4_dma_wait() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ hx8 e4_frag(unsigned off) { return __builtin_bit_cast(hx8, *(e4_lds_u16x8)(unsigned long)off); }
__device__ __forceinline__ f32x4 e4_ldsf4(unsigned off) { return *(e4_lds_f32x4)(unsigned long)off; }
__device__ __forceinline__ hx8 e4_gfrag(const char* p) { return __builtin_bit_cast(hx8, *(const u16x8*)p); }
template <int BYTES>
__device__ __forceinline__ void e4_dma_chunk(const char* __restrict__ src, unsigned dst, int tid, int wave) {
  static_assert(BYTES % (E4_THREADS * 16) == 0, "whole DMA instructions");
#pragma unroll
  for (int u = 0; u < BYTES / (E4_THREADS * 16); ++u)
    if (!(E4_ABL & 4)) e4_dma16(src + (size_t)(u * E4_THREADS + tid) * 16, dst + (unsigned)(u * E4_THREADS + wave * 64) * 16);  // (scalar destination)
}
__device__ __forceinline__ f32x16 e4_mfma(hx8 a, hx8 b, f32x16 c) {
  if (E4_ABL & 2) { c[0] += (float)a[0]; return c; }
  return fd_mfma32(a, b, c);
}
// relu + bf16: C/D of one tile -> the two B fragments it hands to the next layer.  ReLU runs after the conversion, on the
// bf16 bit patterns as signed 16-bit integers (negative values have the sign bit set): one v_pk_max_i16 per two values.
__device__ __forceinline__ void e4_hand_off(const f32x16& acc, hx8& h0, hx8& h1) {
  e4_u32x4 w0, w1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    w0[k] = fd_cvt_pk(acc[2 * k], acc[2 * k + 1]);
    w1[k] = fd_cvt_pk(acc[8 + 2 * k], acc[8 + 2 * k + 1]);
  }
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  h0 = __builtin_bit_cast(hx8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, w0), zero));
  h1 = __builtin_bit_cast(hx8, __builtin_elementwise_max(__builtin_bit_cast(s16x8, w1), zero));
  __builtin_amdgcn_sched_barrier(0);  // the hand-off of a tile happens here, not batched with later tiles' (register pressure)
}

// one 32-feature tile: KS weight fragments at `pa` (this lane's 16 B of fragment 0) against B fragments Bf[0..KS)
template <int KS, int DEPTH>
__device__ __forceinline__ void e4_tile(f32x16& acc, unsigned pa, const hx8* Bf) {
  hx8 r[DEPTH];
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) r[s] = e4_frag(pa + s * 1024);
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + DEPTH - 1 < KS) r[(s + DEPTH - 1) % DEPTH] = e4_frag(pa + (s + DEPTH - 1) * 1024);
    acc = e4_mfma(r[s % DEPTH], Bf[s], acc);
    __builtin_amdgcn_sched_barrier(0);  // pin: one ds_read, one MFMA per k-step (hipcc otherwise sinks every read to its use)
  }
}

// selection fragment of the fold k-step (B operand): k = 8 half + e; half 0 picks row k = p >> 2 of the patch, half 1 picks
// column j = p & 3 — of the patch's first sample (e < 4) or, for rows >= ns of a patch that straddles two samples, of the
// next one (e >= 4).  Rebuilt where it is used (a few VALU instructions) instead of living through layer 2.
__device__ __forceinline__ hx8 e4_sel(int lane, int ns) {
  asm volatile("" : "+v"(lane));
  const int p = lane & 31, want = (lane >> 5) ? (p & 3) + ((p >> 2) >= ns ? 4 : 0) : (p >> 2);
  const unsigned one = (want & 1) ? (FD_H_ONE_BITS << 16) : FD_H_ONE_BITS;  // 1.0 in the odd / even half of a word
  e4_u32x4 w;
#pragma unroll
  for (int k = 0; k < 4; ++k) w[k] = (want >> 1) == k ? one : 0u;
  return __builtin_bit_cast(hx8, w);
}

// ------------------------------------------------------------------ kernel
// A wave's patch: rows 8 rt .. +7 of the flattened [B*N] residue rows (row = b N + i) x columns 4 jt .. +3.  When N % 8 != 0
// a patch can straddle two samples: rows k >= ns belong to sample b0 + 1.  All fields are wave-uniform.
struct E4Tile {
  int rt, jt, b0, ns;
  bool valid;
};
__device__ __forceinline__ E4Tile e4_tile_of(int w, int n_wt, int N, int NJ4) {
  E4Tile t;
  t.valid = w < n_wt;
  if (!t.valid) w = n_wt - 1;
  w = __builtin_amdgcn_readfirstlane(w);
  t.rt = w / NJ4;
  t.jt = w - t.rt * NJ4;
  t.b0 = (8 * t.rt) / N;
  const int left = (t.b0 + 1) * N - 8 * t.rt;
  t.ns = left < 8 ? left : 8;
  return t;
}

// z rows of the wave's patch -> its LDS rows (row p = 4 k + (j - 4 jt), 256 B, unit u of row p at u ^ (p & 15)):
// 8 DMA instructions, one per residue row k (4 pairs = 1 KB contiguous in HBM), swizzle applied on the source side
__device__ __forceinline__ void e4_request_z(const ET2Args& a, const E4Tile& t, int lane, unsigned zst, int M) {
  const int N = a.N;
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    int row = 8 * t.rt + r;
    if (row > M - 1) row = M - 1;
    const int lrow = 4 * r + (lane >> 4);
    const int u = (lane & 15) ^ (lrow & 15);
    const long pair = ((long)row * N + 4 * t.jt) + (lane >> 4);
    e4_dma16_z(a.z_in + pair * E4_CZ + 8 * u, zst + r * 1024);
  }
}

// The LayerNorm epilogue, in slices (statistics, four 32-feature tiles, pair bias)
struct E4Epi {
  f32x16 Y[4];        // final-layer output of the finished tile: feature 32 t + 8 g + 4 half + q in Y[t][4 g + q]
  E4Tile t;
  float em;           // res_mask[i] * res_mask[j]
};
struct E4EpiTmp {     // lives inside one epilogue run only
  f32x16 accb;        // pair bias of the next block, accumulated tile by tile
  f32x16 accd;        // pair_z = down_z(z') of the next block's IPA, transposed: D[pair, d] (PZ kernels)
  f32x2 sa, sc;       // rstd, -mu * rstd (both halves equal)
  float s1, s2;
  long prow;          // this lane's pair (row of z)
  long srow[2];       // the pairs whose 64 B row segments this lane stores (rows (lane >> 2) and 16 + (lane >> 2) of the patch)
  bool valid, svalid[2];
  unsigned moff;      // LDS address of this lane's pair mask
  unsigned boff;      // LDS address of the linear_b bias
};
typedef __attribute__((address_space(3))) e4_u32x2* e4_lds_w64;
// x(lane) + x(lane ^ 32).  ds_bpermute with the partner address computed on the spot from the caller's (opaque) lane index:
// __shfl_xor computes its own lane id, which hipcc hoists out of the tile loop and spills — and the reload is a vmcnt wait
// that drains the whole DMA queue of the next tile.  (v_permlane32_swap would avoid LDS, but hipcc mis-handles its second
// result when the operands are equal or constant: tools/micro/permlane_test.hip.)
__device__ __forceinline__ float e4_both_halves(float x, int lane) {
  return x + __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((lane ^ 32) << 2, __builtin_bit_cast(int, x)));
}
// PZ (round 6): the epilogue also emits pair_z = down_z(z') + b of the NEXT block's IPA (ipa_pytorch.py:158,318: o_pair = sum_j a_ij pair_z_ij)
// as half precision [row][j / 4][32 d][4 j] (ET2Args.pz_out), so that opair_pz_kernel reads 64 B per pair instead of the 256 B of z'.
// z' is the A operand here (D[pair, d]: a lane ends up with four consecutive j of one row i for its d = 8 B of the image, no transposition);
// weights hi + lo (the rounding of W_dz is shared by all keys of a row: tests/err_budget.py `opair.w`); both images arrive as the last
// chunk of the weight stream (ring slot 3: `dzf`), which nobody overwrites before the next tile's first stream point
// STZ = false (round 6): z' itself is not stored — the launch behind the LAST trunk block that has an EdgeTransition: the next block's
// attention takes its pair bias and its pair_z from this epilogue, and nothing else reads z' any more (184 MB of writes, the staging and
// sixteen 16 B stores per lane and tile less)
template <int SLOT, bool PZ, bool STZ = true, bool EMR = true>
__device__ __forceinline__ void e4_epi(E4Epi& E, E4EpiTmp& X, const ET2Args& a, int lane, unsigned vec, unsigned wbi, unsigned stg, int M, unsigned dzf) {
  const int p = lane & 31, half = lane >> 5;
  if constexpr (SLOT == 0) {  // sums of y and y^2 (packed fp32 math); the pair mask is requested here
    // the pair mask comes from LDS (parked there at the start of the tile): a global load here would be waited for with
    // vmcnt, i.e. together with the z rows and weights of the next tile requested just before the epilogue
    if constexpr (EMR) E.em = *(const __attribute__((address_space(3))) float*)(unsigned long)(X.moff);
    f32x2 u1 = {0.f, 0.f}, u2 = {0.f, 0.f};
    if (!(E4_ABL & 32))
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const f32x2 y = {E.Y[t][r], E.Y[t][r + 1]};
        u1 += y;
        u2 = __builtin_elementwise_fma(y, y, u2);
      }
    X.s1 = u1[0] + u1[1];
    X.s2 = u2[0] + u2[1];
  } else if constexpr (SLOT == 1) {
    const float s1 = e4_both_halves(X.s1, lane), s2 = e4_both_halves(X.s2, lane);
    const float mu = s1 * (1.0f / E4_CZ);
    const float var = fmaxf(s2 * (1.0f / E4_CZ) - mu * mu, 0.f);
    const float rstd = 1.0f / sqrtf(var + 1e-5f);
    X.sa = f32x2{rstd, rstd};
    X.sc = f32x2{-mu * rstd, -mu * rstd};
    const int row = 8 * E.t.rt + (p >> 2);
    X.valid = E.t.valid && row < M;
    X.prow = (long)(row < M ? row : M - 1) * a.N + 4 * E.t.jt + (p & 3);
#pragma unroll
    for (int k = 0; k < 2; ++k) {  // patch row 16 k + (lane >> 2) = pair (i = that >> 2, j = that & 3)
      const int pr = 16 * k + (lane >> 2), r2 = 8 * E.t.rt + (pr >> 2);
      X.svalid[k] = E.t.valid && r2 < M;
      X.srow[k] = (long)(r2 < M ? r2 : M - 1) * a.N + 4 * E.t.jt + (pr & 3);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) X.accb[r] = 0.f;
    if constexpr (PZ) {  // D[pair, d] starts as the bias of down_z (this lane's d = p)
      const float bd = *(const __attribute__((address_space(3))) float*)(unsigned long)(vec + 4 * (E4_H + 2 * E4_CZ + p));
#pragma unroll
      for (int r = 0; r < 16; ++r) X.accd[r] = bd;
    }
  } else if constexpr (SLOT >= 2 && SLOT < 6) {  // one 32-feature tile: normalise, mask, bf16, store, its share of the pair bias
    constexpr int t = SLOT - 2;
    const unsigned gml = vec + 4 * (E4_H + 4 * half + 32 * t);
    const unsigned btl = gml + 4 * E4_CZ;
    e4_u32x4 zB[2];
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {  // two halves of the tile: their gamma / beta first, then the math (no control flow between)
      f32x4 gm[2], bt[2];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        gm[k] = e4_ldsf4(gml + 32 * (2 * h2 + k));
        bt[k] = e4_ldsf4(btl + 32 * (2 * h2 + k));
      }
      f32x2 o[4];
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int g = 2 * h2 + k;
        f32x2 o0 = {E.Y[t][4 * g], E.Y[t][4 * g + 1]}, o1 = {E.Y[t][4 * g + 2], E.Y[t][4 * g + 3]};
        if (E4_ABL & 32) { o[2 * k] = o0; o[2 * k + 1] = o1; continue; }
        o0 = __builtin_elementwise_fma(o0, X.sa, X.sc);
        o1 = __builtin_elementwise_fma(o1, X.sa, X.sc);
        o[2 * k] = __builtin_elementwise_fma(o0, f32x2{gm[k][0], gm[k][1]}, f32x2{bt[k][0], bt[k][1]});
        o[2 * k + 1] = __builtin_elementwise_fma(o1, f32x2{gm[k][2], gm[k][3]}, f32x2{bt[k][2], bt[k][3]});
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) o[k] *= f32x2{E.em, E.em};  // (a wave-uniform "all ones" test gets if-converted into selects: dearer)
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int g = 2 * h2 + k;
        const e4_u32x2 ow = {fd_cvt_pk(o[2 * k][0], o[2 * k][1]), fd_cvt_pk(o[2 * k + 1][0], o[2 * k + 1][1])};
        // registers 4 g .. 4 g + 3 of tile t -> B fragment 2 t + (g >> 1) of z'
        zB[h2][2 * k] = ow[0];
        zB[h2][2 * k + 1] = ow[1];
        // staging row p (64 B = this tile's 32 features), 16 B chunk g at g ^ ((p >> 2) & 3), 8 B half
        if constexpr (STZ) *(e4_lds_w64)(unsigned long)(stg + p * 64 + ((g ^ ((p >> 2) & 3)) << 4) + 8 * half) = ow;
      }
      if (a.trace && X.valid) {
        float* tr_row = a.trace + X.prow * E4_CZ + 4 * half + 32 * t + 16 * h2;
#pragma unroll
        for (int k = 0; k < 2; ++k) *(f32x4*)(tr_row + 8 * k) = f32x4{o[2 * k][0], o[2 * k][1], o[2 * k + 1][0], o[2 * k + 1][1]};
      }
    }
    if (a.wb_img) {  // D[head, pair] += Wb[:, this tile's features] z'  (compact image: lanes >= 8 read the zero unit)
      const unsigned wl = p < 8 ? wbi + half * 128 + p * 16 : wbi + 2048, ws = p < 8 ? 256u : 0u;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2)
        X.accb = fd_mfma32(e4_frag(wl + (2 * t + h2) * ws), __builtin_bit_cast(hx8, zB[h2]), X.accb);
    }
    if constexpr (PZ) {  // D[pair, d] += z' Wdz[d, this tile's features]  (hi from LDS, lo from registers)
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        if (!(E4_PZ_ABL & 8)) X.accd = fd_mfma32(__builtin_bit_cast(hx8, zB[h2]), e4_frag(dzf + (2 * t + h2) * 1024 + lane * 16), X.accd);
        if (!(E4_PZ_ABL & 1)) X.accd = fd_mfma32(__builtin_bit_cast(hx8, zB[h2]), e4_frag(dzf + 8192 + (2 * t + h2) * 1024 + lane * 16), X.accd);
      }
    }
    // read the staged tile back as 64 B row segments (the LDS operations of one wave execute in order: no barrier) and store
    if constexpr (STZ)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pr = 16 * k + (lane >> 2);
      const u16x8 v = *(e4_lds_u16x8)(unsigned long)(stg + pr * 64 + (((lane & 3) ^ ((pr >> 2) & 3)) << 4));
      if (X.svalid[k] && (!(E4_ABL & 8) || X.prow == -12345)) e4_store_z(a.z_out + X.srow[k] * E4_CZ + 32 * t + 8 * (lane & 3), v);
    }
  } else if constexpr (SLOT == 6) {
    if (a.wb_img) {
      // pair bias of the next block's attention: head 4 half + r in register r < 4
      const unsigned bbo = X.boff;
      const int row = 8 * E.t.rt + (p >> 2), jj = 4 * E.t.jt + (p & 3);
      if (E.t.valid && row < M && (!(E4_ABL & 8) || row == -12345)) {
        const int b = row / a.N, ii = row - b * a.N, nt = (a.N + 31) >> 5;
        float* bo = a.bias_out + fd_bias_frag_off((long)b * a.H + 4 * half, nt, ii, jj);
        const long hstride = (long)nt * nt * 1024;
        const f32x4 bbv = e4_ldsf4(bbo + 16 * half);  // (from LDS: a global load here would be a vmcnt(0) drain per head)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (4 * half + r < a.H) bo[r * hstride] = X.accb[r] + bbv[r];
      }
    }
    if constexpr (PZ) {
      // registers 4 g .. 4 g + 3 = pairs 8 g + 4 half + q = patch row 2 g + half, columns 4 jt .. + 3: 8 B of the image at [row][jt][d = p]
      const int NJ4 = a.N >> 2;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 8 * E.t.rt + 2 * g + half;
        const e4_u32x2 ow = {fd_cvt_pk(X.accd[4 * g], X.accd[4 * g + 1]), fd_cvt_pk(X.accd[4 * g + 2], X.accd[4 * g + 3])};
        if (E.t.valid && row < M && (!(E4_ABL & 8) || row == -12345) && (!(E4_PZ_ABL & 4) || row == -12345))
          *(e4_u32x2*)(a.pz_out + (((long)row * NJ4 + E.t.jt) * 32 + p) * 4) = ow;
      }
    }
  }
}

// E4_PIPE: the same epilogue in two parts with the same arithmetic (bit-identical outputs).  NORM (at the tile's end: slots 0, 1 above, then
// this for t = 0 .. 3) turns the final layer's 64 fp32 registers into the eight half-precision B fragments of z' (32 registers); EMIT (under
// the next tile's layer 1) stages a tile, runs its products of the pair bias / pair_z emissions and stores its z' rows.
template <int t>
__device__ __forceinline__ void e4_epi_norm(const E4Epi& E, const E4EpiTmp& X, const ET2Args& a, int lane, unsigned vec, e4_u32x4 (&Z)[8]) {
  const int half = lane >> 5;
  const unsigned gml = vec + 4 * (E4_H + 4 * half + 32 * t);
  const unsigned btl = gml + 4 * E4_CZ;
#pragma unroll
  for (int h2 = 0; h2 < 2; ++h2) {
    f32x4 gm[2], bt[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      gm[k] = e4_ldsf4(gml + 32 * (2 * h2 + k));
      bt[k] = e4_ldsf4(btl + 32 * (2 * h2 + k));
    }
    f32x2 o[4];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int g = 2 * h2 + k;
      f32x2 o0 = {E.Y[t][4 * g], E.Y[t][4 * g + 1]}, o1 = {E.Y[t][4 * g + 2], E.Y[t][4 * g + 3]};
      o0 = __builtin_elementwise_fma(o0, X.sa, X.sc);
      o1 = __builtin_elementwise_fma(o1, X.sa, X.sc);
      o[2 * k] = __builtin_elementwise_fma(o0, f32x2{gm[k][0], gm[k][1]}, f32x2{bt[k][0], bt[k][1]});
      o[2 * k + 1] = __builtin_elementwise_fma(o1, f32x2{gm[k][2], gm[k][3]}, f32x2{bt[k][2], bt[k][3]});
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) o[k] *= f32x2{E.em, E.em};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      Z[2 * t + h2][2 * k] = fd_cvt_pk(o[2 * k][0], o[2 * k][1]);
      Z[2 * t + h2][2 * k + 1] = fd_cvt_pk(o[2 * k + 1][0], o[2 * k + 1][1]);
    }
    if (a.trace && X.valid) {
      float* tr_row = a.trace + X.prow * E4_CZ + 4 * half + 32 * t + 16 * h2;
#pragma unroll
      for (int k = 0; k < 2; ++k) *(f32x4*)(tr_row + 8 * k) = f32x4{o[2 * k][0], o[2 * k][1], o[2 * k + 1][0], o[2 * k + 1][1]};
    }
  }
  __builtin_amdgcn_sched_barrier(0);
}
template <int t, bool PZ, bool STZ>
__device__ __forceinline__ void e4_epi_emit(E4EpiTmp& X, const ET2Args& a, int lane, unsigned wbi, unsigned stg, unsigned dzf, const e4_u32x4 (&Z)[8]) {
  const int p = lane & 31, half = lane >> 5;
  if constexpr (STZ)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const e4_u32x2 ow = {Z[2 * t + (g >> 1)][2 * (g & 1)], Z[2 * t + (g >> 1)][2 * (g & 1) + 1]};
      *(e4_lds_w64)(unsigned long)(stg + p * 64 + ((g ^ ((p >> 2) & 3)) << 4) + 8 * half) = ow;
    }
  if (a.wb_img) {
    const unsigned wl = p < 8 ? wbi + half * 128 + p * 16 : wbi + 2048, ws = p < 8 ? 256u : 0u;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) X.accb = fd_mfma32(e4_frag(wl + (2 * t + h2) * ws), __builtin_bit_cast(hx8, Z[2 * t + h2]), X.accb);
  }
  if constexpr (PZ) {
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      X.accd = fd_mfma32(__builtin_bit_cast(hx8, Z[2 * t + h2]), e4_frag(dzf + (2 * t + h2) * 1024 + lane * 16), X.accd);
      X.accd = fd_mfma32(__builtin_bit_cast(hx8, Z[2 * t + h2]), e4_frag(dzf + 8192 + (2 * t + h2) * 1024 + lane * 16), X.accd);
    }
  }
  if constexpr (STZ)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int pr = 16 * k + (lane >> 2);
      const u16x8 v = *(e4_lds_u16x8)(unsigned long)(stg + pr * 64 + (((lane & 3) ^ ((pr >> 2) & 3)) << 4));
      if (X.svalid[k]) e4_store_z(a.z_out + X.srow[k] * E4_CZ + 32 * t + 8 * (lane & 3), v);
    }
  __builtin_amdgcn_sched_barrier(0);
}

// ------------------------------------------------------------------ flat-stream kernel (round 2, end)
// Same arithmetic, same MFMA order per accumulator (bit-identical results) as edge_transition4_kernel above; what changes is how
// the 512 weight fragments of a tile reach the matrix cores.  Above: 20 chunks through a 2 x 32 KB ring, a DMA wait + barrier at
// the END of every chunk and a fresh LDS prefetch prologue for each of the 28 (tile | chunk) units — 28 exposed LDS round trips and
// 20 drained pipelines per 256 pairs.  Here the fragments are one cyclic stream f = 0 .. 511 (the next tile's continue it):
//   * ring of FOUR 16 KB chunks (same 64 KB); chunk c + 3 is requested, and chunk c + 1 declared valid, at a barrier in the
//     MIDDLE of chunk c (after fragment 16 c + 8): the wait there is for a DMA issued two chunks ago and nobody stands at a
//     chunk boundary — the barrier orders (all waves are past chunk c - 1, whose slot the new DMA overwrites) without draining;
//   * one operand ring of E4_DR fragments per wave runs through the whole tile, across feature tiles, chunks and layers;
//   * no barrier and no vmcnt(0) at the tile end: z rows, fold fragments and pair masks are wave-private and are waited for by
//     count (in-order vmcnt; every count below is a LOWER bound of the younger instructions, so it can only over-wait).
#ifndef E4_DR
#define E4_DR 4
#endif
// E4_FINE (round 6, experiment): the ring as SIX 8 KB slots over the 60 chunks that hold weights (the same 2 k cycles between a request and
// its use, twice the stream points), the down_z fragments (chunk 31 of the image) loaded ONCE per block into the 16 KB behind them, where
// ring slot 3 was: they then survive the tile boundary (what an epilogue that runs under the next tile's layer 1 needs), and the unused
// chunk and the per-tile down_z chunk leave the stream.
#ifndef E4_FINE
#define E4_FINE 0
#endif
#if E4_FINE
#define E4_CHUNK 8192
#define E4_NSLOT 6
#define E4_NCHUNK 60
#else
#define E4_CHUNK 16384
#define E4_NSLOT 4
#define E4_NCHUNK 32
#endif
// E4_PIPE (needs E4_FINE): the LayerNorm epilogue of a tile runs in slices behind the first feature tiles of the NEXT tile's layer 1 (the
// last tile of a block: at its end, as before); E4_PIPE_T0 = the feature tile behind which the first slice runs
#ifndef E4_PIPE
#define E4_PIPE 0
#endif
#ifndef E4_PIPE_T0
#define E4_PIPE_T0 0
#endif
static_assert(!E4_PIPE || E4_FINE, "the pipelined epilogue needs the down_z fragments resident (E4_FINE)");
#define E4_CFR (E4_CHUNK / 1024)               // fragments per chunk
#define E4_DZ_LDS 49152u                       // down_z hi | lo in LDS: ring slot 3 of the four-slot ring = the 16 KB behind the six 8 KB slots
#define E4_DPC (E4_CHUNK / (E4_THREADS * 16))  // DMA instructions per chunk and wave
__device__ __forceinline__ void e4_vm_wait(int n) {  // s_waitcnt vmcnt(n) (expcnt / lgkmcnt untouched); n folds after unrolling
  __builtin_amdgcn_sched_barrier(0);
#define E4_VMC(k) case k: __builtin_amdgcn_s_waitcnt(0x0F70 | ((k) & 15) | (((k) >> 4) << 14)); break;
  switch (n) {
    E4_VMC(1) E4_VMC(2) E4_VMC(3) E4_VMC(4) E4_VMC(5) E4_VMC(6) E4_VMC(7) E4_VMC(8) E4_VMC(9) E4_VMC(10) E4_VMC(11) E4_VMC(12)
    E4_VMC(13) E4_VMC(14) E4_VMC(15) E4_VMC(16) E4_VMC(17) E4_VMC(18) E4_VMC(19) E4_VMC(20)
    default: __builtin_amdgcn_s_waitcnt(0x0F70); break;
  }
#undef E4_VMC
  __builtin_amdgcn_sched_barrier(0);
}
// younger VM instructions (guaranteed ones) than the DMA of chunk c + 1 at the barrier point of chunk c: always the DMA of chunk
// c + 2; plus the 8 z requests issued at point E4_PT_Z, the 4 final-layer fold loads issued at point E4_PT_FL, the 12 + 2 fold / mask
// loads of the tile boundary
// (general form: the awaited DMA was issued at point c - (E4_NSLOT - 2); younger are the DMAs of the E4_NSLOT - 3 points behind it and
//  whatever was issued behind the DMA of a point P with c - (E4_NSLOT - 2) <= P < c; the tile boundary counts as P = -1)
#if E4_FINE
#define E4_PT_Z 48
#define E4_PT_FL 52
#else
#define E4_PT_Z 24   // stream point behind which the next tile's z rows are requested
#define E4_PT_FL 26  // ... the final layer's fold fragments
#endif
__device__ __forceinline__ constexpr int e4_vm_younger(int c) {
  return (E4_NSLOT - 3) * E4_DPC + (c > E4_PT_Z && c <= E4_PT_Z + E4_NSLOT - 2 ? 8 : (c > E4_PT_FL && c <= E4_PT_FL + E4_NSLOT - 2 ? 4 : (c <= E4_NSLOT - 3 ? 14 : 0)));
}
__device__ __forceinline__ constexpr unsigned e4_ring_off(int f) { return (unsigned)(((f / E4_CFR) % E4_NSLOT) * E4_CHUNK + (f % E4_CFR) * 1024); }
struct E4Flat {
  const char* stream;
  unsigned lds0, pa;
  int tid, wave;
};
__device__ __forceinline__ void e4_point(const E4Flat& F, int c) {
  e4_vm_wait(e4_vm_younger(c));
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  const int cn = (c + E4_NSLOT - 1) % E4_NCHUNK;
  // (E4_ABL & 64, timing only: layer 1's six chunks are never streamed - the upper bound of what keeping them resident in LDS could buy)
  if (!(E4_ABL & 64) || cn * E4_CFR >= 96)
    e4_dma_chunk<E4_CHUNK>(F.stream + (size_t)cn * E4_CHUNK, F.lds0 + (unsigned)(cn % E4_NSLOT) * E4_CHUNK, F.tid, F.wave);
}
// E4_PIPE: the stores of the slices that ran behind feature tile P (issued between stream points P and P + 1) are younger than the DMA a
// point c awaits for P < c <= P + E4_NSLOT - 2: counted when every lane of the wave stored (full tile, all eight heads), otherwise the
// plain lower bound over-waits.  Slices: EMIT(t) behind tile E4_PIPE_T0 + t (two z' stores), the pair bias / pair_z stores behind + 4.
template <bool PZ, bool STZ>
__device__ __forceinline__ constexpr int e4_pipe_extra(int c) {
  int n = 0;
  for (int t = 0; t < 4; ++t)
    if (E4_PIPE_T0 + t < c && c <= E4_PIPE_T0 + t + E4_NSLOT - 2) n += STZ ? 2 : 0;
  if (E4_PIPE_T0 + 4 < c && c <= E4_PIPE_T0 + 4 + E4_NSLOT - 2) n += PZ ? 8 : 0;
  return n;
}
__device__ __forceinline__ void e4_point_x(const E4Flat& F, int c, int extra, bool counted) {  // (c and extra fold after unrolling)
  if (extra > 0 && counted) e4_vm_wait(e4_vm_younger(c) + extra);
  else e4_vm_wait(e4_vm_younger(c));
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);
  const int cn = (c + E4_NSLOT - 1) % E4_NCHUNK;
  e4_dma_chunk<E4_CHUNK>(F.stream + (size_t)cn * E4_CHUNK, F.lds0 + (unsigned)(cn % E4_NSLOT) * E4_CHUNK, F.tid, F.wave);
}
// one k-step of the stream: the operand ring is refilled E4_DR - 1 fragments ahead (not past the tile's last fragment)
#define E4_STEP(f_, B_, acc_)                                                                                   \
  do {                                                                                                          \
    if ((f_) + E4_DR - 1 < E4_NCHUNK * E4_CFR && (!(E4_ABL & 16) || (f_) < 8)) r[((f_) + E4_DR - 1) % E4_DR] = e4_frag(F.pa + e4_ring_off((f_) + E4_DR - 1)); \
    acc_ = e4_mfma(r[(f_) % E4_DR], B_, acc_);                                                                  \
    __builtin_amdgcn_sched_barrier(0);                                                                          \
  } while (0)

template <bool PZ, bool STZ = true>
__global__ __launch_bounds__(E4_THREADS, 8 / E4_WAVES) void edge_transition4_flat_kernel(ET2Args a, int n_tiles, int n_wt) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)smem;
  const unsigned vec = lds0 + E4_VOFF;
  const unsigned wbi = lds0 + E4_VOFF + E4_VEC_BYTES;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef E4_PRIO  // experiment: 1 = the first wave of every SIMD at a higher issue priority (runs ahead inside the ring's slack), 2 = the second one
  if (E4_PRIO == 1 ? wave < 4 : wave >= 4) __builtin_amdgcn_s_setprio(2);
#endif
  auto lane_id = [] {
    int l = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    asm volatile("" : "+v"(l));
    return l;
  };
  const int lane0 = lane_id(), tid0 = wave * 64 + lane0;
  const int N = a.N, NJ4 = N >> 2, M = a.B * N;
  const char* stream = (const char*)a.stream;
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  E4_CLK_BEGIN;
#ifdef E4_PROF
  unsigned ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = (unsigned)__builtin_amdgcn_s_memtime();
  const unsigned long long span0 = __builtin_amdgcn_s_memrealtime();
#endif
  E4Tile tc = e4_tile_of(tile * E4_WAVES + wave, n_wt, N, NJ4);
  e4_request_z(a, tc, lane0, lds0 + E4_ZOFF + wave * 8192, M);
#pragma unroll
  for (int c = 0; c < E4_NSLOT - 1; ++c) e4_dma_chunk<E4_CHUNK>(stream + c * E4_CHUNK, lds0 + c * E4_CHUNK, tid0, wave);
#if E4_FINE
  if (PZ) e4_dma_chunk<16384>(stream + (size_t)E4_DZ_FR0 * 1024, lds0 + E4_DZ_LDS, tid0, wave);  // down_z hi | lo: resident for the block
#endif
  if (tid0 < (PZ ? 168 : 160)) {
    const float* src = tid0 < 96 ? a.b2 + 4 * tid0 : (tid0 < 128 ? a.gamma + 4 * (tid0 - 96) : (tid0 < 160 ? a.beta + 4 * (tid0 - 128) : a.bdz + 4 * (tid0 - 160)));
    e4_dma16(src, vec + (tid0 & ~63) * 16);
  }
  if (a.wb_img && tid0 < 128) e4_dma16((const char*)a.wb_img + tid0 * 16, wbi + (tid0 & ~63) * 16);  // compact image (2 KB)
  if (tid0 < 4) *(__attribute__((address_space(3))) unsigned*)(unsigned long)(wbi + 2048 + tid0 * 4) = 0u;  // the zero unit
  auto fold_ptr = [&](const E4Tile& t, int lane) {
    const unsigned fold_b1 = (unsigned)((const char*)a.b1_img - (const char*)a.a1_img);
    const unsigned ob = fold_b1 + (unsigned)((t.b0 * NJ4 + t.jt) * 16) * 512u, oa = (unsigned)(t.rt * 16) * 512u;  // scalar
    return oa + (unsigned)(lane >> 5) * (ob - oa) + (lane & 31) * 16;
  };
  auto fold_ld = [&](unsigned off) { return e4_gfrag((const char*)a.a1_img + off); };
  unsigned fold_base = fold_ptr(tc, lane0);
  hx8 FA[12];
#pragma unroll
  for (int k = 0; k < 12; ++k) FA[k] = fold_ld(fold_base + k * 512);
  auto mask_of = [&](const E4Tile& t, int lane) {
    int row = 8 * t.rt + ((lane & 31) >> 2);
    if (row > M - 1) row = M - 1;
    return a.res_mask[row] * a.res_mask[(row / N) * N + 4 * t.jt + (lane & 3)];
  };
  if (tid0 < 8) *(__attribute__((address_space(3))) float*)(unsigned long)(lds0 + E4_BOFF + tid0 * 4) = (a.wb_img && tid0 < a.H) ? a.bb[tid0] : 0.f;
  float em_req = mask_of(tc, lane0);
  E4Epi E;
  E4EpiTmp X;           // (E4_PIPE: lives across the tile boundary)
  bool pend = false;    // E4_PIPE: the previous tile's epilogue is still to run (wave-uniform)
  bool pend_full = false;  // ... and every lane of the wave will issue every store of it (the counts of e4_pipe_extra hold)
#define E4_EPI(k, emr) e4_epi<k, PZ, STZ, emr>(E, X, a, lane_id(), vec, wbi, lds0 + E4_SOFF + wave * 2048, M, lds0 + E4_DZ_LDS)
#if E4_PIPE
  e4_u32x4 Zq[8];       // z' of the finished tile as half-precision B fragments (between NORM and EMIT)
#define E4_NORM(t) e4_epi_norm<t>(E, X, a, lane_id(), vec, Zq)
#define E4_EMIT(t)                                                                                                          \
  do {                                                                                                                      \
    if (!(E4_ABL & 128)) e4_epi_emit<t, PZ, STZ>(X, a, lane_id(), wbi, lds0 + E4_SOFF + wave * 2048, lds0 + E4_DZ_LDS, Zq);  \
    else if (Zq[2 * t][0] == 0x12345678u) a.z_out[t] = 1;  /* (E4_ABL & 128, timing only: the emitting part costs nothing) */ \
  } while (0)
#endif
  e4_dma_wait();
  *(__attribute__((address_space(3))) float*)(unsigned long)(lds0 + E4_MOFF + tid0 * 4) = em_req;
  __syncthreads();
  E4_STAMP(0);
#pragma unroll 1
  for (;;) {
    const int lane = lane_id(), tid = wave * 64 + lane;
    const int p = lane & 31, half = lane >> 5;
    const unsigned zst = lds0 + E4_ZOFF + wave * 8192;
    const unsigned zrow = zst + p * 256;
    E4Flat F;
    F.stream = stream; F.lds0 = lds0; F.pa = lds0 + lane * 16; F.tid = tid; F.wave = wave;
    hx8 r[E4_DR];
#pragma unroll
    for (int m = 0; m < E4_DR - 1; ++m) r[m] = e4_frag(F.pa + e4_ring_off(m));
    hx8 H1[24], H2[24];
    // ================= layer 1: fragments 0 .. 95 (12 tiles x 8 k-steps of z, + the fold step)
    {
      const hx8 SEL = e4_sel(lane, tc.ns);
      hx8 Zf[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) Zf[s] = e4_frag(zrow + (((2 * s) ^ (half ^ (p & 15))) << 4));
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int T = 0; T < 12; ++T) {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        acc = e4_mfma(FA[T], SEL, acc);
#pragma unroll
        for (int s = 0; s < 8; ++s) {
          const int f = 8 * T + s;
#if E4_PIPE
          if (f % E4_CFR == E4_CFR / 2) e4_point_x(F, f / E4_CFR, e4_pipe_extra<PZ, STZ>(f / E4_CFR), pend_full);
#else
          if (f % E4_CFR == E4_CFR / 2) e4_point(F, f / E4_CFR);
#endif
          E4_STEP(f, Zf[s], acc);
        }
        e4_hand_off(acc, H1[2 * T], H1[2 * T + 1]);
        if constexpr (E4_PIPE != 0) {
#if E4_PIPE
          if (pend) {
#ifdef E4_PIPE_PRIO  // a slice's vector / LDS / store instructions ahead of the SIMD's other wave's MFMAs (they otherwise get one turn per MFMA)
            __builtin_amdgcn_s_setprio(E4_PIPE_PRIO);
#endif
            if (T == E4_PIPE_T0) E4_EMIT(0);
            if (T == E4_PIPE_T0 + 1) E4_EMIT(1);
            if (T == E4_PIPE_T0 + 2) E4_EMIT(2);
            if (T == E4_PIPE_T0 + 3) E4_EMIT(3);
            if (T == E4_PIPE_T0 + 4) E4_EPI(6, false);
            __builtin_amdgcn_sched_barrier(0);
#ifdef E4_PIPE_PRIO
            __builtin_amdgcn_s_setprio(0);
#endif
          }
#endif
        }
      }
    }
    pend = false;
    pend_full = false;
    E4_STAMP(1);
    // ================= layer 2: fragments 96 .. 383 (12 tiles x 24 k-steps); the accumulator starts as b2.  The first four tiles are the
    // hidden features that face z in the residual trunk(x) + x (ipa_pytorch.py:99): z joins them as they are handed over (e4_add_z)
#pragma unroll
    for (int T = 0; T < 12; ++T) {
      f32x16 acc;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const f32x4 bv = e4_ldsf4(vec + 4 * (32 * T + 8 * g + 4 * half));
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[4 * g + q] = bv[q];
      }
#pragma unroll
      for (int s = 0; s < 24; ++s) {
        const int f = E4_L1_FR + 24 * T + s;
        if (f % E4_CFR == E4_CFR / 2) e4_point(F, f / E4_CFR);
        E4_STEP(f, H1[s], acc);
      }
      e4_hand_off(acc, H2[2 * T], H2[2 * T + 1]);
      if (T < 4) {
        // z in hand-off order: element (half, e) of fragment 2 T + u = feature 32 T + 16 u + 8 (e >> 2) + 4 half + (e & 3): two 8 B pieces of
        // the lane's z row (16 B unit n of row p at n ^ (p & 15)); packed half-precision adds
        const int l3 = lane_id();  // (from an opaque copy: the layer-1 row addresses must not stay live)
        const unsigned zrow3 = lds0 + E4_ZOFF + wave * 8192 + (l3 & 31) * 256 + 8 * (l3 >> 5), zsw = l3 & 15;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          e4_u32x4 zw;
          const e4_u32x2 z0 = *(e4_lds_w64)(unsigned long)(zrow3 + (((4 * T + 2 * u) ^ zsw) << 4));
          const e4_u32x2 z1 = *(e4_lds_w64)(unsigned long)(zrow3 + (((4 * T + 2 * u + 1) ^ zsw) << 4));
          zw[0] = z0[0]; zw[1] = z0[1]; zw[2] = z1[0]; zw[3] = z1[1];
          H2[2 * T + u] = H2[2 * T + u] + __builtin_bit_cast(hx8, zw);
        }
      }
    }
    E4_STAMP(2);
    // ================= final layer: fragments 384 .. 479, k-major over the 4 output tiles: 24 k-steps of h2 + x (the z part of x was added
    // at the hand-over, its e_i / e_j parts arrive through the fold step below); the stream's last two chunks hold no layer weights
    const int ntile = tile + gridDim.x;
    const bool has_next = ntile < n_tiles;
    const E4Tile tn = e4_tile_of((has_next ? ntile : tile) * E4_WAVES + wave, n_wt, N, NJ4);
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int q = 0; q < 16; ++q) E.Y[t][q] = 0.f;
    hx8 FL[4];
#pragma unroll
    for (int s = 0; s < 24; ++s)
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        const int f = E4_L1_FR + E4_L2_FR + 4 * s + t;
        if (f % E4_CFR == E4_CFR / 2) {
          e4_point(F, f / E4_CFR);
          // point E4_PT_Z: the wave's z rows are free since the fourth hand-over of layer 2 — the next tile's are requested here, seven chunks
          // before the tile ends (always issued, the last tile re-requests its own: the counts of e4_vm_younger stay static)
          if (f / E4_CFR == E4_PT_Z) e4_request_z(a, tn, lane_id(), lds0 + E4_ZOFF + wave * 8192, M);
          if (f / E4_CFR == E4_PT_FL) {
            const unsigned fb = fold_ptr(tc, lane_id());
#pragma unroll
            for (int k = 0; k < 4; ++k) FL[k] = fold_ld(fb + (12 + k) * 512);
          }
        }
        E4_STEP(f, H2[s], E.Y[t]);
      }
#if !E4_FINE
    e4_point(F, 30);  // the unused chunk and the down_z chunk: their stream points without products (the ring keeps turning)
    e4_point(F, 31);
#endif
    {
      const hx8 SEL = e4_sel(lane, tc.ns);
#pragma unroll
      for (int t = 0; t < 4; ++t) E.Y[t] = e4_mfma(FL[t], SEL, E.Y[t]);
    }
    E4_STAMP(3);
    // ================= tile boundary: the next tile's fold fragments and pair mask are requested, then the LayerNorm epilogue runs
    fold_base = fold_ptr(tn, lane);
#pragma unroll
    for (int k = 0; k < 12; ++k) FA[k] = fold_ld(fold_base + k * 512);
    E.t = tc;
    em_req = mask_of(tn, lane);
#ifdef E4_IDLE  // experiment: E4_IDLE x 64 idle cycles per tile and wave - how much of an idle cycle the power-capped clock gives back
    __builtin_amdgcn_s_sleep(E4_IDLE);
#endif
    X.moff = lds0 + E4_MOFF + tid * 4;
    X.boff = lds0 + E4_BOFF;
#if E4_PIPE
    {  // statistics and the normalised half-precision z' now (32 registers instead of the final layer's 64); with a next tile its products
       // and stores run under that tile's layer 1 (E.t, X and Zq stay untouched until then), without one right here
      E4_EPI(0, true); E4_EPI(1, true);
      E4_NORM(0); E4_NORM(1); E4_NORM(2); E4_NORM(3);
      if (has_next) {
        pend = true;
        pend_full = PZ && E.t.valid && 8 * E.t.rt + 7 < M && a.wb_img && a.H == 8;
      } else { E4_EMIT(0); E4_EMIT(1); E4_EMIT(2); E4_EMIT(3); E4_EPI(6, false); }
    }
    if (false) {
#else
    if (!(E4_ABL & 1)) {
#endif
      E4_EPI(0, true); E4_EPI(1, true); E4_EPI(2, true); E4_EPI(3, true); E4_EPI(4, true); E4_EPI(5, true); E4_EPI(6, true);
    } else if (E.Y[0][0] == 1234.5f) a.z_out[tile] = 1;
    E4_STAMP(4);
    if (!has_next) break;
    tile = ntile;
    tc = tn;
    // the z rows (requested at point 26) are older than the 12 + 2 boundary loads and the epilogue's stores: in-order vmcnt
    e4_vm_wait(14);
    *(__attribute__((address_space(3))) float*)(unsigned long)(lds0 + E4_MOFF + tid * 4) = em_req;  // (this lane's own slot)
    E4_STAMP(5);
  }
  e4_dma_wait();  // the last tile's stream DMAs (chunks 0..2 again, never read) must not outlive the block's LDS
  E4_CLK_END;
#ifdef E4_PROF
  if (tid0 == 0 && blockIdx.x < 256)
    for (int k = 0; k < 8; ++k) e4_prof[blockIdx.x * 8 + k] = ph[k];
  if (tid0 == 0 && blockIdx.x < 1024) {
    e4_span[blockIdx.x * 3] = span0;
    e4_span[blockIdx.x * 3 + 1] = __builtin_amdgcn_s_memrealtime();
    e4_span[blockIdx.x * 3 + 2] = 0;
  }
#endif
}

int fd_edge_transition4_supported(int N) { return N >= 8 && N <= 2048 && N % 4 == 0; }

#ifndef E4_FLAT
#define E4_FLAT 1  // 1: edge_transition4_flat_kernel, 0: the chunk-synchronous kernel
#endif
int fd_edge_transition4_variant(const ET2Args& a, hipStream_t st, int flat);
int fd_edge_transition4(const ET2Args& a, hipStream_t st) {
#ifdef FDIPT_DEV  // development build: FDIPT_ET4_FLAT=0/1 selects the kernel (same-box A/B runs)
  static const int flat = [] { const char* e = getenv("FDIPT_ET4_FLAT"); return e ? atoi(e) : E4_FLAT; }();
  return fd_edge_transition4_variant(a, st, flat);
#else
  return fd_edge_transition4_variant(a, st, E4_FLAT);
#endif
}
int fd_edge_transition4_variant(const ET2Args& a, hipStream_t st, int flat) {
  const long n_pairs = (long)a.B * a.N * a.N;
  if (n_pairs >= (1L << 31) - 256 || !a.a1_img || !a.b1_img || a.N % 4) return FDIPT_EINVAL;  // 32-bit pair indices in the kernel
  {  // the kernel addresses both fold images with 32-bit offsets from a1_img
    const long d = (const char*)a.b1_img - (const char*)a.a1_img;
    if (d < 0 || d + (long)fd_et4_b_image_bytes(a.B, a.N) >= (1L << 32)) return FDIPT_EINVAL;
  }
  const int n_wt = ((a.B * a.N + 7) / 8) * (a.N / 4);
  const int n_tiles = cdiv(n_wt, E4_WAVES);
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)edge_transition4_flat_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, E4_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)edge_transition4_flat_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, E4_LDS) != hipSuccess ||
        hipFuncSetAttribute((const void*)edge_transition4_flat_kernel<true, false>, hipFuncAttributeMaxDynamicSharedMemorySize, E4_LDS) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  const int n_cu = fd_cu_count();
  // persistent: one block per CU (minus the CUs left to concurrent streams, in whole XCD rounds of 8)
  const int cus = a.reserve_cus > 0 && a.reserve_cus < n_cu - 8 ? (n_cu - a.reserve_cus) & ~7 : n_cu;
  const int slots = cus * (E4_LDS <= 81920 ? 2 : 1);
  const int grid = n_tiles < slots ? n_tiles : slots;
  (void)flat;
  if (a.pz_out) {  // + pair_z of the next block (needs its bias emission: the zero unit / images share its set-up)
    if (!a.wb_img || !a.bdz) return FDIPT_EINVAL;  // (down_z hi / lo: the stream's last chunk, fd_et4_set_dz)
    if (!a.z_out) {  // (only next to both emissions and without a trace: checked by the caller's conditions, and here)
      if (a.trace) return FDIPT_EINVAL;
      hipLaunchKernelGGL((edge_transition4_flat_kernel<true, false>), dim3(grid), dim3(E4_THREADS), E4_LDS, st, a, n_tiles, n_wt);
    } else
    hipLaunchKernelGGL(edge_transition4_flat_kernel<true>, dim3(grid), dim3(E4_THREADS), E4_LDS, st, a, n_tiles, n_wt);
  } else
    hipLaunchKernelGGL(edge_transition4_flat_kernel<false>, dim3(grid), dim3(E4_THREADS), E4_LDS, st, a, n_tiles, n_wt);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

