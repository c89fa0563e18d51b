// edge_embed2.hip — half-precision pair embedder (framedipt/model/score_network.py:98-105,184-197: the edge MLP of
// Embedder.forward) for the reference widths, activations in REGISTERS.
//
//   * the first layer has no GEMM: cross-concat / relative-index / distogram features are per-residue or one-hot, so its
//     pre-activation is four table rows summed (Pi[i] + Pj[j] + R[idx_i - idx_j] + D[bin(|ca_i - ca_j|)]), gathered as whole
//     512 B rows through a wave-private LDS tile, the next tile's rows requested a tile ahead;
//   * layers 2 / 3 are computed TRANSPOSED, D[out feature, pair] = W[out, k] * X^T[k, pair]: weights are the MFMA A operand
//     from two LDS-resident 32 KB images (persistent block), activations the B operand in registers; the C/D fragment of a
//     layer is the B fragment of the next one up to a fixed 16-wise permutation of k folded into the images;
//   * LayerNorm + mask epilogue in registers; the same epilogue emits the first block's pair bias linear_b(z)/sqrt(3) from
//     its output fragments (saves a pass over z).
#include "common.hpp"
#include "kernels.hpp"

#define ET2_CZ 128
#define EE2_WBC 2048     // compact linear_b image of the first block (8 head rows); then a 64 B zero unit, the hi / lo down_z images (8 KB each)
#define EE2_EPI_IMG (EE2_WBC + 64 + 2 * 8192 + 128)  // ... and the down_z bias [32] f32

// logical (row, 16-byte chunk) -> byte offset inside a weight image (bank-conflict-free ds_read_b128)
__host__ __device__ __forceinline__ int et2_off_wide(int row, int c, int row_bytes) {
  return row * row_bytes + ((c ^ (row & 15)) << 4);
}
// position inside a 16-group of k  ->  feature offset inside the 16-group produced by the C/D fragment layout
__host__ __device__ __forceinline__ int et2_perm16(int pos) {
  const int hi = pos >> 3, e = pos & 7;
  return 4 * hi + (e & 3) + 8 * (e >> 2);
}
// One 16 B-per-lane LDS-DMA (global_load_lds_dwordx4; LDS destination = wave-uniform `lds_dst` + lane * 16), written as
// inline asm ON PURPOSE: hipcc's waitcnt pass treats the builtin as a FLAT access that is pending on both counters and
// then forces EVERY later LDS wait to lgkmcnt(0) until the DMA has been waited for.  With the asm form the pass does not see
// the DMA at all: every consumer of DMA'd data therefore sits behind an explicit et2_dma_wait() + barrier.
__device__ __forceinline__ void et2_dma16(const void* gsrc, const char* lds_dst) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(gsrc) : "memory", "m0");
}
__device__ __forceinline__ void et2_dma_wait() {  // every DMA (and ordinary vector-memory op) of this wave retired
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ hx8 lds_frag(const char* slab, int off) {
  return __builtin_bit_cast(hx8, *(const u16x8*)(slab + off));
}
// acc += W_slab[32 x 16*KS] * B[16*KS x 32]: A fragments stream from LDS through a DEPTH-deep register ring so that
// every ds_read_b128 is issued DEPTH MFMAs (= DEPTH*32 cycles) ahead of its consumer.
template <int KS, int ROWB, int DEPTH = 8, int ABL = 0>
__device__ __forceinline__ void mma_slab(f32x16& acc, const char* slab, int li, int hi, const hx8* Bf) {
  hx8 ring[DEPTH];
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) ring[s] = lds_frag(slab, et2_off_wide(li, 2 * s + hi, ROWB));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    acc = fd_mfma32(ring[s % DEPTH], Bf[s], acc);
    if (s + DEPTH < KS) ring[s % DEPTH] = lds_frag(slab, et2_off_wide(li, 2 * (s + DEPTH) + hi, ROWB));
    __builtin_amdgcn_sched_barrier(0);  // pin the MFMA / ds_read interleave (hipcc otherwise sinks the reads)
  }
}

// ---- VALU-lean pieces for the embedder (its waves are issue-bound: ~1500 VALU instructions per 32-pair tile before) ----
typedef float ee_f32x2 __attribute__((ext_vector_type(2)));
typedef fd_h ee_hx2 __attribute__((ext_vector_type(2)));
typedef unsigned ee_u32x2 __attribute__((ext_vector_type(2)));
typedef unsigned ee_u32x4 __attribute__((ext_vector_type(4)));
typedef short ee_s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned ee_cvt_pk(float lo, float hi) {  // one v_cvt_pk_bf16_f32
  return __builtin_bit_cast(unsigned, __builtin_convertvector(ee_f32x2{lo, hi}, ee_hx2));
}
// relu + bf16 of an accumulator tile (bias already in it): conversion first, then max(x, 0) on the bf16 bit patterns as signed
// 16-bit integers (negative values have the sign bit set): 8 + 8 instructions instead of 16 + 16 + 8
__device__ __forceinline__ void ee_hand_off(const f32x16& acc, hx8& h0, hx8& h1) {
  ee_u32x4 w0, w1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    w0[k] = ee_cvt_pk(acc[2 * k], acc[2 * k + 1]);
    w1[k] = ee_cvt_pk(acc[8 + 2 * k], acc[8 + 2 * k + 1]);
  }
  const ee_s16x8 zero = {0, 0, 0, 0, 0, 0, 0, 0};
  h0 = __builtin_bit_cast(hx8, __builtin_elementwise_max(__builtin_bit_cast(ee_s16x8, w0), zero));
  h1 = __builtin_bit_cast(hx8, __builtin_elementwise_max(__builtin_bit_cast(ee_s16x8, w1), zero));
}
// LayerNorm epilogue of the embedder in packed fp32 math (one pass: sum and sum of squares; the layer bias is already in Y):
// same staging / stores / pair-bias emission as ln_epilogue_staged
// pz_tile (round 6): when set, the epilogue also emits pair_z = down_z(z') + b of the first block's IPA for this row's 32 keys (8 key
// groups of 256 B, layout fd_pz_bytes; groups >= ng_valid are beyond N): z' is the A operand of these MFMAs, so a lane (d, half) ends
// up with four consecutive keys per register quad = 8 B of the image; hi + lo weight fragments follow the compact linear_b image in LDS
template <bool TRACE>
__device__ __forceinline__ void ee_ln_epilogue(f32x16 (&Y)[4], const float* gamma_l, const float* beta_l, float em, int li, int hi,
                                               int lane, char* stage, half_t* __restrict__ z_tile, int nvalid,
                                               float* __restrict__ tr_row, bool valid, const char* wb_lds, const f32x4 bbv,
                                               float* __restrict__ bias_out, int H, long bidx, int ii, int jj, int nt,
                                               half_t* __restrict__ pz_tile, int ng_valid) {
  ee_f32x2 u1 = {0.f, 0.f}, u2 = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 16; r += 2) {
      const ee_f32x2 y = {Y[t][r], Y[t][r + 1]};
      u1 += y;
      u2 = __builtin_elementwise_fma(y, y, u2);
    }
  float s1 = u1[0] + u1[1], s2 = u2[0] + u2[1];
  s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 32, 64);
  const float mu = s1 * (1.0f / ET2_CZ);
  const float rstd = 1.0f / sqrtf(fmaxf(s2 * (1.0f / ET2_CZ) - mu * mu, 0.f) + 1e-5f);
  const ee_f32x2 sa = {rstd, rstd}, sc = {-mu * rstd, -mu * rstd}, em2 = {em, em};
#ifndef EE2_ABL
#define EE2_ABL 0  // timing ablations (tools/micro/ee2_bench.hip; results wrong): what an un-transposed layer-3 epilogue could save at most -
#endif             // 1 gamma / beta without their LDS reads, 2 no staging round trip of the z' tile, 4 layer biases without their LDS reads
  ee_u32x4 zB[8];  // half-precision z' as B fragments
  // (gamma, beta) of a feature group come from LDS one group AHEAD of their use; the interleave is pinned: left alone hipcc
  // emits read -> s_waitcnt lgkmcnt(0) -> use for each of the 16 groups (16 exposed LDS round trips per tile)
  f32x4 gq[2], bq[2];
  gq[0] = (EE2_ABL & 1) ? f32x4{1.f, 1.f, 1.f, 1.f} : *(const f32x4*)(gamma_l + 4 * hi);
  bq[0] = (EE2_ABL & 1) ? f32x4{0.f, 0.f, 0.f, 0.f} : *(const f32x4*)(beta_l + 4 * hi);
#pragma unroll
  for (int idx = 0; idx < 16; ++idx) {
    const int t = idx >> 2, g = idx & 3, f0 = 32 * t + 8 * g + 4 * hi;
    if (idx + 1 < 16) {
      const int f1 = 32 * ((idx + 1) >> 2) + 8 * ((idx + 1) & 3) + 4 * hi;
      if (!(EE2_ABL & 1)) {
        gq[(idx + 1) & 1] = *(const f32x4*)(gamma_l + f1);
        bq[(idx + 1) & 1] = *(const f32x4*)(beta_l + f1);
      }
    }
    const f32x4 gm = gq[idx & 1], bt = bq[idx & 1];
    ee_f32x2 o0 = {Y[t][4 * g], Y[t][4 * g + 1]}, o1 = {Y[t][4 * g + 2], Y[t][4 * g + 3]};
    o0 = __builtin_elementwise_fma(o0, sa, sc);
    o1 = __builtin_elementwise_fma(o1, sa, sc);
    o0 = __builtin_elementwise_fma(o0, ee_f32x2{gm[0], gm[1]}, ee_f32x2{bt[0], bt[1]}) * em2;
    o1 = __builtin_elementwise_fma(o1, ee_f32x2{gm[2], gm[3]}, ee_f32x2{bt[2], bt[3]}) * em2;
    const ee_u32x2 ow = {ee_cvt_pk(o0[0], o0[1]), ee_cvt_pk(o1[0], o1[1])};
    // features f0..f0+3 = bytes 2 f0 .. 2 f0 + 7 of the pair's row: 16 B unit 4t + g, half hi; unit u of row r at u ^ (r & 15)
    if (!(EE2_ABL & 2)) *(ee_u32x2*)(stage + li * 256 + (((4 * t + g) ^ (li & 15)) << 4) + 8 * hi) = ow;
    zB[2 * t + (g >> 1)][2 * (g & 1)] = ow[0];
    zB[2 * t + (g >> 1)][2 * (g & 1) + 1] = ow[1];
    if (TRACE && valid) *(f32x4*)(tr_row + f0) = f32x4{o0[0], o0[1], o1[0], o1[1]};
    __builtin_amdgcn_sched_barrier(0);
  }
  // every LDS operand of the tail is requested before the first use: the 8 row segments of the z tile (other lanes' writes above:
  // DS operations of a wave execute in order) and the 8 linear_b fragments
  const int sr = lane >> 4, sc16 = lane & 15;
  u16x8 zrow[8];
#pragma unroll
  for (int it = 0; it < 8; ++it) {
    const int r = 4 * it + sr;
    zrow[it] = (EE2_ABL & 2) ? __builtin_bit_cast(u16x8, zB[it]) : *(const u16x8*)(stage + r * 256 + ((sc16 ^ (r & 15)) << 4));
  }
  const bool full = nvalid == 32;  // wave-uniform: 9 of 10 tiles at N = 300 take the branch-free stores
  if (wb_lds) {
    hx8 wf[8];
    {  // compact image (8 head rows): lanes >= 8 read the zero unit behind it
      const int wl = li < 8 ? hi * 128 + li * 16 : EE2_WBC, ws = li < 8 ? 256 : 0;
#pragma unroll
      for (int s = 0; s < 8; ++s) wf[s] = lds_frag(wb_lds, wl + s * ws);
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 accb;
#pragma unroll
    for (int r = 0; r < 16; ++r) accb[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 8; ++s) accb = fd_mfma32(wf[s], __builtin_bit_cast(hx8, zB[s]), accb);
    float* bo = bias_out + fd_bias_frag_off(bidx * H + 4 * hi, nt, ii, valid ? jj : 0);  // 32 lanes = 32 consecutive keys: 128 B rows
    const long hstride = (long)nt * nt * 1024;  // floats per (sample, head)
    if (full && H == 8) {
#pragma unroll
      for (int r = 0; r < 4; ++r) bo[r * hstride] = accb[r] + bbv[r];
    } else if (valid) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * hi + r < H) bo[r * hstride] = accb[r] + bbv[r];
    }
    if (pz_tile) {
      const char* dzh = wb_lds + EE2_WBC + 64;
      const float bd = *(const float*)(dzh + 2 * 8192 + 4 * li);
      f32x16 accd;
#pragma unroll
      for (int r = 0; r < 16; ++r) accd[r] = bd;
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {  // hi image, then lo image: 8 fragments each, requested a batch ahead of their MFMAs
#pragma unroll
        for (int s = 0; s < 8; ++s) wf[s] = lds_frag(dzh, h2 * 8192 + s * 1024 + lane * 16);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 8; ++s) accd = fd_mfma32(__builtin_bit_cast(hx8, zB[s]), wf[s], accd);
      }
      // registers 4 g .. 4 g + 3 = keys 8 g + 4 hi + q of the tile = key group 2 g + hi, channel d = li
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const ee_u32x2 ow = {ee_cvt_pk(accd[4 * g], accd[4 * g + 1]), ee_cvt_pk(accd[4 * g + 2], accd[4 * g + 3])};
        if (2 * g + hi < ng_valid) *(ee_u32x2*)(pz_tile + ((2 * g + hi) * 32 + li) * 4) = ow;
      }
    }
  }
  if (full) {
#pragma unroll
    for (int it = 0; it < 8; ++it) *(u16x8*)(z_tile + (4 * it + sr) * ET2_CZ + 8 * sc16) = zrow[it];
  } else {
#pragma unroll
    for (int it = 0; it < 8; ++it)
      if (4 * it + sr < nvalid) *(u16x8*)(z_tile + (4 * it + sr) * ET2_CZ + 8 * sc16) = zrow[it];
  }
}

// edge_embed2_kernel — bf16 pair branch of Embedder.forward (framedipt/model/score_network.py:98-105,173-196) in the
// same register-resident style.  Layer 1 has no GEMM: the cross-concat / relative-index / distogram features are
// one-hot or per-residue, so h1 = relu(Pi[i] + Pj[j] + R[idx_i - idx_j] + D[bin(|ca_i - ca_j|)]) is four table rows
// summed directly in B-fragment layout.  Layers 2 and 3 (128x128 each, 64 KB bf16 together) stay RESIDENT in LDS for
// the whole persistent block, so there is no per-tile barrier: waves loop over 32-pair tiles independently.
#define EE2_IMG (128 * 128 * 2)  // 32 KB per layer

__global__ void ee2_build_images_kernel(const float* __restrict__ w2, const float* __restrict__ w3,
                                        half_t* __restrict__ img) {
  const int n_chunks = 2 * EE2_IMG / 16;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n_chunks; g += gridDim.x * blockDim.x) {
    const int layer = g / (EE2_IMG / 16), q = g % (EE2_IMG / 16);
    const int row = q / 16, cp = q % 16, c = cp ^ (row & 15);  // row = out feature (slab = row/32), 256-byte rows
    const float* src = layer == 0 ? w2 : w3;
    for (int e = 0; e < 8; ++e) {
      const int k = c * 8 + e;
      const int col = layer == 0 ? k : (k & ~15) + et2_perm16(k & 15);
      img[(long)g * 8 + e] = f2h(src[(long)row * 128 + col]);
    }
  }
}
int fd_ee2_build_images(const float* w2, const float* w3, void* img, hipStream_t st) {
  hipLaunchKernelGGL(ee2_build_images_kernel, dim3(16), dim3(256), 0, st, w2, w3, (half_t*)img);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
size_t fd_ee2_image_bytes() { return 2 * EE2_IMG; }

// 512-thread persistent blocks: 8 independent waves (two per SIMD) share the 64 KB weight images; every wave owns an
// 8 KB LDS tile that transposes between "whole 512 B table rows per 32 lanes" (the global side) and MFMA fragments.
//
// Work decomposition (round 2): a tile is one query row i against 32 consecutive keys j0..j0+31 of one sample, and a wave walks
// CONSECUTIVE rows i of one (sample, key tile) group.  Of the four table rows a pair sums, only R[idx_i - idx_j] is then a per-tile
// gather from L2 (16 KB): Pj[j] of the 32 keys stays in 64 registers for the whole walk, Pi[i] is one 512 B row per tile and the
// distogram table D (num_bins + 1 rows) sits in LDS.  The previous generation walked flat 32-pair tiles and fetched all four rows of
// every pair (64 KB per tile through the 64 B/clk L2 -> CU path: 1.5 GB per launch).  The next row's R rows and Pi row are
// requested before the LayerNorm epilogue of the current one.
#define EE2_THREADS 512
// phase profile (-DEE2_PROF, tools/micro/ee2_bench.hip): cycles of wave 0 of every block, accumulated over its tiles
#ifdef EE2_PROF
__device__ unsigned ee2_prof[256 * 8];
#define EE2_STAMP(k)                                             \
  do {                                                           \
    __builtin_amdgcn_sched_barrier(0);                           \
    const unsigned t_ = (unsigned)__builtin_amdgcn_s_memtime(); \
    ph[k] += t_ - tlast;                                         \
    tlast = t_;                                                  \
    __builtin_amdgcn_sched_barrier(0);                           \
  } while (0)
#else
#define EE2_STAMP(k) \
  do {               \
  } while (0)
#endif
#ifndef EE2_RESIDENT
#define EE2_RESIDENT 0   // Pj rows of the key tile stay in registers for the whole walk
#endif
#ifndef EE2_EARLY
#define EE2_EARLY 0      // R rows of the next tile requested before the LayerNorm epilogue (the rest after it: register budget)
#endif
#define EE2_MAXB 63      // distogram bins (edges in LDS)
#define EE2_MAXB_LDS 39  // ... with the table rows in LDS as well (20 KB)
#define EE2_LDS_BASE (2 * EE2_IMG + 8 * 8192 + 4 * ET2_CZ * 4 + EE2_EPI_IMG + 256)  // ... + the epilogue's images + distogram edges
#define EE2_LDS_MAX 163840
template <bool DLDS, bool TRACE>
__global__ __launch_bounds__(EE2_THREADS, 1) void edge_embed2_kernel(EdgeEmbedArgs a, const char* __restrict__ img, int nt, int wpg,
                                                                     int rpw, int n_items) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef EE2_PROF
  unsigned ph[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tlast = (unsigned)__builtin_amdgcn_s_memtime();
#endif
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hi = lane >> 5, li = lane & 31;
  char* stage = smem + 2 * EE2_IMG + wave * 8192;
  float* vec = (float*)(smem + 2 * EE2_IMG + 8 * 8192);  // [b2 | b3 | gamma | beta] x 128
  char* wbl = (char*)(vec + 4 * ET2_CZ);                 // compact image of linear_b | zero unit | down_z hi | lo | bias (all optional)
  float* edg = (float*)(wbl + EE2_EPI_IMG);              // [num_bins + 1] distogram edges, the last one 1e8
  float* dl = edg + 64;                                  // DLDS: [num_bins + 1][128] distogram rows of the first layer
  if (tid <= a.num_bins) edg[tid] = tid < a.num_bins ? a.edges[tid] : 1e8f;
  if (a.wb_img && tid < EE2_WBC / 16) et2_dma16((const char*)a.wb_img + tid * 16, wbl + (tid & ~63) * 16);
  if (tid < 16) *(unsigned*)(wbl + EE2_WBC + 4 * tid) = 0u;
  if (a.pz_out) {
    et2_dma16((const char*)a.wdz_img + tid * 16, wbl + EE2_WBC + 64 + (tid & ~63) * 16);
    et2_dma16((const char*)a.wdz_img_lo + tid * 16, wbl + EE2_WBC + 64 + 8192 + (tid & ~63) * 16);
    if (tid < 32) *(float*)(wbl + EE2_WBC + 64 + 2 * 8192 + 4 * tid) = a.bdz[tid];
  }
  for (int u = 0; u < 2 * EE2_IMG / 16 / EE2_THREADS; ++u)
    et2_dma16(img + (size_t)(u * EE2_THREADS + tid) * 16, smem + (size_t)(u * EE2_THREADS + (tid & ~63)) * 16);
  if (tid < 4 * ET2_CZ) {
    const int which = tid >> 7, c = tid & 127;
    vec[tid] = which == 0 ? a.b2[c] : (which == 1 ? a.b3[c] : (which == 2 ? a.gamma[c] : a.beta[c]));
  }
  if (DLDS) {
    f32x4 t[3];
#pragma unroll
    for (int u = 0; u < 3; ++u) {  // (EE2_MAXB_LDS + 1) * 32 units of 16 B <= 3 * 512
      const int v = u * EE2_THREADS + tid;
      if (v < (a.num_bins + 1) * 32) t[u] = *(const f32x4*)(a.dtab + (long)v * 4);
    }
#pragma unroll
    for (int u = 0; u < 3; ++u) {
      const int v = u * EE2_THREADS + tid;
      if (v < (a.num_bins + 1) * 32) *(f32x4*)(dl + (long)v * 4) = t[u];
    }
  }
  et2_dma_wait();
  __syncthreads();
  EE2_STAMP(0);
  const int N = a.N, nb = a.num_bins;
  const float* b2row = vec + 4 * hi;
  const f32x4 bbv = a.wb_img ? f32x4{a.bb[4 * hi], a.bb[4 * hi + 1], a.bb[4 * hi + 2], a.bb[4 * hi + 3]} : f32x4{0.f, 0.f, 0.f, 0.f};
  // distogram bin of one distance: calc_distogram's strict inequalities against the stored edges (last upper edge 1e8); the
  // candidate comes from the edge spacing, its neighbours are re-tested, so the result is the one a full scan finds
  const float e0 = edg[0], inv_step = nb > 1 ? 1.0f / (edg[1] - edg[0]) : 0.f;
  auto bin_of = [&](float d) {
    int k0 = (int)((d - e0) * inv_step);
    k0 = k0 < 1 ? 1 : (k0 > nb - 2 ? nb - 2 : k0);
    int bin = nb;
#pragma unroll
    for (int k = -1; k <= 1; ++k) {
      const int kk = k0 + k;
      if (kk >= 0 && kk < nb && d > edg[kk] && d < edg[kk + 1]) bin = kk;
    }
    return bin;
  };
  for (int item = blockIdx.x * 8 + wave; item < n_items; item += gridDim.x * 8) {
    // ---- group (sample b, key tile jt): per-key state of the 32 keys, resident for the whole walk
    const int grp = item / wpg, part = item - grp * wpg;
    const int b = grp / nt, jt = grp - b * nt, j0 = jt * 32;
    const int i_first = part * rpw, i_end = (i_first + rpw < N) ? i_first + rpw : N;
    if (i_first >= i_end) continue;
    const int nvalid = N - j0 < 32 ? N - j0 : 32;
    const bool valid = li < nvalid;
    const int j = valid ? j0 + li : N - 1;
    const long bj = (long)b * N + j;
    const int sj = a.seq_idx[bj];
    const float mj = a.res_mask[bj];
    const float cj0 = a.sc_ca[bj * 3], cj1 = a.sc_ca[bj * 3 + 1], cj2 = a.sc_ca[bj * 3 + 2];
    const float* pj_base = a.pj + ((long)b * N + j0) * ET2_CZ + 4 * li;  // row r of the tile at + min(r, nvalid - 1) * 128
    // per-row scalars (wave-uniform addresses: scalar loads) -> this lane's table row ids
    // per-row scalars: requested (scalar loads) at the top of the previous row, turned into table row ids after its gather
    struct RowIn { int si; float mi, c0, c1, c2; };
    auto row_request = [&](int i) {
      const long bi = (long)b * N + i;
      return RowIn{a.seq_idx[bi], a.res_mask[bi], a.sc_ca[bi * 3], a.sc_ca[bi * 3 + 1], a.sc_ca[bi * 3 + 2]};
    };
    auto row_ids = [&](const RowIn& r, int& rel, int& bin, float& msk) {
      msk = r.mi * mj;
      rel = b * a.n_rel + r.si - sj + a.rel_off;
      const float dx = r.c0 - cj0, dy = r.c1 - cj1, dz = r.c2 - cj2;
      bin = bin_of(sqrtf(dx * dx + dy * dy + dz * dz));
    };
    f32x4 RR[16], PJ[16], PI;
#if EE2_RESIDENT
#pragma unroll
    for (int it = 0; it < 16; ++it) {
      const int r = 2 * it + hi;
      PJ[it] = *(const f32x4*)(pj_base + (r < nvalid ? r : nvalid - 1) * ET2_CZ);
    }
#endif
    auto request = [&](int i, int rel, const int it0, const int it1) {
      if (it0 == 0) PI = *(const f32x4*)(a.pi + ((long)b * N + i) * ET2_CZ + 4 * li);
#pragma unroll
      for (int it = it0; it < it1; ++it) {
        const int r = 2 * it + hi;
        const int rrel = __shfl(rel, r, 64);
        RR[it] = *(const f32x4*)(a.rtab + (long)rrel * ET2_CZ + 4 * li);
#if !EE2_RESIDENT
        PJ[it] = *(const f32x4*)(pj_base + (r < nvalid ? r : nvalid - 1) * ET2_CZ);  // (the same 16 KB for every row of the walk: L2 hits)
#endif
      }
    };
    int rel, bin;
    float msk;
    row_ids(row_request(i_first), rel, bin, msk);
    request(i_first, rel, 0, 16);
    EE2_STAMP(1);
    for (int i = i_first; i < i_end; ++i) {
      // Lane-derived LDS / global offsets are recomputed per row: left to itself hipcc hoists ~100 of them out of this loop and
      // spills them (every reload is a vmcnt wait in front of the matrix phase).
      int lane_o = lane;
      asm volatile("" : "+v"(lane_o));
      const int hi = lane_o >> 5, li = lane_o & 31;
      const float* b2row = vec + 4 * hi;
      const int i_next = i + 1 < i_end ? i + 1 : i;
      const RowIn rin = row_request(i_next);
      // ---- layer 1 has no GEMM: h1 = relu(Pi[i] + Pj[j] + R[rel] + D[bin]).  Two pairs per instruction: lanes 0..31 / 32..63
      // hold one whole 512 B row each.  The 16 bin shuffles, then the 16 distogram rows, are issued as batches (one LDS round trip
      // each instead of one per pair of rows).
#pragma unroll
      for (int half = 0; half < 4; ++half) {
      int rb[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) rb[q] = __shfl(bin, 2 * (4 * half + q) + hi, 64);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 DD[4];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        DD[q] = DLDS ? *(const f32x4*)(dl + rb[q] * ET2_CZ + 4 * li) : *(const f32x4*)(a.dtab + (long)rb[q] * ET2_CZ + 4 * li);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int it = 4 * half + q, r = 2 * it + hi;
        const f32x4 x1 = PI, x2 = PJ[it], x3 = RR[it], x4 = DD[q];
        // packed adds, conversion, then relu on the half-precision bit patterns (v_pk_max_i16)
        const ee_f32x2 sA = (ee_f32x2{x1[0], x1[1]} + ee_f32x2{x2[0], x2[1]}) + (ee_f32x2{x3[0], x3[1]} + ee_f32x2{x4[0], x4[1]});
        const ee_f32x2 sB = (ee_f32x2{x1[2], x1[3]} + ee_f32x2{x2[2], x2[3]}) + (ee_f32x2{x3[2], x3[3]} + ee_f32x2{x4[2], x4[3]});
        typedef short s16x4 __attribute__((ext_vector_type(4)));
        const ee_u32x2 cw = {ee_cvt_pk(sA[0], sA[1]), ee_cvt_pk(sB[0], sB[1])};
        const ee_u32x2 pk = __builtin_bit_cast(ee_u32x2, __builtin_elementwise_max(__builtin_bit_cast(s16x4, cw), s16x4{0, 0, 0, 0}));
        // [32 pairs][256 B] tile, 16 B unit u of row r at u ^ (r & 15)
        *(ee_u32x2*)(stage + r * 256 + (((li >> 1) ^ (r & 15)) << 4) + 8 * (li & 1)) = pk;
      }
      }
      EE2_STAMP(2);
      const float msk_cur = msk;
      row_ids(rin, rel, bin, msk);  // the next row's ids (this row's were consumed by the gather above)
      hx8 H1[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) H1[s] = lds_frag(stage, li * 256 + (((2 * s + hi) ^ (li & 15)) << 4));
      hx8 H2[8];
#pragma unroll
      for (int T = 0; T < 4; ++T) {
        f32x16 acc;  // starts as the layer bias (LDS reads straight into the accumulator: no VALU)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 bv = *(const f32x4*)(b2row + 32 * T + 8 * g);
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[4 * g + q] = bv[q];
        }
        mma_slab<8, 256>(acc, smem + T * 32 * 256, li, hi, H1);
        ee_hand_off(acc, H2[2 * T], H2[2 * T + 1]);
      }
      EE2_STAMP(3);
      f32x16 Y[4];
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {  // starts as the layer bias b3
          const f32x4 bv = (EE2_ABL & 4) ? f32x4{0.1f, 0.2f, 0.3f, 0.4f} : *(const f32x4*)(vec + ET2_CZ + 4 * hi + 32 * t + 8 * g);
#pragma unroll
          for (int q = 0; q < 4; ++q) Y[t][4 * g + q] = bv[q];
        }
        mma_slab<8, 256>(Y[t], smem + EE2_IMG + t * 32 * 256, li, hi, H2);
      }
      EE2_STAMP(4);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      request(i_next, rel, 0, EE2_EARLY);  // in flight under the epilogue
      __builtin_amdgcn_sched_barrier(0);
      const int rel_next = rel;
      const long prow = ((long)b * N + i) * N + j0;  // first pair of the tile
      ee_ln_epilogue<TRACE>(Y, vec + 2 * ET2_CZ, vec + 3 * ET2_CZ, msk_cur, li, hi, lane, stage, (half_t*)a.z_out + prow * ET2_CZ, nvalid,
                     TRACE ? a.trace + (prow + (valid ? li : 0)) * ET2_CZ : nullptr, valid, a.wb_img ? wbl : nullptr, bbv,
                     a.bias_out, a.H, b, i, j0 + li, nt,
                     a.pz_out ? a.pz_out + (((long)b * N + i) * ((N + 3) >> 2) + (j0 >> 2)) * 128 : nullptr, (nvalid + 3) >> 2);
      asm volatile("" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      request(i_next, rel_next, EE2_EARLY, 16);
      EE2_STAMP(5);
    }
  }
#ifdef EE2_PROF
  if (tid == 0 && blockIdx.x < 256)
    for (int k = 0; k < 8; ++k) ee2_prof[blockIdx.x * 8 + k] = ph[k];
#endif
}

int fd_edge_embed2(const EdgeEmbedArgs& a, const void* img, hipStream_t st) {
  if (a.num_bins > EE2_MAXB || a.num_bins < 3) return FDIPT_EINVAL;
  if (a.pz_out && (!a.wb_img || !a.wdz_img || !a.wdz_img_lo || !a.bdz)) return FDIPT_EINVAL;
  const bool dlds = a.num_bins <= EE2_MAXB_LDS && EE2_LDS_BASE + (a.num_bins + 1) * ET2_CZ * 4 <= EE2_LDS_MAX;  // (else the distogram rows come from L2)
  const int lds = EE2_LDS_BASE + (dlds ? (a.num_bins + 1) * ET2_CZ * 4 : 0);
  typedef void (*kern_t)(EdgeEmbedArgs, const char*, int, int, int, int);
  static const kern_t kerns[4] = {edge_embed2_kernel<false, false>, edge_embed2_kernel<false, true>, edge_embed2_kernel<true, false>,
                                  edge_embed2_kernel<true, true>};
  const int kid = 2 * dlds + (a.trace != nullptr);
  static FdPerDevice attr_dev[4];
  const int dev_ = fd_device();
  if (!attr_dev[kid].get(dev_)) {
    if (hipFuncSetAttribute((const void*)kerns[kid], hipFuncAttributeMaxDynamicSharedMemorySize, EE2_LDS_MAX) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev[kid].set(dev_, 1);
  }
  // one work item = (sample, key tile of 32, range of rpw consecutive query rows); wpg items per group so that ~2048 waves are busy
  const int nt = cdiv(a.N, 32), n_groups = a.B * nt, n_waves = 256 * 8;
  int wpg = n_waves / n_groups < 1 ? 1 : n_waves / n_groups;
  if (wpg > a.N) wpg = a.N;
  const int rpw = cdiv(a.N, wpg);
  wpg = cdiv(a.N, rpw);
  const int n_items = n_groups * wpg;
  const int cus = a.reserve_cus > 0 && a.reserve_cus < 248 ? (256 - a.reserve_cus) & ~7 : 256;  // (whole XCD rounds of 8)
  const int grid = cdiv(n_items, 8) < cus ? cdiv(n_items, 8) : cus;
  hipLaunchKernelGGL(kerns[kid], dim3(grid), dim3(EE2_THREADS), lds, st, a, (const char*)img, nt, wpg, rpw, n_items);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
