// edge_transition3.hip — bf16 EdgeTransition (framedipt/model/ipa_pytorch.py:84-102), second generation of the
// register-resident kernel (edge_transition2.hip), re-tiled so that TWO waves share every SIMD.
//
// edge_transition2 gives a wave 32 pairs and the whole 512-register file: one wave per SIMD, so everything a wave does
// besides issuing MFMAs (LDS-DMA issue ≈ 60 cycles per KB, bias / ReLU / pack epilogues, barrier skew, the tile prologue
// and the LayerNorm epilogue) leaves its matrix core idle — 37 % MFMA utilisation in the phase profile.  Here a wave owns
// 16 pairs and works with v_mfma_f32_16x16x32_bf16:
//   * activations per wave: x 32 + h1 48 + h2 48 + y 32 registers  ->  < 256 VGPRs  ->  8 waves per block, 2 per SIMD:
//     one wave's non-MFMA work hides under the other wave's MFMA stream;
//   * same transposed scheme, D^T[feature, pair] = W[feature, k] X^T[k, pair]: weights are the A operand (LDS), activations
//     the B operand (registers).  C/D of a 16-feature tile holds features 4q + r (q = lane >> 4) of pair lane & 15; a B
//     fragment of the next layer (32 k) wants 8 k per lane: the C/D registers of a PAIR of tiles, i.e. k position 8q + e
//     = feature 16 T0 + 4q + e (e < 4) or 16 T1 + 4q + e - 4 — a fixed permutation folded into the weight stream;
//   * with h2 affordable in registers the three layers run strictly one after the other (no x-part / h-part interleave of
//     the final layer): y = Wf[:, z|e_j] x + Wf[:, h] h2 is one K = 640 product;
//   * weight stream: lane-linear 1 KB fragments [feature tile][k-step][lane][8 bf16] (no swizzle: a fragment read is
//     ds_read_b128 at lane * 16), 640 KB per 128 pairs in 13 chunks through a 2 x 64 KB LDS double buffer by LDS-DMA.
// Concat-free exactly as edge_transition2: the e_i columns of layer 1 / the final layer are per-residue rows A1[i], Af[i].
#include "common.hpp"
#include "kernels.hpp"

#ifndef E3_ABL
#define E3_ABL 0  // timing ablations (tools/micro/et3_bench.hip): 1 no epilogue at all, 2 no MFMA, 4 no weight DMA
#endif
#define E3_CZ 128
#define E3_CB 128
#define E3_H 384
#define E3_THREADS 512
#define E3_CHUNK 65536
// chunks: layer 1 = 3 x 4 tile pairs (K 256: 8 k-steps, 16 KB per pair); layer 2 = 6 x 2 pairs (K 384: 12 k-steps, 24 KB per
// pair); final = 4 x 1 pair (K 640: 20 k-steps, 40 KB per pair)
#define E3_L1_BYTES (24 * 8 * 1024)
#define E3_L2_BYTES (24 * 12 * 1024)
#define E3_LF_BYTES (8 * 20 * 1024)
#define E3_STREAM_BYTES (E3_L1_BYTES + E3_L2_BYTES + E3_LF_BYTES)
#define E3_BROWS 4
#define E3_BROW_BYTES 2048
#define E3_LDS (2 * E3_CHUNK + 2 * E3_BROWS * E3_BROW_BYTES + 1536 + 1024 + 4096)  // ... + linear_b image of the next block

typedef __attribute__((ext_vector_type(4))) float e3_f32x4;
typedef fd_h e3_hx4 __attribute__((ext_vector_type(4)));
typedef unsigned int e3_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int e3_u32x2 __attribute__((ext_vector_type(2)));

// k position (0..31) of a k-step whose B fragment is the C/D hand-off of tiles (2s, 2s+1) -> feature offset in [0, 32)
__host__ __device__ __forceinline__ int e3_chain_feat(int pos) {
  const int q = pos >> 3, e = pos & 7;
  return e < 4 ? 4 * q + e : 16 + 4 * q + (e - 4);
}

// ------------------------------------------------------------------ prepare: weight stream image
// w1 [384,384], w2 [384,384], wf [128,384] fp32 row-major (out, in); in = [z(0:128) | e_i(128:256) | e_j(256:384)]
__global__ void et3_build_stream_kernel(const float* __restrict__ w1, const float* __restrict__ w2,
                                        const float* __restrict__ wf, half_t* __restrict__ stream) {
  const int n_units = E3_STREAM_BYTES / 16;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < n_units; g += gridDim.x * blockDim.x) {
    int frag = g >> 6;
    const int lane = g & 63, m = lane & 15, q = lane >> 4;
    const float* src;
    int n, s, mode;  // mode 0: x columns (natural), 1: chained hidden columns, 2: final layer (x then chained h2)
    if (frag < 24 * 8) { src = w1; n = 16 * (frag / 8) + m; s = frag % 8; mode = 0; }
    else if (frag < 24 * 8 + 24 * 12) { frag -= 24 * 8; src = w2; n = 16 * (frag / 12) + m; s = frag % 12; mode = 1; }
    else { frag -= 24 * 8 + 24 * 12; src = wf; n = 16 * (frag / 20) + m; s = frag % 20; mode = 2; }
    half_t out[8];
    for (int e = 0; e < 8; ++e) {
      const int pos = 8 * q + e;
      int col;
      if (mode == 0 || (mode == 2 && s < 8)) {
        const int k = 32 * s + pos;  // x = [z | e_j]
        col = k < E3_CZ ? k : (E3_CZ + E3_CB) + (k - E3_CZ);
      } else {
        const int sh = mode == 1 ? s : s - 8;
        col = 32 * sh + e3_chain_feat(pos);  // hidden feature index (w2: all 384 inputs are h1; wf: columns 0..383 are h2)
      }
      out[e] = f2h(src[(long)n * E3_H + col]);
    }
    for (int e = 0; e < 8; ++e) stream[(long)g * 8 + e] = out[e];
  }
}
int fd_et3_build_stream(const float* w1, const float* w2, const float* wf, void* stream, hipStream_t st) {
  hipLaunchKernelGGL(et3_build_stream_kernel, dim3(160), dim3(256), 0, st, w1, w2, wf, (half_t*)stream);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
size_t fd_et3_stream_bytes() { return E3_STREAM_BYTES; }

// linear_b of the NEXT block's attention (ipa_pytorch.py:247,256-257) as 4 fragments [k-step][lane][8]: row = head (8 of 16
// used), k in the hand-off order of the LayerNorm output tiles; `scale` = sqrt(1/3)
__global__ void et3_bias_image_kernel(const float* __restrict__ wb, int H, float scale, half_t* __restrict__ img) {
  for (int g = threadIdx.x; g < 4 * 64; g += blockDim.x) {
    const int s = g >> 6, lane = g & 63, m = lane & 15, q = lane >> 4;
    for (int e = 0; e < 8; ++e)
      img[g * 8 + e] = m < H ? f2h(wb[m * E3_CZ + 32 * s + e3_chain_feat(8 * q + e)] * scale) : (half_t)0;
  }
}
int fd_et3_build_bias_image(const float* wb, int H, float scale, void* img, hipStream_t st) {
  if (H > 8) return FDIPT_ESIZE;
  hipLaunchKernelGGL(et3_bias_image_kernel, dim3(1), dim3(256), 0, st, wb, H, scale, (half_t*)img);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}

// ------------------------------------------------------------------ device helpers
typedef __attribute__((address_space(3))) void e3_lds_t;
// 16 B-per-lane LDS-DMA as inline asm (see edge_transition2.hip: the builtin makes hipcc force lgkmcnt(0) everywhere)
__device__ __forceinline__ void e3_dma16(const void* gsrc, const char* lds_dst) {
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)lds_dst);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(gsrc) : "memory", "m0");
}
__device__ __forceinline__ void e3_dma_wait() {
  __builtin_amdgcn_sched_barrier(0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_sched_barrier(0);
}
__device__ __forceinline__ hx8 e3_frag(const char* p) { return __builtin_bit_cast(hx8, *(const u16x8*)p); }
__device__ __forceinline__ hx8 e3_pack8(const float* v) {
  hx8 o;
#pragma unroll
  for (int e = 0; e < 8; ++e) o[e] = (fd_h)v[e];
  return o;
}
template <int BYTES>
__device__ __forceinline__ void e3_dma_chunk(const char* __restrict__ src, char* dst, int tid) {
#pragma unroll
  for (int u = 0; u < (BYTES / 16 + E3_THREADS - 1) / E3_THREADS; ++u)
    if (!(E3_ABL & 4) && ((u + 1) * E3_THREADS * 16 <= BYTES || (u * E3_THREADS + tid) * 16 < BYTES))
      e3_dma16(src + (size_t)(u * E3_THREADS + tid) * 16, dst + (size_t)(u * E3_THREADS + (tid & ~63)) * 16);
}

// two 16-feature tiles (A, B) against the same B fragments: fragments of the pair are [tile A: KS KB][tile B: KS KB] at `base`
template <int KS>
__device__ __forceinline__ void e3_pair(e3_f32x4& accA, e3_f32x4& accB, const char* base, int lane, const hx8* Bf) {
  constexpr int DEPTH = 3;  // fragments are requested 2 k-steps (4 MFMAs of this wave) ahead of their use (4: slower, 5+: spills)
  const char* pa = base + lane * 16;
  const char* pb = pa + KS * 1024;
  hx8 rA[DEPTH], rB[DEPTH];
#pragma unroll
  for (int s = 0; s < DEPTH - 1; ++s) {
    rA[s] = e3_frag(pa + s * 1024);
    rB[s] = e3_frag(pb + s * 1024);
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    if (s + DEPTH - 1 < KS) {
      rA[(s + DEPTH - 1) % DEPTH] = e3_frag(pa + (s + DEPTH - 1) * 1024);
      rB[(s + DEPTH - 1) % DEPTH] = e3_frag(pb + (s + DEPTH - 1) * 1024);
    }
    if (!(E3_ABL & 2)) {
      accA = fd_mfma16(rA[s % DEPTH], Bf[s], accA);
      accB = fd_mfma16(rB[s % DEPTH], Bf[s], accB);
    } else {
      accA[0] += (float)rA[s % DEPTH][0];
      accB[0] += (float)rB[s % DEPTH][0];
    }
    __builtin_amdgcn_sched_barrier(0);  // pin: 2 ds_reads, 2 MFMAs per k-step (hipcc otherwise sinks every read to its use)
  }
}

// ------------------------------------------------------------------ kernel
// Persistent: one block per CU walks the 128-pair tiles (stride gridDim.x).  At the end of a tile the operands of the NEXT
// tile (first weight chunk, z rows by DMA, e_j rows, A1 | Af rows) are requested before the LayerNorm epilogue of the
// current one runs, so the epilogue's VALU work, its stores and the next tile's memory latency overlap.
struct E3Tile {      // per-lane description of a tile (the 4 lane groups q of a wave hold the same pair); B*N*N < 2^31
  int p;             // this lane's pair (clamped)
  int bi, bj;        // rows b*N + i, b*N + j
  int i_lo;          // first A1/Af row staged for the tile
  bool valid;
};
__device__ __forceinline__ E3Tile e3_tile(int tile, int wave, int n, int N, int n_pairs) {
  E3Tile t;
  const int p_raw = tile * 128 + wave * 16 + n;
  t.valid = p_raw < n_pairs;
  t.p = t.valid ? p_raw : n_pairs - 1;
  t.bi = t.p / N;
  t.bj = (t.bi / N) * N + (t.p - t.bi * N);
  t.i_lo = (tile * 128) / N;
  return t;
}

// requests everything tile `tile` needs before its first MFMA, all by LDS-DMA (no registers, nothing to wait for until the
// tile starts): z rows and e_j rows (bf16) -> xst with the swizzle applied on the source side, first weight chunk -> buffer
// 0, A1 | Af rows -> brow
__device__ __forceinline__ void e3_request(const ET2Args& a, int tile, int tid, int lane, int wave, int n_pairs, int n_rows,
                                           const char* stream, char* smem, char* xst, char* brow) {
  const int N = a.N;
  int pw = tile * 128 + wave * 16;
  if (pw > n_pairs - 1) pw = n_pairs - 1;
  const int rem = n_pairs - 1 - pw;
  const int rmax = rem < 15 ? rem : 15;
  const int bi0 = pw / N;
  const int j0 = pw - bi0 * N;
  const int b0 = bi0 / N;
  const bool last_i = bi0 - b0 * N == N - 1;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    int r = 4 * k + (lane >> 4);
    const int u = (lane & 15) ^ r;
    if (r > rmax) r = rmax;
    e3_dma16(a.z_in + (long)(pw + r) * E3_CZ + 8 * u, xst + k * 1024);
    int rbj = b0 * N + j0 + r;
    if (j0 + r >= N && !last_i) rbj -= N;   // wrap to (i + 1, j - N); past the sample's last row it is the next sample
    e3_dma16(a.e_h16 + (long)rbj * E3_CB + 8 * u, xst + 4096 + k * 1024);
  }
  e3_dma_chunk<E3_CHUNK>(stream, smem, tid);
  {  // A1 | Af rows i_lo .. i_lo+3: 4 x 128 units of 16 B = one per thread
    const int row = tid >> 7, qq = tid & 127;
    int r = (tile * 128) / N + row;
    if (r >= n_rows) r = n_rows - 1;
    const float* src = qq < 96 ? a.a1 + (long)r * E3_H + 4 * qq : a.af + (long)r * E3_CZ + 4 * (qq - 96);
    e3_dma16(src, brow + (size_t)(tid & ~63) * 16);
  }
}

__global__ __launch_bounds__(E3_THREADS, 1) void edge_transition3_kernel(ET2Args a, int n_tiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* brow_lds = smem + 2 * E3_CHUNK;                                       // [2][4][2048]: A1[384] | Af[128] rows
  const float* vec = (const float*)(brow_lds + 2 * E3_BROWS * E3_BROW_BYTES); // b2[384] | gamma[128] | beta[128]
  const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave = tid0 >> 6, n0 = lane0 & 15, q0 = lane0 >> 4;
  const int N = a.N;
  const int n_pairs = a.B * N * N, n_rows = a.B * N;
  const char* stream = (const char*)a.stream;
  // per-wave x stage: [z: 16 rows x 256 B][e: 16 rows x 256 B], unit u of row r at u ^ r
  int tile = blockIdx.x;
  if (tile >= n_tiles) return;
  FD_STAMP(0);
  // ---- first tile: request, small vectors
  {
    const int tid = tid0, lane = lane0;
    e3_request(a, tile, tid, lane, wave, n_pairs, n_rows, stream, smem, smem + E3_CHUNK + wave * 8192, brow_lds);
    if (tid < 160) {
      const float* src = tid < 96 ? a.b2 + 4 * tid : (tid < 128 ? a.gamma + 4 * (tid - 96) : a.beta + 4 * (tid - 128));
      e3_dma16(src, (const char*)vec + (tid & ~63) * 16);
    }
    if (a.wb_img && tid < 256) e3_dma16((const char*)a.wb_img + tid * 16, (const char*)vec + 2560 + (tid & ~63) * 16);
  }
  E3Tile tc = e3_tile(tile, wave, n0, N, n_pairs);
  float em = a.res_mask[tc.bi] * a.res_mask[tc.bj];
  int par = 0;  // parity of the A1 | Af row buffer of the current tile
  e3_dma_wait();
  __syncthreads();
  FD_STAMP(1);
#pragma unroll 1
  for (;;) {
    // opaque per-iteration copies: keep hipcc from hoisting the loop-invariant LDS / global address arithmetic of the whole
    // tile body out of the loop (hundreds of values that would stay live across it and spill)
    int lane = lane0, tid = tid0, n = n0, q = q0;
    asm volatile("" : "+v"(lane), "+v"(tid), "+v"(n), "+v"(q));
    char* xst = smem + E3_CHUNK + (tid >> 6) * 8192;
    // B fragments of x = [z | e_j]: k-step s (32 columns) = units 4 (s & 3) + q of the z / e row of pair n
    hx8 X[8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      X[s] = e3_frag(xst + n * 256 + (((4 * s + q) ^ n) << 4));
      X[4 + s] = e3_frag(xst + 4096 + n * 256 + (((4 * s + q) ^ n) << 4));
    }
    const float* a1l = (const float*)(brow_lds + par * E3_BROWS * E3_BROW_BYTES + (tc.bi - tc.i_lo) * E3_BROW_BYTES) + 4 * q;
    __builtin_amdgcn_s_waitcnt(0x0070);
    __syncthreads();  // every wave has its x fragments: chunk buffer 1 may receive the second chunk

    hx8 H1[12], H2[12];
    e3_f32x4 Y[8];
    size_t soff = 0;
    int buf = 0;
    // ================= layer 1: 3 chunks x 4 tile pairs, K = 256
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if (c < 2) e3_dma_chunk<E3_CHUNK>(stream + soff + E3_CHUNK, smem + (buf ^ 1) * E3_CHUNK, tid);
      else e3_dma_chunk<2 * 24 * 1024>(stream + soff + E3_CHUNK, smem + (buf ^ 1) * E3_CHUNK, tid);  // first layer-2 chunk (48 KB)
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int P = 4 * c + u;  // tiles 2P, 2P+1
        e3_f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
        e3_pair<8>(accA, accB, smem + buf * E3_CHUNK + u * 16384, lane, X);
        const f32x4 bA = *(const f32x4*)(a1l + 32 * P), bB = *(const f32x4*)(a1l + 32 * P + 16);
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaxf(accA[r] + bA[r], 0.f);
          v[4 + r] = fmaxf(accB[r] + bB[r], 0.f);
        }
        H1[P] = e3_pack8(v);
      }
      e3_dma_wait();
      __syncthreads();
      buf ^= 1;
      soff += E3_CHUNK;
    }
    FD_STAMP(2);
    // ================= layer 2: 6 chunks x 2 tile pairs, K = 384 (48 KB per chunk)
    const float* b2l = vec + 4 * q;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
      if (c < 5) e3_dma_chunk<2 * 24 * 1024>(stream + soff + 2 * 24 * 1024, smem + (buf ^ 1) * E3_CHUNK, tid);
      else e3_dma_chunk<40 * 1024>(stream + soff + 2 * 24 * 1024, smem + (buf ^ 1) * E3_CHUNK, tid);  // first final-layer chunk
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int P = 2 * c + u;
        e3_f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
        e3_pair<12>(accA, accB, smem + buf * E3_CHUNK + u * 24576, lane, H1);
        const f32x4 bA = *(const f32x4*)(b2l + 32 * P), bB = *(const f32x4*)(b2l + 32 * P + 16);
        float v[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[r] = fmaxf(accA[r] + bA[r], 0.f);
          v[4 + r] = fmaxf(accB[r] + bB[r], 0.f);
        }
        H2[P] = e3_pack8(v);
      }
      e3_dma_wait();
      __syncthreads();
      buf ^= 1;
      soff += 2 * 24 * 1024;
    }
    FD_STAMP(3);
    // ================= final layer: 4 chunks x 1 tile pair, K = 640: B fragments = x (8) then h2 (12)
    hx8 XF[20];
#pragma unroll
    for (int s = 0; s < 8; ++s) XF[s] = X[s];
#pragma unroll
    for (int s = 0; s < 12; ++s) XF[8 + s] = H2[s];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      if (c < 3) e3_dma_chunk<40 * 1024>(stream + soff + 40 * 1024, smem + (buf ^ 1) * E3_CHUNK, tid);
      e3_f32x4 accA = {0.f, 0.f, 0.f, 0.f}, accB = {0.f, 0.f, 0.f, 0.f};
      e3_pair<20>(accA, accB, smem + buf * E3_CHUNK, lane, XF);
      Y[2 * c] = accA;
      Y[2 * c + 1] = accB;
      if (c < 3) {
        e3_dma_wait();
        __syncthreads();
        buf ^= 1;
        soff += 40 * 1024;
      }
    }
    FD_STAMP(4);
    if (E3_ABL & 1) { if (Y[0][0] == 1234.5f) a.z_out[tc.p] = 1; if (tile + (int)gridDim.x >= n_tiles) return; }
    // ================= tile boundary: every wave is done with both chunk buffers after this barrier; the next tile's
    // operands are requested, THEN the LayerNorm epilogue of this tile runs under their latency
    const int ntile = tile + gridDim.x;
    const bool has_next = ntile < n_tiles;
    __syncthreads();
    if (has_next) e3_request(a, ntile, tid, lane, wave, n_pairs, n_rows, stream, smem, xst, brow_lds + (par ^ 1) * E3_BROWS * E3_BROW_BYTES);
    // ---- epilogue: + Af[i], LayerNorm over the pair's 128 features (32 here, the rest in the 3 other lane groups), mask,
    // bf16, 8-byte stores (features 16t + 4q .. +3 of the pair's 256 B row)
    {
      const float* afl = a1l + E3_H;
      const float* gml = vec + E3_H + 4 * q;
      const float* btl = vec + E3_H + E3_CZ + 4 * q;
      float s1 = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const f32x4 bv = *(const f32x4*)(afl + 16 * t);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          Y[t][r] += bv[r];
          s1 += Y[t][r];
        }
      }
      s1 += __shfl_xor(s1, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      const float mu = s1 * (1.0f / E3_CZ);
      float s2 = 0.f;
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float d = Y[t][r] - mu;
          s2 += d * d;
        }
      s2 += __shfl_xor(s2, 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      const float rstd = 1.0f / sqrtf(s2 * (1.0f / E3_CZ) + 1e-5f);
      e3_u32x4 zB[4];
      half_t* zo = a.z_out + (long)tc.p * E3_CZ + 4 * q;
      float* tr_row = a.trace ? a.trace + (long)tc.p * E3_CZ + 4 * q : nullptr;
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const f32x4 gm = *(const f32x4*)(gml + 16 * t), bt = *(const f32x4*)(btl + 16 * t);
        float of[4];
        e3_hx4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          of[r] = ((Y[t][r] - mu) * rstd * gm[r] + bt[r]) * em;
          o[r] = (fd_h)of[r];
        }
        if (tc.valid) {
          *(e3_hx4*)(zo + 16 * t) = o;
          if (tr_row) {
            f32x4 tv = {of[0], of[1], of[2], of[3]};
            *(f32x4*)(tr_row + 16 * t) = tv;
          }
        }
        const e3_u32x2 ow = __builtin_bit_cast(e3_u32x2, o);  // C/D hand-off of tiles (2s, 2s+1) -> B fragment s of z'
        zB[t >> 1][2 * (t & 1)] = ow[0];
        zB[t >> 1][2 * (t & 1) + 1] = ow[1];
      }
      if (a.wb_img) {
        // pair bias of the next block's attention: D[head, pair] = Wb z' (4 MFMAs), heads 4q + r live in lane groups q < 2
        e3_f32x4 accb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < 4; ++s)
          accb = fd_mfma16(e3_frag((const char*)vec + 2560 + s * 1024 + lane * 16),
                                                         __builtin_bit_cast(hx8, zB[s]), accb);
        if (tc.valid && q < 2) {
          const int b_idx = tc.bi / N, ii = tc.bi - b_idx * N, jj = tc.bj - b_idx * N, nt = (N + 31) >> 5;
          float* bo = a.bias_out + fd_bias_frag_off((long)b_idx * a.H + 4 * q, nt, ii, jj);
          const long hstride = (long)nt * nt * 1024;
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (4 * q + r < a.H) bo[r * hstride] = accb[r] + a.bb[4 * q + r];
        }
      }
    }
    FD_STAMP(5);
    if (!has_next) break;
    // ---- next tile becomes current
    tile = ntile;
    tc = e3_tile(tile, wave, n, N, n_pairs);
    em = a.res_mask[tc.bi] * a.res_mask[tc.bj];
    par ^= 1;
    e3_dma_wait();
    __syncthreads();
  }
}

int fd_edge_transition3_supported(int N) { return N >= 43 && N <= 2048; }  // 4 staged A1/Af rows cover a 128-pair tile

int fd_edge_transition3(const ET2Args& a, hipStream_t st) {
  const long n_pairs = (long)a.B * a.N * a.N;
  if (n_pairs >= (1L << 31) - 256 || !a.e_h16) return FDIPT_EINVAL;  // 32-bit pair indices in the kernel
  const int n_tiles = cdiv(n_pairs, 128);
  static FdPerDevice attr_dev;
  const int dev_ = fd_device();
  if (!attr_dev.get(dev_)) {
    if (hipFuncSetAttribute((const void*)edge_transition3_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, E3_LDS) != hipSuccess)
      return FDIPT_ELAUNCH;
    attr_dev.set(dev_, 1);
  }
  const int n_cu = fd_cu_count();
  // persistent: one block per CU (minus the CUs left to concurrent streams, in whole XCD rounds of 8)
  const int cus = a.reserve_cus > 0 && a.reserve_cus < n_cu - 8 ? (n_cu - a.reserve_cus) & ~7 : n_cu;
  const int grid = n_tiles < cus ? n_tiles : cus;
  hipLaunchKernelGGL(edge_transition3_kernel, dim3(grid), dim3(E3_THREADS), E3_LDS, st, a, n_tiles);
  FD_CHECK_LAUNCH();
  return FDIPT_OK;
}
