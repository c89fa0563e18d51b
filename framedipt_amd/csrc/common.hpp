// common.hpp — shared device helpers for libfdipt_hip (gfx950 / CDNA4 only).
//
// Everything matmul-shaped in the library goes through one MFMA shape, 32x32 per wave:
//   fp32 mode : v_mfma_f32_32x32x2_f32   (exact fp32 FMA chain, 157 TF/s peak)
//   bf16 mode : v_mfma_f32_32x32x16_bf16 (bf16 operands, fp32 accumulate, 2.5 PF/s peak)
// Fragment maps (MI355X guide, section 3):
//   A operand, lane l : row i = l&31, k = (l>>5)*KL .. +KL   (KL = 1 for f32, 8 for bf16)
//   B operand, lane l : col j = l&31, same k range
//   C/D,       lane l : col j = l&31, row i = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16)
// A and B tiles live in LDS k-contiguous ([row][k]), so both fragments are one ds_read per step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/fdipt.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef unsigned short bf16_t;  // raw bf16 bits

#define FD_WAVE 64
#define FD_THREADS 256

__device__ __forceinline__ bf16_t f2bf(float f) {  // round to nearest even
  unsigned u = __float_as_uint(f);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)(u >> 16);
}
__device__ __forceinline__ float bf2f(bf16_t h) { return __uint_as_float(((unsigned)h) << 16); }

__device__ __forceinline__ int c_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ------------------------------------------------------------------ precision traits
struct PrecF32 {
  typedef float T;                    // operand element in LDS / global weights
  static constexpr int BK = 32;       // k-tile staged per step
  static constexpr int PAD = 1;       // LDS row padding (elements): stride 33 words -> conflict-free ds_read_b32
  static constexpr int KL = 1;        // k elements per lane per MFMA
  static constexpr int KS = 2;        // k per MFMA
  static __device__ __forceinline__ T from_f32(float x) { return x; }
  static __device__ __forceinline__ float to_f32(T x) { return x; }
  static __device__ __forceinline__ void mma(f32x16& acc, const T* a_row, const T* b_row, int hi) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_row[hi], b_row[hi], acc, 0, 0, 0);
  }
};
struct PrecBF16 {
  typedef bf16_t T;
  static constexpr int BK = 32;
  static constexpr int PAD = 8;       // stride 40 elem = 80 B: 16-B aligned rows, conflict-free ds_read_b128
  static constexpr int KL = 8;
  static constexpr int KS = 16;
  static __device__ __forceinline__ T from_f32(float x) { return f2bf(x); }
  static __device__ __forceinline__ float to_f32(T x) { return bf2f(x); }
  static __device__ __forceinline__ void mma(f32x16& acc, const T* a_row, const T* b_row, int hi) {
    bf16x8 a = __builtin_bit_cast(bf16x8, *(const u16x8*)(a_row + hi * 8));
    bf16x8 b = __builtin_bit_cast(bf16x8, *(const u16x8*)(b_row + hi * 8));
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};

// One wave: acc(32x32) += A[32 x kt] * B[32 x kt]^T with both tiles k-contiguous in LDS.
// a_row / b_row point at this lane's row (row = base + (lane&31)) at k = 0 of a full BK tile (zero-filled past K).
template <class P>
__device__ __forceinline__ void wave_mma(f32x16& acc, const typename P::T* a_row, const typename P::T* b_row,
                                         int lane) {
  const int hi = lane >> 5;
#pragma unroll
  for (int k = 0; k < P::BK; k += P::KS) P::mma(acc, a_row + k, b_row + k, hi);
}

// ------------------------------------------------------------------ staging: global -> LDS tile [rows][BK+PAD]
// src element (r, k) = src[(row0 + r) * ld + k0 + k]; rows >= nrows and k >= K are zero-filled.
// SrcT = float (activations / fp32 weights) or bf16_t (prepared bf16 weights). ld % 4 == 0 (float) / % 8 == 0 (bf16).
template <class P, class SrcT, int ROWS>
__device__ __forceinline__ void stage_tile(typename P::T* dst, const SrcT* __restrict__ src, long ld, int row0,
                                           int nrows, int k0, int K, int tid) {
  constexpr int LDT = P::BK + P::PAD;
  if constexpr (sizeof(SrcT) == 4) {
    constexpr int VPR = P::BK / 4;  // float4 per row
#pragma unroll
    for (int v = tid; v < ROWS * VPR; v += FD_THREADS) {
      const int r = v / VPR, kk = (v % VPR) * 4;
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (row0 + r < nrows && k0 + kk < K) x = *(const f32x4*)(src + (long)(row0 + r) * ld + k0 + kk);
      typename P::T* d = dst + r * LDT + kk;
      if constexpr (sizeof(typename P::T) == 4) {
        d[0] = x[0]; d[1] = x[1]; d[2] = x[2]; d[3] = x[3];
      } else {
        u16x4 h = {f2bf(x[0]), f2bf(x[1]), f2bf(x[2]), f2bf(x[3])};
        *(u16x4*)d = h;
      }
    }
  } else {
    static_assert(sizeof(typename P::T) == 2, "bf16 source needs bf16 operands");
    constexpr int VPR = P::BK / 8;  // 16-byte vectors per row
#pragma unroll
    for (int v = tid; v < ROWS * VPR; v += FD_THREADS) {
      const int r = v / VPR, kk = (v % VPR) * 8;
      u16x8 x = {0, 0, 0, 0, 0, 0, 0, 0};
      if (row0 + r < nrows && k0 + kk < K) x = *(const u16x8*)(src + (long)(row0 + r) * ld + k0 + kk);
      *(u16x8*)(dst + r * LDT + kk) = x;
    }
  }
}

// ------------------------------------------------------------------ reductions
// L2 warm-up hand-over.  Every kernel of the node path starts with dependent reads of weights that no XCD's L2 holds any more
// (each weight set is used once per forward, 2 ms and ~1 GB of pair traffic ago).  A kernel can touch the NEXT kernel's weights
// (one dword per 128 B line, result discarded) at a point where it has nothing left to wait for, so that those first reads hit.
// Blocks are dispatched round-robin over the 8 XCDs: block `id` runs on XCD id & 7 and takes slice id >> 3 of ceil(n_blocks / 8)
// slices, i.e. every XCD's L2 sees every line.  The returned token keeps the destination register reserved: hand it to
// fd_l2_warm_done() at the end of the kernel (the loads are invisible to the compiler's vmcnt bookkeeping).
struct L2Warm { const void* p[3]; unsigned bytes[3]; };
__device__ __forceinline__ unsigned fd_l2_warm(const L2Warm& w, int block_id, int n_blocks, int t, int nthreads) {
  unsigned tok = 0;
  const unsigned slice = (unsigned)block_id >> 3, ns = ((unsigned)n_blocks + 7) >> 3;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (!w.p[k]) continue;
    const unsigned lines = (w.bytes[k] + 127) >> 7, per = (lines + ns - 1) / ns;
    const unsigned l0 = slice * per, l1 = l0 + per < lines ? l0 + per : lines;
    for (unsigned l = l0 + t; l < l1; l += nthreads)
      asm volatile("global_load_dword %0, %1, off" : "+v"(tok) : "v"((const char*)w.p[k] + ((size_t)l << 7)) : "memory");
  }
  return tok;
}
// ... by the LAST `last` blocks of a launch only (kernels that stream more than an L2's worth of data after their first
// blocks would evict lines touched earlier): `id` is the linear block index in dispatch order
__device__ __forceinline__ unsigned fd_l2_warm_last(const L2Warm& w, int id, int n_blocks, int last, int t, int nthreads) {
  const int base = n_blocks > last ? (n_blocks - last + 7) & ~7 : 0;  // multiple of 8: (id - base) & 7 is still the XCD
  return id >= base ? fd_l2_warm(w, id - base, n_blocks - base, t, nthreads) : 0u;
}
__device__ __forceinline__ void fd_l2_warm_done(unsigned tok) { asm volatile("s_waitcnt vmcnt(0)" : : "v"(tok) : "memory"); }

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

#define FD_CHECK_LAUNCH()                                   \
  do {                                                      \
    if (hipGetLastError() != hipSuccess) return FDIPT_ELAUNCH; \
  } while (0)

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// Optional in-kernel phase timestamps for the stand-alone harnesses under tools/micro (-DFD_PROF): thread 0 of every
// block records s_memtime (shader cycles) at phase boundaries into fd_prof[block][16].  Never enabled in the library.
#ifdef FD_PROF
__device__ unsigned long long fd_prof[8192 * 16];
#define FD_STAMP(k)                                                                                              \
  do {                                                                                                           \
    if (threadIdx.x == 0)                                                                                        \
      fd_prof[((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 8191) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define FD_STAMP(k) \
  do {              \
  } while (0)
#endif

