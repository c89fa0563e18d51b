// common.hpp — shared device helpers for libfdipt_hip (gfx950 / CDNA4 only).
//
// Everything matmul-shaped in the library goes through one MFMA shape, 32x32 per wave:
//   fp32 mode : v_mfma_f32_32x32x2_f32   (exact fp32 FMA chain, 157 TF/s peak)
//   half mode : v_mfma_f32_32x32x16_f16  (fp16 operands - bf16 in a -DFDIPT_HALF_BF16 build - fp32 accumulate, 2.5 PF/s peak)
// Fragment maps (MI355X guide, section 3):
//   A operand, lane l : row i = l&31, k = (l>>5)*KL .. +KL   (KL = 1 for f32, 8 for half)
//   B operand, lane l : col j = l&31, same k range
//   C/D,       lane l : col j = l&31, row i = (r&3) + 8*(r>>2) + 4*(l>>5), r in [0,16)
// A and B tiles live in LDS k-contiguous ([row][k]), so both fragments are one ds_read per step.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>

#include "../../include/fdipt.h"

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) unsigned short u16x8;
typedef __attribute__((ext_vector_type(4))) unsigned short u16x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;

// ------------------------------------------------------------------ half-precision operand type of the throughput mode
// One 16-bit operand type per build.  Default: IEEE fp16 (v_mfma_f32_32x32x16_f16 runs at the bf16 rate with three more
// significant bits; every half-precision operand of this network is a post-LayerNorm / post-ReLU activation, a softmax
// weight or a weight matrix, all far inside the fp16 range).  -DFDIPT_HALF_BF16 builds the bf16 variant for comparison.
// `half_t` is the raw bit pattern (what buffers and LDS hold), `fd_h` the arithmetic type, `hx8` one MFMA fragment.
typedef unsigned short half_t;
#ifdef FDIPT_HALF_BF16
typedef __bf16 fd_h;
#define FDIPT_PREC_HALF FDIPT_PREC_BF16
#define FD_H_ONE_BITS 0x3F80u   // 1.0
#define FD_H_NEG_BIG (-1e30f)   // "minus infinity" that stays finite in the operand type (key-padding channel)
#else
typedef _Float16 fd_h;
#define FDIPT_PREC_HALF FDIPT_PREC_F16
#define FD_H_ONE_BITS 0x3C00u
#define FD_H_NEG_BIG (-60000.f)
#endif
typedef __attribute__((ext_vector_type(8))) fd_h hx8;
typedef __attribute__((ext_vector_type(4))) fd_h hx4;
typedef __attribute__((ext_vector_type(2))) fd_h hx2;
__device__ __forceinline__ f32x16 fd_mfma32(hx8 a, hx8 b, f32x16 c) {  // 32x32x16, 8 k-elements per lane
#ifdef FDIPT_HALF_BF16
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}
__device__ __forceinline__ f32x4 fd_mfma16(hx8 a, hx8 b, f32x4 c) {  // 16x16x32
#ifdef FDIPT_HALF_BF16
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

// IEEE fp16 fragments in EVERY build: the point logits of attention3 run on fp16 hi / lo parts (22 significant bits) also in the
// bf16 build — bf16 hi / lo would carry 16
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
__device__ __forceinline__ f32x16 fd_mfma32_f16(f16x8 a, f16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
__device__ __forceinline__ half_t f2f16(float f) { return __builtin_bit_cast(half_t, (_Float16)f); }
__device__ __forceinline__ float f162f(half_t h) { return (float)__builtin_bit_cast(_Float16, h); }

#define FD_WAVE 64
#define FD_THREADS 256

// round to nearest even (v_cvt_f16_f32 / the bf16 conversion of gfx950)
__device__ __forceinline__ half_t f2h(float f) { return __builtin_bit_cast(half_t, (fd_h)f); }
__device__ __forceinline__ float h2f(half_t h) { return (float)__builtin_bit_cast(fd_h, h); }
// two fp32 -> one word of two half values: a 2-vector conversion, which hipcc selects as ONE packed conversion
// (element-wise conversions + bit casts become two conversions and a v_perm_b32)
__device__ __forceinline__ unsigned fd_cvt_pk(float lo, float hi) {
  return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, hx2));
}

__device__ __forceinline__ int c_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// ------------------------------------------------------------------ precision traits
struct PrecF32 {
  typedef float T;                    // operand element in LDS / global weights
  static constexpr int BK = 32;       // k-tile staged per step
  static constexpr int PAD = 1;       // LDS row padding (elements): stride 33 words -> conflict-free ds_read_b32
  static constexpr int KL = 1;        // k elements per lane per MFMA
  static constexpr int KS = 2;        // k per MFMA
  static constexpr int LDMUL = 1;
  static __device__ __forceinline__ T from_f32(float x) { return x; }
  static __device__ __forceinline__ float to_f32(T x) { return x; }
  static __device__ __forceinline__ void mma(f32x16& acc, const T* a_row, const T* b_row, int hi) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a_row[hi], b_row[hi], acc, 0, 0, 0);
  }
};
struct PrecHalf {
  typedef half_t T;
  static constexpr int BK = 32;
  static constexpr int PAD = 8;       // stride 40 elem = 80 B: 16-B aligned rows, conflict-free ds_read_b128
  static constexpr int KL = 8;
  static constexpr int KS = 16;
  static constexpr int LDMUL = 1;
  static __device__ __forceinline__ T from_f32(float x) { return f2h(x); }
  static __device__ __forceinline__ float to_f32(T x) { return h2f(x); }
  static __device__ __forceinline__ void mma(f32x16& acc, const T* a_row, const T* b_row, int hi) {
    hx8 a = __builtin_bit_cast(hx8, *(const u16x8*)(a_row + hi * 8));
    hx8 b = __builtin_bit_cast(hx8, *(const u16x8*)(b_row + hi * 8));
    acc = fd_mfma32(a, b, acc);
  }
};

// split operands for the tiled GEMM of gemm.hip (64-wide k-tiles): an LDS row holds the 64 hi parts, then the 64 lo parts of
// its k-tile (x = hi + lo, each a half-precision value; 22 significant bits together); a k-step is hi.hi + hi.lo + lo.hi
struct PrecSplit {
  typedef half_t T;
  static constexpr int PAD = 8;
  static constexpr int KL = 8;
  static constexpr int KS = 16;
  static constexpr int LDMUL = 2;     // row length in k-tiles
  static __device__ __forceinline__ void mma(f32x16& acc, const T* a_row, const T* b_row, int hi) {
    const hx8 ah = __builtin_bit_cast(hx8, *(const u16x8*)(a_row + hi * 8)), al = __builtin_bit_cast(hx8, *(const u16x8*)(a_row + 64 + hi * 8));
    const hx8 bh = __builtin_bit_cast(hx8, *(const u16x8*)(b_row + hi * 8)), bl = __builtin_bit_cast(hx8, *(const u16x8*)(b_row + 64 + hi * 8));
    acc = fd_mfma32(ah, bh, acc);
    acc = fd_mfma32(ah, bl, acc);
    acc = fd_mfma32(al, bh, acc);
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
// SrcT = float (activations / fp32 weights) or half_t (prepared bf16 weights). ld % 4 == 0 (float) / % 8 == 0 (bf16).
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
        u16x4 h = {f2h(x[0]), f2h(x[1]), f2h(x[2]), f2h(x[3])};
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

// Register-staged halves of stage_tile / stage_tile_T for fp32 sources: the global loads of k-tile t + 1 are issued before the matrix
// instructions of k-tile t and stored to LDS after them (round 3: the unpipelined load -> LDS -> barrier -> MFMA -> barrier loop exposed
// a full memory round trip per k-tile, 128 of them per block at N = 1000: 9x the matrix time).
template <int ROWS, int BK>
struct StageRegs { f32x4 v[(ROWS * (BK / 4) + FD_THREADS - 1) / FD_THREADS]; };
template <class P, int ROWS>
__device__ __forceinline__ void stage_load(StageRegs<ROWS, P::BK>& R, const float* __restrict__ src, long ld, int row0, int nrows, int k0, int K, int tid) {
  constexpr int VPR = P::BK / 4, NV = (ROWS * VPR + FD_THREADS - 1) / FD_THREADS;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int v = tid + u * FD_THREADS, r = v / VPR, kk = (v % VPR) * 4;
    R.v[u] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (v < ROWS * VPR && row0 + r < nrows && k0 + kk < K) R.v[u] = *(const f32x4*)(src + (long)(row0 + r) * ld + k0 + kk);
  }
}
template <class P, int ROWS>
__device__ __forceinline__ void stage_store(typename P::T* dst, const StageRegs<ROWS, P::BK>& R, int tid) {
  constexpr int LDT = P::BK + P::PAD, VPR = P::BK / 4, NV = (ROWS * VPR + FD_THREADS - 1) / FD_THREADS;
#pragma unroll
  for (int u = 0; u < NV; ++u) {
    const int v = tid + u * FD_THREADS, r = v / VPR, kk = (v % VPR) * 4;
    if (v < ROWS * VPR) {
      typename P::T* d = dst + r * LDT + kk;
      if constexpr (sizeof(typename P::T) == 4) {
        d[0] = R.v[u][0]; d[1] = R.v[u][1]; d[2] = R.v[u][2]; d[3] = R.v[u][3];
      } else {
        const u16x4 h = {f2h(R.v[u][0]), f2h(R.v[u][1]), f2h(R.v[u][2]), f2h(R.v[u][3])};
        *(u16x4*)d = h;
      }
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

// Opt-in in-kernel shader-clock probe of the EdgeTransition kernels (FdiptForwardArgs.clock_out): thread 0 of a block adds its
// core-clock cycles (s_memtime) and 100 MHz ticks (s_memrealtime) from start to end to `ctr[0..1]` and counts the block in ctr[2];
// ctr == NULL (every product launch): no atomics.
#define FD_CLK_BEGIN const unsigned long long clk0_ = __builtin_amdgcn_s_memtime(), rt0_ = __builtin_amdgcn_s_memrealtime()
#define FD_CLK_END(ctr)                                                                       \
  if ((ctr) != nullptr && threadIdx.x == 0) {                                                                     \
    atomicAdd(&(ctr)[0], (unsigned long long)__builtin_amdgcn_s_memtime() - clk0_);           \
    atomicAdd(&(ctr)[1], (unsigned long long)__builtin_amdgcn_s_memrealtime() - rt0_);        \
    atomicAdd(&(ctr)[2], 1ull);                                                               \
  }

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
// Runs of three floats (points, translations, atom positions) are stored dword by dword.  hipcc merges adjacent stores into
// global_store_dwordx3; the data registers of that store are read LATE when the memory pipeline is contended by another kernel on
// the CU, and a VALU write of the first data register a few instructions after the store (the one wait state the ISA manual asks
// for is kept) then reaches memory instead of the stored value, 16 lanes at a time: points16_kernel produced wrong x coordinates for
// 16 points of a residue in ~10 % of the forwards that ran next to another forward's attention kernels (DESIGN.md section 5;
// tools/check_store_hazard.py lists the remaining wide stores with early overwrites).
// (inline asm rather than a volatile store: hipcc follows every volatile access with s_waitcnt vmcnt(0) - reverse_step_kernel went from
// 20 to 71 us - while it cannot merge asm statements; a store the compiler's counter model does not see only makes its later vmcnt
// waits conservative, the counter being in order)
__device__ __forceinline__ void fd_st(float* p, float v) { asm volatile("global_store_dword %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void fd_st(double* p, double v) { asm volatile("global_store_dwordx2 %0, %1, off" : : "v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void fd_store3(float* p, float x, float y, float z) { fd_st(p, x); fd_st(p + 1, y); fd_st(p + 2, z); }

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

// Launch-site state that belongs to a DEVICE, not to the process (the dynamic-LDS cap of a kernel, the CU count): one slot per
// device id, filled on the first launch from that device — a process may drive several GPUs, and callers on different host threads
// / streams may race here (relaxed atomics: setting an attribute twice is harmless).
#define FD_MAX_DEVICES 64
struct FdPerDevice {
  std::atomic<int> v[FD_MAX_DEVICES];
  int get(int dev) const { return dev >= 0 ? v[dev].load(std::memory_order_relaxed) : 0; }
  void set(int dev, int x) { if (dev >= 0) v[dev].store(x, std::memory_order_relaxed); }
};
static inline int fd_device() {
  int d = 0;
  return hipGetDevice(&d) == hipSuccess && d >= 0 && d < FD_MAX_DEVICES ? d : -1;
}
// CU count of the current device (256 on MI355X), cached per device
static inline int fd_cu_count() {
  static FdPerDevice cache;
  const int dev = fd_device();
  int n = cache.get(dev);
  if (!n) {
    hipDeviceProp_t prop;
    n = (dev >= 0 && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    cache.set(dev, n);
  }
  return n;
}

// Optional in-kernel phase timestamps for the stand-alone harnesses under tools/micro (-DFD_PROF): thread 0 of every
// block records s_memtime (shader cycles) at phase boundaries into fd_prof[block][16].  Never enabled in the library.
#ifdef FD_PROF
__device__ unsigned long long fd_prof[8192 * 16];
#define FD_STAMP(k)                                                                                              \
  do {                                                                                                           \
    if (threadIdx.x == 0) {                                                                                      \
      fd_prof[((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 8191) * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
      if ((k) < 8) /* slots 8 .. 15: the same instants on the chip-wide 100 MHz clock (s_memtime counters differ between XCDs) */    \
        fd_prof[((blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) & 8191) * 16 + 8 + (k)] = __builtin_amdgcn_s_memrealtime(); \
    }                                                                                                            \
  } while (0)
#else
#define FD_STAMP(k) \
  do {              \
  } while (0)
#endif

