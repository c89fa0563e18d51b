// Stand-alone reproducer for the concurrency corruption of rounds 2 - 4 — NO library code.
//   hipcc --offload-arch=gfx950 -O3 -w hazard_repro.hip -o hazard_repro && ./hazard_repro [seconds_per_cell=4] [only]
//
// FINDING (round 4, profiles/r04_hazard_*.txt).  Two kernels on two HIP streams whose waves share SIMDs:
//   victim     a 16-lanes-per-item series kernel shaped like the IGSO(3) rotation score (fp64 or integer-only), every item fed the same
//              input, every result compared on the device with a quiet launch;
//   aggressor  exec_aggressor_kernel: a half-precision MFMA (v_mfma_f32_32x32x16_f16 or 16x16x32_f16) followed after GAP wait states by
//              `s_and_saveexec_b64 (lanes 0-7, 32-39); v_add_u32; s_or_b64 exec` — what hipcc emits for `if (lane % 32 < 8) x = lds[..]`
//              next to an MFMA.
// The victim then computes WRONG VALUES in lanes 48 - 63 of a wave (item % 4 == 3): 0.3 % of its launches at GAP 0, 30 % at GAP 2, practically
// all of them at GAP >= 16; none with the EXEC sequence alone, none with the MFMA alone (but for the kernel's own epilogue branch), none
// with the fp32 MFMA (v_mfma_f32_32x32x2_f32) at any gap; 100x fewer when the mask keeps lanes 48 - 63 enabled.  The wrong value is
// deterministic (a stale register: one VALU result of the victim's first instructions is not written in its last 16-lane pass): the
// victim's write enables of that pass follow the OTHER wave's EXEC[63:48].  In-register double evaluations never see it (both copies
// are wrong alike or the event falls on launch-time state), memory and LDS traffic do not matter, AGPR operands do not matter
// (chain_aggressor_kernel: the few failures there come from the epilogue branch behind the last MFMA).
// The library's failures (a wrong rotation score of one residue next to another stream's forward) are this: bisected with
// tools/hazard_lib_repro.py to the attention's logit MFMAs and the o_pair kernel, both with lane-masked loads between MFMAs.
// It needs waves of different kernels on one SIMD: a single stream never co-schedules two kernels, so the product path is not exposed.
//
// Older sections below (kept: they are the negative results): a persistent MFMA power kernel that does NOT share CUs with the victim,
// memory streamers, instruction-class victims with in-register checks.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <chrono>
typedef _Float16 hx8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

#define RS_L 1000
__device__ __forceinline__ int rs_cut(double sg) { const double c = sqrt(1520.0) / sg + 2.0; return c < (double)RS_L ? (int)c : RS_L; }

// victim, float64 series: result[item] (3 doubles) ; INTONLY: an integer hash chain with the same shuffle pattern instead
template <int THREADS, int INTONLY>
__global__ __launch_bounds__(THREADS) void victim_kernel(int n_items, const float* __restrict__ q, double sigma, double* __restrict__ res,
                                                         const double* __restrict__ expect, unsigned* __restrict__ bad) {
  __shared__ double wtab[RS_L];
  const int cut = rs_cut(sigma);
  if (!INTONLY) {
    for (int v = threadIdx.x; v < cut; v += THREADS) wtab[v] = (double)(2 * v + 1) * exp(-(double)v * (double)(v + 1) * sigma * sigma / 2);
    __syncthreads();
  }
  const int item = blockIdx.x * (THREADS / 16) + threadIdx.x / 16, sub = threadIdx.x % 16;
  const int it = item < n_items ? item : n_items - 1;
  // every item reads the same 8 floats (two unit quaternions): rotation vector of q0^-1 * qt
  const float a0 = q[0], a1 = -q[1], a2 = -q[2], a3 = -q[3], b0 = q[4], b1 = q[5], b2 = q[6], b3 = q[7];
  float w = a0 * b0 - a1 * b1 - a2 * b2 - a3 * b3, x = a0 * b1 + a1 * b0 + a2 * b3 - a3 * b2, y = a0 * b2 - a1 * b3 + a2 * b0 + a3 * b1,
        z = a0 * b3 + a1 * b2 - a2 * b1 + a3 * b0;
  if (w < 0.f) { w = -w; x = -x; y = -y; z = -z; }
  const float nv = sqrtf(x * x + y * y + z * z);
  const float ang = 2.f * atan2f(nv, w);
  const float sc = ang / sinf(ang / 2.f + 1e-6f);
  const float rv[3] = {x * sc, y * sc, z * sc};
  const float omega = sqrtf(rv[0] * rv[0] + rv[1] * rv[1] + rv[2] * rv[2]) + 1e-6f;
  double out[3];
  unsigned long long pre = 0;  // this lane's contribution before the 16-lane butterfly (recorded on a mismatch)
  unsigned trips = 0;
  if (INTONLY) {
    unsigned long long hsh = 0x9E3779B97F4A7C15ull ^ (unsigned long long)__float_as_uint(omega);
    for (int l = sub; l < cut; l += 16) {
      hsh ^= (unsigned long long)l * 0xD6E8FEB86659FD93ull;
      hsh = (hsh << 13) | (hsh >> 51);
      hsh *= 0xFF51AFD7ED558CCDull;
      ++trips;
    }
    pre = hsh;
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) hsh += __shfl_xor(hsh, o, 64);
    out[0] = out[1] = out[2] = __longlong_as_double((long long)((hsh & 0x000FFFFFFFFFFFFFull) | 0x3FF0000000000000ull));
  } else {
    const float lo = sinf(omega / 2.f), dlo = 0.5f * cosf(omega / 2.f);
    const float den = lo * lo;
    double f = 0, ds = 0;
    for (int l = sub; l < cut; l += 16) {
      ++trips;
      const double wv = wtab[l];
      const float lh = (float)l + 0.5f;
      const float arg = omega * lh;
      const float hi = sinf(arg), dhi = lh * cosf(arg);
      f += wv * (double)hi / (double)lo;
      const float num = lo * dhi - hi * dlo;
      ds += wv * (double)num / (double)den;
    }
    pre = (unsigned long long)__double_as_longlong(f);
#pragma unroll
    for (int o = 1; o < 16; o <<= 1) {
      f += __shfl_xor(f, o, 64);
      ds += __shfl_xor(ds, o, 64);
    }
    const double s = ds / (f + 1e-4);
    for (int c = 0; c < 3; ++c) out[c] = s * (double)rv[c] / (double)omega;
  }
  if (item >= n_items) return;
  {
    // per-lane record of an item's 16 lanes: [pre-butterfly contribution | omega bits, trip count | item, lane of the block, sub, hardware id].
    // Slot 7 = item 7 of a quiet reference launch (expect == NULL); slots 0..3 = the first mismatching items
    const bool mism = expect && __double_as_longlong(out[0]) != __double_as_longlong(expect[0]);
    const bool refdump = !expect && item == 7;
    if (mism || refdump) {
      unsigned slot = 7;
      if (mism) {
        if (sub == 0) slot = atomicAdd(&bad[7], 1u);
        slot = __shfl(slot, (threadIdx.x & 63) & ~15, 64);
      }
      if (slot < 4 || refdump) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned long long* r = (unsigned long long*)(bad + 256) + ((size_t)slot * 16 + sub) * 3;
        r[0] = pre;
        r[1] = (unsigned long long)__float_as_uint(omega) | ((unsigned long long)trips << 32);
        r[2] = (unsigned long long)item | ((unsigned long long)threadIdx.x << 24) | ((unsigned long long)hw << 34);
      }
    }
  }
  if (sub < 3) {
    res[it * 3 + sub] = out[sub];
    const long long first = expect ? __double_as_longlong(expect[sub]) : 0;
    if (expect && __double_as_longlong(out[sub]) != first) {
      atomicAdd(&bad[0], 1u);
      atomicAdd(&bad[1 + (item & 3)], 1u);
      if (atomicAdd(&bad[5], 1u) < 8) { /* keep the first few for the report: the value, the expectation as first loaded, and re-loaded at L2 */
        const unsigned slot = atomicAdd(&bad[6], 1u);
        if (slot < 8) {
          bad[8 + 4 * slot] = item; bad[9 + 4 * slot] = sub;
          long long* rec = (long long*)(bad + 64) + 3 * slot;
          rec[0] = __double_as_longlong(out[sub]);
          rec[1] = first;
          rec[2] = (long long)__hip_atomic_load((const unsigned long long*)expect + sub, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
  }
}

// aggressor: persistent MFMA loop, one block per CU (152 KB of LDS), optional global read stream beside the matrix work
template <int MEM>
__global__ __launch_bounds__(512, 1) void aggressor_kernel(const u32x4* __restrict__ ops, float* __restrict__ out, int iters, const u32x4* __restrict__ gsrc,
                                                           size_t gwords) {
  extern __shared__ u32x4 lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  hx8 A[4], B[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    A[i] = __builtin_bit_cast(hx8, ops[((blockIdx.x * 8 + wave) % 64 * 8 + i) * 64 + lane]);
    B[i] = __builtin_bit_cast(hx8, ops[(((blockIdx.x * 8 + wave) % 64) * 8 + 4 + i) * 64 + lane]);
  }
  lds[tid] = __builtin_bit_cast(u32x4, A[0]);
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  u32x4 sink = {0, 0, 0, 0};
  size_t goff = ((size_t)blockIdx.x * 512 + tid) % gwords;
  for (int it = 0; it < iters; ++it) {
    u32x4 g = {0, 0, 0, 0};
    if (MEM) { g = gsrc[goff]; goff += 512 * 256; if (goff >= gwords) goff -= gwords; }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[u & 3], B[(u >> 2) & 3], acc[u & 3], 0, 0, 0);
    if (MEM) sink += g;
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 1.2345e-30f || sink[0] == 0x1234567u) out[tid] = s + lds[tid][0];
}

static void fill_ops(u32x4* d_ops, int pattern) {
  const size_t n = (size_t)64 * 8 * 64 * 8;
  std::vector<unsigned short> h(n);
  unsigned st = 777u;
  for (size_t i = 0; i < n; ++i) {
    st = st * 1664525u + 1013904223u;
    unsigned short bits = pattern ? (unsigned short)(st >> 16) : 0;
    if ((bits & 0x7c00) == 0x7c00) bits &= ~0x0400;
    h[i] = bits;
  }
  hipMemcpy(d_ops, h.data(), n * 2, hipMemcpyHostToDevice);
}

template <int THREADS, int INTONLY>
static void launch_victim(hipStream_t s, int n_items, const float* q, double sigma, double* res, const double* expect, unsigned* bad) {
  const int per = THREADS / 16;
  hipLaunchKernelGGL((victim_kernel<THREADS, INTONLY>), dim3((n_items + per - 1) / per), dim3(THREADS), 0, s, n_items, q, sigma, res, expect, bad);
}

template <int THREADS, int INTONLY>
static void cell(const char* name, int pattern, int mem, int reserve, double seconds, double sigma, u32x4* d_ops, u32x4* d_g, size_t gwords, float* d_out,
                 const float* d_q, double* d_res, double* d_expect, unsigned* d_bad, hipStream_t sa, hipStream_t sb) {
  const int n_items = 2896;  // N = 724, 4 samples: the size at which the library's soak failed
  // expectation from a quiet launch (no aggressor)
  hipDeviceSynchronize();
  launch_victim<THREADS, INTONLY>(sa, n_items, d_q, sigma, d_res, nullptr, d_bad);
  hipStreamSynchronize(sa);
  hipMemcpy(d_expect, d_res, 24, hipMemcpyDeviceToDevice);
  hipMemset(d_bad, 0, 8192);
  // quiet self-check: every item of the quiet launch equals item 0
  launch_victim<THREADS, INTONLY>(sa, n_items, d_q, sigma, d_res, d_expect, d_bad);
  hipStreamSynchronize(sa);
  unsigned hb[64];
  hipMemcpy(hb, d_bad, 256, hipMemcpyDeviceToHost);
  const unsigned quiet_bad = hb[0];
  hipMemset(d_bad, 0, 8192);
  if (pattern >= 0) fill_ops(d_ops, pattern);
  const int nblk = 256 - reserve;
  const int iters = 450;  // ~300 us per aggressor launch at the ~1.55 GHz random operands sustain (1024 cycles per iteration and SIMD)
  hipFuncSetAttribute((const void*)aggressor_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  hipFuncSetAttribute((const void*)aggressor_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  long launches = 0;
  const auto t0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    // keep both queues fed: 8 aggressor launches (~2.4 ms), victim launches beside them
    if (pattern >= 0)
      for (int i = 0; i < 8; ++i) {
        if (mem) hipLaunchKernelGGL((aggressor_kernel<1>), dim3(nblk), dim3(512), 152 * 1024, sb, d_ops, d_out, iters, d_g, gwords);
        else hipLaunchKernelGGL((aggressor_kernel<0>), dim3(nblk), dim3(512), 152 * 1024, sb, d_ops, d_out, iters, d_g, gwords);
      }
    for (int i = 0; i < 64; ++i) launch_victim<THREADS, INTONLY>(sa, n_items, d_q, sigma, d_res, d_expect, d_bad);
    launches += 64;
    hipStreamSynchronize(sa);
    hipStreamSynchronize(sb);
  }
  hipDeviceSynchronize();
  hipMemcpy(hb, d_bad, 256, hipMemcpyDeviceToHost);
  printf("%-64s reserve %3d  victim launches %6ld  items %9ld  quiet-bad %u  BAD %u  [item%%4: %u %u %u %u]", name, reserve, launches, launches * n_items,
         quiet_bad, hb[0], hb[1], hb[2], hb[3], hb[4]);
  for (unsigned s = 0; s < hb[6] && s < 3; ++s) printf("  (item %u comp %u)", hb[8 + 4 * s], hb[9 + 4 * s]);
  printf("\n");
  fflush(stdout);
}

// ---- victims that isolate one instruction class each (every check is IN-REGISTER: two evaluations of the same function, or a loaded
// value against the function of its address; mismatches are counted per 16-lane quarter of the wave in bad[1 + lane / 16])
//   KIND 0: VALU only (integer hash chain twice, compared)            KIND 1: + a 16-lane ds_bpermute butterfly on both chains
//   KIND 2: global loads of a pattern (value = hash(index)) checked against the recomputed hash
//   KIND 3: LDS round trips (ds_write_b32 / ds_read_b32 of hash values, wave-private rows) checked against the registers
//   KIND 7: the work-item id in v0 (written by the hardware at wave launch) against the EXEC-derived lane index
//   KIND 4: transcendental unit (v_exp / v_rcp / v_sqrt / v_sin)    KIND 5: library float math (division, atan2f, sinf)    KIND 6: float64 (fma, division, sqrt)
template <int KIND>
__global__ __launch_bounds__(256) void victim2_kernel(int rounds, const unsigned* __restrict__ pattern, unsigned pat_mask, unsigned* __restrict__ bad) {
  __shared__ unsigned lrow[256 * 4];
  const int tid = threadIdx.x, lane = tid & 63;
  unsigned a = 0x9E3779B9u ^ (unsigned)(blockIdx.x * 256 + tid), b = a;
  unsigned nbad = 0;
  for (int r = 0; r < rounds; ++r) {
    // two copies of the same chain; the opaque asm keeps the compiler from merging them
    a = (a ^ (unsigned)r) * 0x85EBCA6Bu; a = (a << 13) | (a >> 19); a *= 0xC2B2AE35u;
    asm volatile("" : "+v"(b));
    b = (b ^ (unsigned)r) * 0x85EBCA6Bu; b = (b << 13) | (b >> 19); b *= 0xC2B2AE35u;
    if (KIND == 1) {
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { a += __shfl_xor(a, o, 64); asm volatile("" : "+v"(b)); b += __shfl_xor(b, o, 64); }
    }
    if (KIND == 2) {
      const unsigned idx = a & pat_mask;
      const unsigned v = pattern[idx];
      const unsigned want = (idx * 0x9E3779B1u) ^ 0x5bd1e995u;
      nbad += v != want;
    }
    if (KIND == 3) {
      lrow[tid * 4 + (r & 3)] = a;
      asm volatile("" ::: "memory");
      const unsigned back = lrow[tid * 4 + (r & 3)];
      nbad += back != a;
    }
    if (KIND == 4) {  // transcendental unit: v_exp_f32 / v_rcp_f32 / v_sqrt_f32 / v_sin_f32 on a value derived from the chain, twice
      const float x = 0.5f + (float)(a & 0xFFFF) * (1.0f / 65536.0f), y = 0.5f + (float)(b & 0xFFFF) * (1.0f / 65536.0f);
      const float fx = __builtin_amdgcn_sinf(__builtin_amdgcn_sqrtf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x))));
      asm volatile("" : "+v"(b));
      const float fy = __builtin_amdgcn_sinf(__builtin_amdgcn_sqrtf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(y))));
      a ^= __float_as_uint(fx); b ^= __float_as_uint(fy);
    }
    if (KIND == 5) {  // library float math: division (v_div_scale / v_div_fmas with the denormal mode switched around it), atan2f, sinf
      const float x = 0.5f + (float)(a & 0xFFFF) * (1.0f / 65536.0f), y = 0.5f + (float)(b & 0xFFFF) * (1.0f / 65536.0f);
      const float fx = atan2f(x, 1.25f - x) / sinf(x * 977.f) + sqrtf(x);
      asm volatile("" : "+v"(b));
      const float fy = atan2f(y, 1.25f - y) / sinf(y * 977.f) + sqrtf(y);
      a ^= __float_as_uint(fx); b ^= __float_as_uint(fy);
    }
    if (KIND == 6) {  // float64: fma, division, sqrt
      const double x = 0.5 + (double)(a & 0xFFFF) * (1.0 / 65536.0), y = 0.5 + (double)(b & 0xFFFF) * (1.0 / 65536.0);
      const double fx = fma(x, 1.000000119, 0.25) / (x * x + 0.75) + sqrt(x);
      asm volatile("" : "+v"(b));
      const double fy = fma(y, 1.000000119, 0.25) / (y * y + 0.75) + sqrt(y);
      a ^= (unsigned)__double_as_longlong(fx) ^ (unsigned)(__double_as_longlong(fx) >> 32);
      b ^= (unsigned)__double_as_longlong(fy) ^ (unsigned)(__double_as_longlong(fy) >> 32);
    }
    nbad += a != b;
  }
  if (KIND == 7) {
    // launch-time state: the work-item id the hardware wrote into v0 before the wave started, against the lane index derived from EXEC
    // (v_mbcnt: independent of v0).  bad[8 ...]: the first few (expected low 6 bits, v0) pairs
    const unsigned mb = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    if ((threadIdx.x & 63u) != mb || threadIdx.x >= 256u) {
      nbad += 1;
      const unsigned slot = atomicAdd(&bad[6], 1u);
      if (slot < 16) { bad[8 + 2 * slot] = mb | (blockIdx.x << 8); bad[9 + 2 * slot] = threadIdx.x; }
    }
  }
  if (nbad) { atomicAdd(&bad[0], nbad); atomicAdd(&bad[1 + (lane >> 4)], nbad); }
}
// aggressor: transcendental-unit load only (v_exp_f32 chains), small footprint: co-resident with the victim's waves
__global__ __launch_bounds__(256) void exp_aggressor_kernel(float* out, int iters) {
  float v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) v[k] = 0.001f * (float)(threadIdx.x + 256 * k + 1);
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_amdgcn_exp2f(v[k] * 0.5f - 1.0f);
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) s += v[k];
  if (s == 1.2345e-30f) out[threadIdx.x] = s;
}
__global__ void pattern_fill_kernel(unsigned* p, unsigned n) {
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = (i * 0x9E3779B1u) ^ 0x5bd1e995u;
}

// aggressor: global loads whose DESTINATION is an AGPR (hipcc emits them in kernels with more than 256 registers: the unpinned attention
// had 204), consumed by an MFMA (AGPR A operand) or just moved to VGPRs; small blocks, co-resident with the victim
//   MODE 0: load -> AGPR, then MFMA with that AGPR as A operand     MODE 1: load -> AGPR, v_accvgpr_read only     MODE 2: load -> VGPR, MFMA (control)
template <int MODE>
__global__ __launch_bounds__(256) void agpr_load_aggressor_kernel(const u32x4* __restrict__ src, float* __restrict__ out, int iters, unsigned words_mask) {
  const int tid = threadIdx.x;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  hx8 B = __builtin_bit_cast(hx8, src[tid & 63]);
  unsigned idx = (blockIdx.x * 256 + tid) & words_mask;
  u32x4 sink = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it) {
    u32x4 t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const u32x4* p = src + ((idx + (unsigned)k * 4096u) & words_mask);
      if (MODE == 2) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t[k]) : "v"(p) : "memory");
      else asm volatile("global_load_dwordx4 %0, %1, off" : "=a"(t[k]) : "v"(p) : "memory");
    }
    idx = (idx + 16384u) & words_mask;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (MODE == 0) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "a"(t[k]), "v"(B));
      else if (MODE == 2) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc) : "v"(t[k]), "v"(B));
      else { u32x4 v = t[k]; asm volatile("" : "+v"(v)); sink += v; }
    }
  }
  float s = (float)sink[0];
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 1.2345e-30f) out[tid] = s;
}
static int launch_agpr_load_aggressor(int mode, int nblk, int iters, const void* src, unsigned words, float* out, void* stream) {
  switch (mode) {
    case 0: hipLaunchKernelGGL((agpr_load_aggressor_kernel<0>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, out, iters, words - 1); break;
    case 1: hipLaunchKernelGGL((agpr_load_aggressor_kernel<1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, out, iters, words - 1); break;
    case 2: hipLaunchKernelGGL((agpr_load_aggressor_kernel<2>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)src, out, iters, words - 1); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}

// aggressor: an MFMA followed by a scalar write of EXEC (what hipcc emits for `if (lane_condition) x = lds[...]` next to an MFMA: the
// unpinned o_pair kernel had `v_mfma ...; s_and_saveexec_b64 ...` 32 times, the unpinned attention 205 EXEC writes within three instructions
// of an MFMA).  GAP = number of `s_nop 0` between the MFMA and the EXEC write; the mask keeps lanes 0 - 7 and 32 - 39 (`lane % 32 < 8`).
//   MODE 0: MFMA, GAP nops, s_and_saveexec / masked v_mov / s_or exec     MODE 1: no MFMA (the EXEC sequence alone)     MODE 2: MFMA alone
template <int MODE, int GAP>
__global__ __launch_bounds__(256) void exec_aggressor_kernel(const u32x4* __restrict__ ops, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  hx8 A = __builtin_bit_cast(hx8, ops[((blockIdx.x * 4 + wave) % 64 * 8) * 64 + lane]);
  hx8 B = __builtin_bit_cast(hx8, ops[((blockIdx.x * 4 + wave) % 64 * 8 + 4) * 64 + lane]);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  // (MODE 5: the f16 MFMA of MODE 0 with a mask that KEEPS lanes 48 - 63 enabled: if the victim's last pass takes its write enables from
  //  the aggressor's EXEC[63:48], this one must be harmless)
  const unsigned long long mask = MODE == 5 ? 0xFFFF0000000000FFull : 0x000000FF000000FFull;
  unsigned dummy = 0;
  typedef float f32x4_ __attribute__((ext_vector_type(4)));
  f32x4_ acc4 = {0.f, 0.f, 0.f, 0.f};
  float fa = (float)A[0], fb = (float)B[0];
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      unsigned long long save;
      // MODE 0 / 2: 32x32x16 f16 (8 passes); MODE 3: 16x16x32 f16 (4 passes); MODE 4: 32x32x2 f32 (16 passes)
      if (MODE == 0 || MODE == 2 || MODE == 5) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc) : "v"(A), "v"(B));
      if (MODE == 3) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc4) : "v"(A), "v"(B));
      if (MODE == 4) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(fa), "v"(fb));
      // GAP wait states (s_nop k = k + 1 of them) between the MFMA and the EXEC write
      if (GAP >= 1 && GAP <= 16) asm volatile("s_nop %0" : : "n"(GAP >= 1 && GAP <= 16 ? GAP - 1 : 0));
      if (GAP > 16) { asm volatile("s_nop 15"); asm volatile("s_nop %0" : : "n"(GAP > 16 ? GAP - 17 : 0)); }
      if (MODE != 2)
        asm volatile("s_and_saveexec_b64 %0, %2\n\tv_add_u32 %1, 1, %1\n\ts_or_b64 exec, exec, %0" : "=&s"(save), "+v"(dummy) : "s"(mask) : "exec", "scc");
    }
  }
  float s = (float)dummy + acc4[0] + acc4[1] + acc4[2] + acc4[3];
#pragma unroll
  for (int r = 0; r < 16; ++r) s += acc[r];
  if (s == 1.2345e-30f) out[tid] = s;
}
static int launch_exec_aggressor(int mode, int gap, int nblk, int iters, const void* ops, float* out, void* stream) {
#define HZ_EX(M, G) if (mode == M && gap == G) { hipLaunchKernelGGL((exec_aggressor_kernel<M, G>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ops, out, iters); return (int)hipGetLastError(); }
  HZ_EX(0, 0) HZ_EX(0, 1) HZ_EX(0, 2) HZ_EX(0, 4) HZ_EX(0, 6) HZ_EX(0, 8) HZ_EX(0, 10) HZ_EX(0, 12) HZ_EX(0, 16) HZ_EX(0, 24) HZ_EX(1, 0) HZ_EX(2, 0)
  HZ_EX(5, 2) HZ_EX(5, 16)
  HZ_EX(3, 0) HZ_EX(3, 2) HZ_EX(3, 4) HZ_EX(3, 6) HZ_EX(3, 8) HZ_EX(4, 2) HZ_EX(4, 8) HZ_EX(4, 14) HZ_EX(4, 16) HZ_EX(4, 18) HZ_EX(4, 20) HZ_EX(4, 24)
  return -1;
}

// second aggressor family: a pure memory streamer (the EdgeTransition weight-stream pattern, tools/micro/wstream_bench.hip) —
//   MODE 0: LDS-DMA (global_load_lds_dwordx4 through m0, inline asm as in the library), 64 KB chunks, one chunk ahead, s_waitcnt vmcnt(16)
//   MODE 1: plain global_load_dwordx4 into registers at the same cadence
//   MODE 2: LDS-DMA, and the wave ENDS with its last chunk still in flight (no final s_waitcnt)
//   MODE 3: "touch" loads whose result nobody waits for until the end of the kernel (the library's L2 warm-up hand-over)
template <int MODE>
__global__ __launch_bounds__(256, 1) void stream_kernel(unsigned* out, const char* gsrc, int passes) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int REGION = 512 * 1024, CHUNK = 64 * 1024, NCH = REGION / CHUNK;
  const int tid = threadIdx.x;
  u32x4 sink = {0, 0, 0, 0};
  unsigned tok = 0;
  const int n = passes * NCH;
  for (int c = 0; c < n; ++c) {
    const size_t off = ((size_t)c * CHUNK) & (REGION - 1);
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      const size_t a = (off + (size_t)(u * 256 + tid) * 16) & (REGION - 1);
      if (MODE == 0 || MODE == 2) {
        const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long)(const __attribute__((address_space(3))) char*)(smem + ((c & 1) << 16) + (u * 256 + (tid & ~63)) * 16));
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" : : "s"(m0v), "v"(gsrc + a) : "memory", "m0");
      } else if (MODE == 3) {
        asm volatile("global_load_dword %0, %1, off" : "+v"(tok) : "v"(gsrc + a) : "memory");
      } else {
        u32x4 t;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(t) : "v"(gsrc + a) : "memory");
        asm volatile("" : "+v"(t));
        if (u == 15) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); sink += t; }
      }
    }
    if (MODE == 0 || MODE == 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  }
  if (MODE != 2) asm volatile("s_waitcnt vmcnt(0)" : : "v"(tok) : "memory");
  if (MODE != 2) __syncthreads();
  if (sink[0] == 0x12345u) out[tid] = sink[1] + ((unsigned*)smem)[tid] + tok;
}

// MFMA aggressors shaped like the attention's logit phase (the phase whose removal makes the library's kernel harmless: A3_ABL=2):
//   ACC 0: ONE accumulator in VGPRs, every MFMA depends on the previous one     ACC 1: the same chain with the accumulator in AGPRs
//   ACC 2: four independent AGPR accumulators     ACC 3: AGPR chain whose A operand is an AGPR as well     ACC 4: ... whose B operand is
//   ACC 5: A and B from AGPRs (four fragments each)     ACC 6: the same with the accumulator in VGPRs     ACC 7: no MFMA, v_accvgpr moves only
template <int ACC>
__global__ __launch_bounds__(256) void chain_aggressor_kernel(const u32x4* __restrict__ ops, float* __restrict__ out, int iters) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  hx8 A[4], B[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    A[i] = __builtin_bit_cast(hx8, ops[((blockIdx.x * 4 + wave) % 64 * 8 + i) * 64 + lane]);
    B[i] = __builtin_bit_cast(hx8, ops[(((blockIdx.x * 4 + wave) % 64) * 8 + 4 + i) * 64 + lane]);
  }
  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  if (ACC == 0) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(A[u & 3], B[(u >> 2) & 3], acc[0], 0, 0, 0);
  } else if (ACC == 1) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(A[u & 3]), "v"(B[(u >> 2) & 3]));
  } else if (ACC == 2) {
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[u & 3]) : "v"(A[u & 3]), "v"(B[(u >> 2) & 3]));
  } else if (ACC == 3) {
    hx8 Aa = A[0];
    asm volatile("" : "+a"(Aa));
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[0]) : "a"(Aa), "v"(B[(u >> 2) & 3]));
  } else if (ACC == 4) {  // B operand from an AGPR
    hx8 Ba = B[0];
    asm volatile("" : "+a"(Ba));
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(A[u & 3]), "a"(Ba));
  } else if (ACC == 5) {  // A and B operands from AGPRs, four different fragments each
    hx8 Aa[4], Ba[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { Aa[i] = A[i]; Ba[i] = B[i]; asm volatile("" : "+a"(Aa[i]), "+a"(Ba[i])); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(acc[0]) : "a"(Aa[u & 3]), "a"(Ba[(u >> 2) & 3]));
  } else if (ACC == 6) {  // A and B from AGPRs, accumulator in VGPRs
    hx8 Aa[4], Ba[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { Aa[i] = A[i]; Ba[i] = B[i]; asm volatile("" : "+a"(Aa[i]), "+a"(Ba[i])); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+v"(acc[0]) : "a"(Aa[u & 3]), "a"(Ba[(u >> 2) & 3]));
  } else {  // ACC == 7: no MFMA at all, AGPR traffic only (v_accvgpr_read / write chains)
    hx8 Aa = A[0];
    for (int it = 0; it < iters; ++it)
#pragma unroll
      for (int u = 0; u < 16; ++u) { asm volatile("" : "+a"(Aa)); asm volatile("" : "+v"(Aa)); }
    acc[0][0] = (float)Aa[0];
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 1.2345e-30f) out[tid] = s;
}
static int launch_chain_aggressor(int acc, int nblk, int iters, const void* ops, float* out, void* stream) {
  switch (acc) {
    case 0: hipLaunchKernelGGL((chain_aggressor_kernel<0>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ops, out, iters); break;
    case 1: hipLaunchKernelGGL((chain_aggressor_kernel<1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ops, out, iters); break;
    case 2: hipLaunchKernelGGL((chain_aggressor_kernel<2>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ops, out, iters); break;
    case 3: hipLaunchKernelGGL((chain_aggressor_kernel<3>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ops, out, iters); break;
    case 4: hipLaunchKernelGGL((chain_aggressor_kernel<4>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ops, out, iters); break;
    case 5: hipLaunchKernelGGL((chain_aggressor_kernel<5>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ops, out, iters); break;
    case 6: hipLaunchKernelGGL((chain_aggressor_kernel<6>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ops, out, iters); break;
    case 7: hipLaunchKernelGGL((chain_aggressor_kernel<7>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, (const u32x4*)ops, out, iters); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
#ifdef HAZARD_LIB
extern "C" int hz_victim2(int kind, int nblk, int rounds, unsigned* pattern, unsigned pat_words, unsigned* bad, int fill, void* stream) {
  if (fill) { hipLaunchKernelGGL(pattern_fill_kernel, dim3(1024), dim3(256), 0, (hipStream_t)stream, pattern, pat_words); return (int)hipGetLastError(); }
  switch (kind) {
    case 0: hipLaunchKernelGGL((victim2_kernel<0>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, rounds, pattern, pat_words - 1, bad); break;
    case 1: hipLaunchKernelGGL((victim2_kernel<1>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, rounds, pattern, pat_words - 1, bad); break;
    case 2: hipLaunchKernelGGL((victim2_kernel<2>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, rounds, pattern, pat_words - 1, bad); break;
    case 3: hipLaunchKernelGGL((victim2_kernel<3>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, rounds, pattern, pat_words - 1, bad); break;
    case 4: hipLaunchKernelGGL((victim2_kernel<4>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, rounds, pattern, pat_words - 1, bad); break;
    case 5: hipLaunchKernelGGL((victim2_kernel<5>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, rounds, pattern, pat_words - 1, bad); break;
    case 6: hipLaunchKernelGGL((victim2_kernel<6>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, rounds, pattern, pat_words - 1, bad); break;
    case 7: hipLaunchKernelGGL((victim2_kernel<7>), dim3(nblk), dim3(256), 0, (hipStream_t)stream, rounds, pattern, pat_words - 1, bad); break;
    default: return -1;
  }
  return (int)hipGetLastError();
}
extern "C" int hz_exp_aggressor(int nblk, int iters, float* out, void* stream) {
  hipLaunchKernelGGL(exp_aggressor_kernel, dim3(nblk), dim3(256), 0, (hipStream_t)stream, out, iters);
  return (int)hipGetLastError();
}
extern "C" int hz_streamer(int mode, int nblk, int passes, unsigned* out, const void* gsrc, void* stream) {
#define HZ_ST(M)                                                                                                          \
  case M:                                                                                                                 \
    hipFuncSetAttribute((const void*)stream_kernel<M>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);              \
    hipLaunchKernelGGL((stream_kernel<M>), dim3(nblk), dim3(256), 131072, (hipStream_t)stream, out, (const char*)gsrc, passes); \
    break;
  switch (mode) { HZ_ST(0) HZ_ST(1) HZ_ST(2) HZ_ST(3) default: return -1; }
  return (int)hipGetLastError();
}
// the same two kernels behind a C ABI (hipcc -shared -fPIC -DHAZARD_LIB -> libhazard.so) for tools/hazard_lib_repro.py, which crosses
// them with the library's own rot_score_kernel / forward: {stand-alone, library} victim x {stand-alone, library} aggressor
extern "C" int hz_victim(int n_items, const float* q, double sigma, double* res, const double* expect, unsigned* bad, void* stream) {
  launch_victim<256, 0>((hipStream_t)stream, n_items, q, sigma, res, expect, bad);
  return (int)hipGetLastError();
}
extern "C" int hz_victim_int(int n_items, const float* q, double sigma, double* res, const double* expect, unsigned* bad, void* stream) {
  launch_victim<256, 1>((hipStream_t)stream, n_items, q, sigma, res, expect, bad);
  return (int)hipGetLastError();
}
extern "C" int hz_agpr_load_aggressor(int mode, int nblk, int iters, const void* src, unsigned words, float* out, void* stream) {
  return launch_agpr_load_aggressor(mode, nblk, iters, src, words, out, stream);
}
extern "C" int hz_exec_aggressor(int mode, int gap, int nblk, int iters, const void* ops, float* out, void* stream) {
  return launch_exec_aggressor(mode, gap, nblk, iters, ops, out, stream);
}
extern "C" int hz_chain_aggressor(int acc, int nblk, int iters, const void* ops, float* out, void* stream) {
  return launch_chain_aggressor(acc, nblk, iters, ops, out, stream);
}
// the MFMA aggressor with a caller-chosen LDS footprint and block size: lds_bytes = 0 and 256 threads leaves room for victim waves on the
// SAME SIMDs (co-residency), which the 152 KB / 512-thread form excludes by construction
extern "C" int hz_aggressor_co(int nblk, int threads, int iters, const void* ops, float* out, int lds_bytes, void* stream) {
  if (lds_bytes < 8192) lds_bytes = 8192;
  hipFuncSetAttribute((const void*)aggressor_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  hipLaunchKernelGGL((aggressor_kernel<0>), dim3(nblk), dim3(threads), lds_bytes, (hipStream_t)stream, (const u32x4*)ops, out, iters, (const u32x4*)ops, (size_t)1024);
  return (int)hipGetLastError();
}
extern "C" int hz_aggressor(int nblk, int iters, const void* ops, float* out, int mem, const void* gsrc, size_t gwords, void* stream) {
  hipFuncSetAttribute((const void*)aggressor_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  hipFuncSetAttribute((const void*)aggressor_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);
  if (mem) hipLaunchKernelGGL((aggressor_kernel<1>), dim3(nblk), dim3(512), 152 * 1024, (hipStream_t)stream, (const u32x4*)ops, out, iters, (const u32x4*)gsrc, gwords);
  else hipLaunchKernelGGL((aggressor_kernel<0>), dim3(nblk), dim3(512), 152 * 1024, (hipStream_t)stream, (const u32x4*)ops, out, iters, (const u32x4*)gsrc, gwords);
  return (int)hipGetLastError();
}
#else
int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 4.0;
  u32x4 *d_ops, *d_g; float *d_out, *d_q; double *d_res, *d_expect; unsigned* d_bad;
  const size_t gwords = (size_t)64 << 20;  // 1 GB read stream (HBM)
  hipMalloc(&d_ops, (size_t)64 * 8 * 64 * 16); hipMalloc(&d_g, gwords * 16); hipMemset(d_g, 1, gwords * 16);
  hipMalloc(&d_out, 4096); hipMalloc(&d_q, 64); hipMalloc(&d_res, 2896 * 24 + 64); hipMalloc(&d_expect, 64); hipMalloc(&d_bad, 8192);
  const float hq[8] = {0.9238795f, 0.2209424f, -0.1913417f, 0.2514080f, 0.3826834f, -0.5334021f, 0.6532815f, 0.3753303f};  // two unit quaternions: omega ~ 2.4 rad
  hipMemcpy(d_q, hq, 32, hipMemcpyHostToDevice);
  hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
  char name[160];
  // ---- MFMA followed by a scalar EXEC write (co-resident): the pattern the library's failures were traced to
  {
    fill_ops(d_ops, 1);
    const int NC = 12;
    const int cases[NC][2] = {{0, 0}, {0, 2}, {0, 4}, {0, 8}, {0, 10}, {0, 16}, {0, 24}, {5, 16}, {3, 4}, {4, 16}, {1, 0}, {2, 0}};
    const char* cn[NC] = {"f16 32x32x16; gap 0; masked VALU", "f16 32x32x16; gap 2; masked VALU", "f16 32x32x16; gap 4; masked VALU", "f16 32x32x16; gap 8; masked VALU",
                          "f16 32x32x16; gap 10; masked VALU", "f16 32x32x16; gap 16; masked VALU", "f16 32x32x16; gap 24; masked VALU",
                          "f16; gap 16; mask keeps lanes 48-63", "f16 16x16x32; gap 4; masked VALU", "fp32 32x32x2; gap 16; masked VALU",
                          "no MFMA, EXEC sequence alone", "MFMA alone"};
    for (int intonly = 0; intonly < 2; ++intonly)
      for (int c = 0; c < NC; ++c) {
        const int n_items = 2896;
        hipDeviceSynchronize();
        if (intonly) launch_victim<256, 1>(sa, n_items, d_q, 0.6, d_res, nullptr, d_bad); else launch_victim<256, 0>(sa, n_items, d_q, 0.6, d_res, nullptr, d_bad);
        hipStreamSynchronize(sa);
        hipMemcpy(d_expect, d_res, 24, hipMemcpyDeviceToDevice);
        hipMemset(d_bad, 0, 8192);
        long launches = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
          for (int i = 0; i < 8; ++i) launch_exec_aggressor(cases[c][0], cases[c][1], 512, 1500, d_ops, d_out, sb);
          for (int i = 0; i < 48; ++i) {
            if (intonly) launch_victim<256, 1>(sa, n_items, d_q, 0.6, d_res, d_expect, d_bad); else launch_victim<256, 0>(sa, n_items, d_q, 0.6, d_res, d_expect, d_bad);
          }
          launches += 48;
          hipStreamSynchronize(sa);
          hipStreamSynchronize(sb);
        }
        unsigned hb[8];
        hipMemcpy(hb, d_bad, 32, hipMemcpyDeviceToHost);
        printf("co-resident | %-7s victim | %-38s victim launches %6ld  items %9ld  BAD %u  [item%%4: %u %u %u %u]\n", intonly ? "integer" : "fp64", cn[c], launches,
               launches * n_items, hb[0], hb[1], hb[2], hb[3], hb[4]);
        fflush(stdout);
      }
  }
  // ---- co-resident MFMA aggressors (small blocks, no LDS: the victim's waves share their SIMDs).  The library's failures were traced to
  // this class (tools/hazard_lib_repro.py): an MFMA that reads its A / B operand from the AGPR half of the register file
  {
    const char* an[8] = {"VGPR A/B, compiler-chosen accumulator", "VGPR A/B, AGPR accumulator chain", "VGPR A/B, four AGPR accumulators", "A operand from an AGPR",
                         "B operand from an AGPR", "A and B from AGPRs", "A and B from AGPRs, VGPR accumulator", "no MFMA, v_accvgpr moves only"};
    fill_ops(d_ops, 1);
    for (int intonly = 0; intonly < 2; ++intonly)
      for (int acc : {1, 3, 4, 5, 6, 7}) {
        const int n_items = 2896;
        hipDeviceSynchronize();
        if (intonly) launch_victim<256, 1>(sa, n_items, d_q, 0.6, d_res, nullptr, d_bad); else launch_victim<256, 0>(sa, n_items, d_q, 0.6, d_res, nullptr, d_bad);
        hipStreamSynchronize(sa);
        hipMemcpy(d_expect, d_res, 24, hipMemcpyDeviceToDevice);
        hipMemset(d_bad, 0, 8192);
        long launches = 0;
        const auto t0 = std::chrono::steady_clock::now();
        while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds * 2) {
          for (int i = 0; i < 8; ++i) launch_chain_aggressor(acc, 512, 900, d_ops, d_out, sb);
          for (int i = 0; i < 48; ++i) {
            if (intonly) launch_victim<256, 1>(sa, n_items, d_q, 0.6, d_res, d_expect, d_bad); else launch_victim<256, 0>(sa, n_items, d_q, 0.6, d_res, d_expect, d_bad);
          }
          launches += 48;
          hipStreamSynchronize(sa);
          hipStreamSynchronize(sb);
        }
        unsigned hb[8];
        hipMemcpy(hb, d_bad, 32, hipMemcpyDeviceToHost);
        printf("co-resident | %-7s victim | MFMA chain: %-40s victim launches %6ld  items %9ld  BAD %u  [item%%4: %u %u %u %u]\n", intonly ? "integer" : "fp64", an[acc],
               launches, launches * n_items, hb[0], hb[1], hb[2], hb[3], hb[4]);
        fflush(stdout);
      }
  }
  if (argc > 2) return 0;  // (second argument: the co-resident section only)
  for (double sigma : {0.25, 1.0}) {
    snprintf(name, sizeof name, "sigma %.2f  fp64 victim 256 thr, NO aggressor", sigma);
    cell<256, 0>(name, -1, 0, 0, seconds * 0.5, sigma, d_ops, d_g, gwords, d_out, d_q, d_res, d_expect, d_bad, sa, sb);
    for (int reserve : {32, 48, 56, 96}) {
      for (int pattern : {1, 0}) {
        for (int mem : {0, 1}) {
          snprintf(name, sizeof name, "sigma %.2f  fp64 victim 256 thr | MFMA %s%s", sigma, pattern ? "random" : "zeros", mem ? " + HBM reads" : "");
          cell<256, 0>(name, pattern, mem, reserve, seconds, sigma, d_ops, d_g, gwords, d_out, d_q, d_res, d_expect, d_bad, sa, sb);
        }
      }
      snprintf(name, sizeof name, "sigma %.2f  fp64 victim  64 thr | MFMA random + HBM reads", sigma);
      cell<64, 0>(name, 1, 1, reserve, seconds, sigma, d_ops, d_g, gwords, d_out, d_q, d_res, d_expect, d_bad, sa, sb);
      snprintf(name, sizeof name, "sigma %.2f  INTEGER victim 256 thr | MFMA random + HBM reads", sigma);
      cell<256, 1>(name, 1, 1, reserve, seconds, sigma, d_ops, d_g, gwords, d_out, d_q, d_res, d_expect, d_bad, sa, sb);
    }
  }
  return 0;
}
#endif
