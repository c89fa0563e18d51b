// cumask_probe.hip — does hipExtStreamCreateWithCUMask restrict placement on this box, and which (XCC, SE, CU) does mask bit k name?
// Build: hipcc --offload-arch=gfx950 -O2 tools/micro/cumask_probe.hip -o /tmp/cumask_probe ; run: /tmp/cumask_probe
// Output: for each mask, the set of physical CUs (xcc.se.cu) on which blocks of a 2048-block spin kernel ran; then two kernels on two
// streams with complementary masks run together and the probe checks that no physical CU was shared.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstring>
#include <map>
#include <set>
#include <vector>

#define CHECK(x)                                                                   \
  do {                                                                             \
    hipError_t e_ = (x);                                                           \
    if (e_ != hipSuccess) {                                                        \
      printf("%s failed: %s\n", #x, hipGetErrorString(e_));                        \
      return 1;                                                                    \
    }                                                                              \
  } while (0)

__global__ void probe_kernel(unsigned* out, long spin) {
  const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
  const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
  const long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin) {
  }
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = hw;
    out[2 * blockIdx.x + 1] = xcc;
  }
}

static unsigned phys(unsigned hw, unsigned xcc) {  // xcc[3:0] | se[2:0] | sh | cu[3:0]
  const unsigned cu = (hw >> 8) & 15, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
  return ((xcc & 15) << 8) | (se << 5) | (sh << 4) | cu;
}

static int run(hipStream_t st, unsigned* d_out, int blocks, std::set<unsigned>& cus) {
  std::vector<unsigned> h(2 * blocks);
  hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(64), 0, st, d_out, 2000L);  // 20 us at 100 MHz
  CHECK(hipStreamSynchronize(st));
  CHECK(hipMemcpy(h.data(), d_out, h.size() * 4, hipMemcpyDeviceToHost));
  for (int b = 0; b < blocks; ++b) cus.insert(phys(h[2 * b], h[2 * b + 1]));
  return 0;
}

static void print_set(const char* tag, const std::set<unsigned>& cus) {
  std::map<unsigned, int> per_xcc;
  for (unsigned c : cus) per_xcc[c >> 8]++;
  printf("%-28s %3zu CUs; per XCC:", tag, cus.size());
  for (auto& kv : per_xcc) printf(" %u:%d", kv.first, kv.second);
  printf("\n");
}

int main() {
  hipDeviceProp_t prop;
  CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs\n", prop.gcnArchName, prop.multiProcessorCount);
  const int blocks = 4096;
  unsigned *d_a, *d_b;
  CHECK(hipMalloc(&d_a, 2 * blocks * 4));
  CHECK(hipMalloc(&d_b, 2 * blocks * 4));
  {
    std::set<unsigned> cus;
    if (run(nullptr, d_a, blocks, cus)) return 1;
    print_set("null stream", cus);
  }
  const int words = 8;
  // single-bit masks: which physical CU is bit k?
  for (int k = 0; k < 256; k += (k < 10 ? 1 : 41)) {
    unsigned m[words];
    memset(m, 0, sizeof(m));
    m[k >> 5] = 1u << (k & 31);
    hipStream_t st;
    hipError_t e = hipExtStreamCreateWithCUMask(&st, words, m);
    if (e != hipSuccess) {
      printf("bit %d: create failed: %s\n", k, hipGetErrorString(e));
      continue;
    }
    std::set<unsigned> cus;
    if (run(st, d_a, 256, cus)) return 1;
    // (an XCC whose part of the mask is empty turns out to be unrestricted: print only the XCCs that were narrowed)
    std::map<unsigned, std::vector<unsigned>> by;
    for (unsigned c : cus) by[c >> 8].push_back(c);
    printf("bit %3d ->", k);
    for (auto& kv : by)
      if (kv.second.size() < 8)
        for (unsigned c : kv.second) printf(" xcc%u.se%u.sh%u.cu%u", c >> 8, (c >> 5) & 7, (c >> 4) & 1, c & 15);
      else printf(" xcc%u:all(%zu)", kv.first, kv.second.size());
    printf("\n");
    CHECK(hipStreamDestroy(st));
  }
  // split masks: side = the low `ns` bits, main = the rest; and side = every 16th bit
  for (int variant = 0; variant < 4; ++variant) {
    unsigned ms[words], mm[words];
    memset(ms, 0, sizeof(ms));
    const int ns = variant == 0 ? 16 : variant == 1 ? 32 : variant == 2 ? 16 : 32;
    for (int k = 0; k < ns; ++k) {
      const int bit = variant < 2 ? k : (variant == 2 ? k * 16 : k * 8);
      ms[bit >> 5] |= 1u << (bit & 31);
    }
    for (int w = 0; w < words; ++w) mm[w] = ~ms[w];
    hipStream_t ss, sm;
    CHECK(hipExtStreamCreateWithCUMask(&ss, words, ms));
    CHECK(hipExtStreamCreateWithCUMask(&sm, words, mm));
    std::set<unsigned> cs, cm;
    // both in flight together
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(64), 0, sm, d_a, 2000L);
    hipLaunchKernelGGL(probe_kernel, dim3(blocks), dim3(64), 0, ss, d_b, 2000L);
    CHECK(hipDeviceSynchronize());
    std::vector<unsigned> ha(2 * blocks), hb(2 * blocks);
    CHECK(hipMemcpy(ha.data(), d_a, ha.size() * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(hb.data(), d_b, hb.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < blocks; ++b) {
      cm.insert(phys(ha[2 * b], ha[2 * b + 1]));
      cs.insert(phys(hb[2 * b], hb[2 * b + 1]));
    }
    int shared = 0;
    for (unsigned c : cs) shared += cm.count(c);
    char tag[64];
    snprintf(tag, sizeof(tag), "variant %d main (256-%d)", variant, ns);
    print_set(tag, cm);
    snprintf(tag, sizeof(tag), "variant %d side (%d)", variant, ns);
    print_set(tag, cs);
    printf("  shared physical CUs: %d\n", shared);
    CHECK(hipStreamDestroy(ss));
    CHECK(hipStreamDestroy(sm));
  }
  return 0;
}
