// mb_xcd.hip — what does one hand-over between two workgroups cost when both sit on the SAME XCD (one L2) and when
// they sit on different XCDs, for the store / load flavours the ISA offers?
//
// Every in-kernel hand-over measured this round used agent-scope accesses (global_store sc1 = write-through to the
// memory side, global_load sc1 = served past this XCD's L2): 1.3-1.5 us per hop, which is why no persistent or
// overlapped structure beat the 1.55 us launch boundary.  Two workgroups of one XCD share an L2, so a store that has
// reached L2 (the vector L1 is write-through) and a load that misses L1 (sc0) should meet there at L2 latency.
//
// Ping-pong: workgroup A stores k to word a and spins on word b until it reads k; workgroup B spins on a until it reads
// k and stores k to b; 2000 round trips, one lane each, 100 MHz wall clock around them -> ns per one-way hop.  The
// workgroup -> XCD placement is READ (s_getreg_b32 HW_REG_XCC_ID), not assumed: grid = 16 workgroups, the program
// picks a same-XCD pair and a cross-XCD pair from what the launch reports.  Spins are bounded (a flavour that never
// becomes visible reports "no progress").
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mb_xcd.hip -o kuiperllama_amd/lib/mb_xcd
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

#define ROUNDS 2000
#define SPIN_LIMIT 200000

__global__ void k_where(int* xcc) {
  int id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
  if (threadIdx.x == 0) xcc[blockIdx.x] = id & 0xf;
}

// ST: 0 plain store, 1 sc0, 2 sc1 (agent), 3 sc0 sc1 (system)      LD: the same for loads
template <int ST>
__device__ __forceinline__ void st_flavour(unsigned* p, unsigned v) {
  if (ST == 0) asm volatile("global_store_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
  if (ST == 1) asm volatile("global_store_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
  if (ST == 2) asm volatile("global_store_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
  if (ST == 3) asm volatile("global_store_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" ::"v"(p), "v"(v) : "memory");
}
template <int LD>
__device__ __forceinline__ unsigned ld_flavour(const unsigned* p) {
  unsigned v;
  if (LD == 0) asm volatile("global_load_dword %0, %1, off\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 1) asm volatile("global_load_dword %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 2) asm volatile("global_load_dword %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  if (LD == 3) asm volatile("global_load_dword %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
  return v;
}

template <int ST, int LD>
__global__ void k_pingpong(unsigned* a, unsigned* b, int wg_a, int wg_b, unsigned long long* out) {
  if (threadIdx.x != 0) return;
  const int me = (int)blockIdx.x;
  if (me != wg_a && me != wg_b) return;
  unsigned long long t0 = 0;
  bool stuck = false;
  if (me == wg_a) {
    t0 = wall_clock64();
    for (unsigned k = 1; k <= ROUNDS && !stuck; ++k) {
      st_flavour<ST>(a, k);
      int spins = 0;
      while (ld_flavour<LD>(b) != k)
        if (++spins > SPIN_LIMIT) {
          stuck = true;
          break;
        }
    }
    out[0] = wall_clock64() - t0;
    out[1] = stuck ? 1 : 0;
    if (stuck) st_flavour<3>(a, 0xffffffffu);  // release the partner
  } else {
    for (unsigned k = 1; k <= ROUNDS; ++k) {
      int spins = 0;
      unsigned v;
      while ((v = ld_flavour<LD>(a)) != k) {
        if (v == 0xffffffffu || ++spins > SPIN_LIMIT) return;
      }
      st_flavour<ST>(b, k);
    }
  }
}

// background load: every other CU streams a large buffer with 16-byte non-temporal loads (the decode step's weight
// stream) while the ping-pong runs
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void k_stream(const f32x4_t* p, size_t n4, int iters, float* sink) {
  float acc = 0.f;
  for (int it = 0; it < iters; ++it)
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
      const f32x4_t v = __builtin_nontemporal_load(p + i);
      acc += v.x + v.y + v.z + v.w;
    }
  if (acc == 123.456f) *sink = acc;
}
static f32x4_t* g_big = nullptr;
static size_t g_big_n4 = 0;
static float* g_sink = nullptr;
static hipStream_t g_s2 = nullptr;
static bool g_loaded = false;

template <int ST, int LD>
static void run(const char* what, unsigned* a, unsigned* b, int wa, int wb, unsigned long long* out) {
  CK(hipMemset(a, 0, 256));
  CK(hipMemset(b, 0, 256));
  CK(hipMemset(out, 0, 16));
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((k_pingpong<ST, LD>), dim3(16), dim3(64), 0, 0, a, b, wa, wb, out);
  if (g_loaded)  // ~6 ms of HBM streaming on the rest of the chip, started right behind the ping-pong launch
    hipLaunchKernelGGL(k_stream, dim3(2048), dim3(256), 0, g_s2, g_big, g_big_n4, 10, g_sink);
  CK(hipDeviceSynchronize());
  unsigned long long h[2];
  CK(hipMemcpy(h, out, 16, hipMemcpyDeviceToHost));
  static const char* nm[4] = {"plain", "sc0", "sc1", "sc0 sc1"};
  if (h[1])
    printf("  %-12s store %-8s load %-8s: no progress (the other workgroup's store never became visible)\n", what,
           nm[ST], nm[LD]);
  else
    printf("  %-12s store %-8s load %-8s: %7.0f ns per one-way hop\n", what, nm[ST], nm[LD],
           (double)h[0] * 10.0 / (2.0 * ROUNDS));
}

int main() {
  int* xcc;
  CK(hipMalloc(&xcc, 16 * 4));
  hipLaunchKernelGGL(k_where, dim3(16), dim3(64), 0, 0, xcc);
  CK(hipDeviceSynchronize());
  int h[16];
  CK(hipMemcpy(h, xcc, 64, hipMemcpyDeviceToHost));
  printf("workgroup -> XCC_ID of a 16-workgroup launch:");
  for (int i = 0; i < 16; ++i) printf(" %d", h[i]);
  printf("\n");
  int same_b = -1, diff_b = -1;
  for (int i = 1; i < 16; ++i) {
    if (h[i] == h[0] && same_b < 0) same_b = i;
    if (h[i] != h[0] && diff_b < 0) diff_b = i;
  }
  // flags on separate 128-byte lines
  unsigned *buf, *a, *b;
  unsigned long long* out;
  CK(hipMalloc(&buf, 4096));
  CK(hipMalloc(&out, 16));
  a = buf;
  b = buf + 64;
  printf("(placement is per launch; the ping-pong launches use the same grid, so the same mapping is assumed to repeat -\n"
         " the same-XCD rows would otherwise look like the cross-XCD ones)\n");
  g_big_n4 = (size_t)4 << 30 >> 4;  // 4 GiB
  CK(hipMalloc(&g_big, g_big_n4 * 16));
  CK(hipMemset(g_big, 0, g_big_n4 * 16));
  CK(hipMalloc(&g_sink, 4));
  CK(hipStreamCreateWithFlags(&g_s2, hipStreamNonBlocking));
  for (int pass = 0; pass < 4; ++pass) {
    g_loaded = pass >= 2;
    const int wb = (pass & 1) == 0 ? same_b : diff_b;
    const char* what = pass == 0 ? "same XCD" : pass == 1 ? "other XCD" : pass == 2 ? "same, LOADED" : "other, LOADED";
    if (wb < 0) {
      printf("  no %s pair in this launch\n", what);
      continue;
    }
    printf("%s: workgroups 0 and %d\n", what, wb);
    run<0, 0>(what, a, b, 0, wb, out);
    run<0, 1>(what, a, b, 0, wb, out);
    run<1, 1>(what, a, b, 0, wb, out);
    run<0, 2>(what, a, b, 0, wb, out);
    run<2, 2>(what, a, b, 0, wb, out);
    run<2, 1>(what, a, b, 0, wb, out);
    run<3, 3>(what, a, b, 0, wb, out);
  }
  return 0;
}
