// mb_grid_barrier.hip — what does a kernel boundary cost on MI355X, and what would a grid barrier
// inside ONE persistent kernel cost instead?  (every decode GEMV sits 0.7–2 us above
// its pure-stream floor and the floor itself is ~2.3 us above bytes / 8 TB/s — launch ramp + drain.)
//
// Workload per phase = one decode-GEMV-like pass: y[r] = sum_c float(M[r][c]) * x[c] over a 4096 x 4096
// int8 matrix (16.8 MB, a DISTINCT slab per phase so nothing hits a cache), x = the previous phase's y
// (every workgroup stages all of x in LDS, as the product kernels do), wave per row pair, whole row in
// flight (4 x 16 B per lane per row), 512 workgroups x 256 threads.
//   A  graph    : P kernel launches with the usual in-stream dependencies, one hipGraph
//   B  flat     : ONE persistent kernel, P phases, one agent-scope counter as grid barrier
//   C  tree     : the same with 8 group counters (blockIdx & 7 = XCD) feeding a master counter
//   D  tree+pf  : C, and the NEXT phase's weight tile is requested BEFORE the barrier (weights do not
//                 depend on the previous phase) so the barrier wait hides the HBM round trip
//   E  nosync   : B without the barrier (wrong results; the pure streaming time of the phases)
// The spin loops are bounded (an error flag is raised instead of hanging).  Results of A, B, C, D must be
// identical (same arithmetic) — printed as a checksum.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_grid_barrier.hip -o kuiperllama_amd/lib/mb_grid_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

constexpr int ROWS = 4096, COLS = 4096, U = 4, WG = 256, NWG = 512;
constexpr size_t SLAB = (size_t)ROWS * COLS;

struct Tile { i32x4 q0[U], q1[U]; };

__device__ __forceinline__ void load_tile(Tile& t, const char* M, int pair, int lane) {
  const i32x4* r0 = (const i32x4*)(M + (size_t)(2 * pair) * COLS);
  const i32x4* r1 = (const i32x4*)(M + (size_t)(2 * pair + 1) * COLS);
#pragma unroll
  for (int u = 0; u < U; ++u) {
    t.q0[u] = __builtin_nontemporal_load(r0 + u * 64 + lane);
    t.q1[u] = __builtin_nontemporal_load(r1 + u * 64 + lane);
  }
}
__device__ __forceinline__ float dot16(const i32x4& q, const float* xs) {
  float a = 0.f;
  const int w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const f32x4 x = *(const f32x4*)(xs + 4 * k);
    a += (float)(int8_t)(w[k] & 0xff) * x.x + (float)(int8_t)((w[k] >> 8) & 0xff) * x.y +
         (float)(int8_t)((w[k] >> 16) & 0xff) * x.z + (float)(w[k] >> 24) * x.w;
  }
  return a;
}
__device__ __forceinline__ void stage_x(float* xs, const float* x) {
  for (int i = threadIdx.x; i < COLS / 4; i += WG) ((f32x4*)xs)[i] = ((const f32x4*)x)[i];
  __syncthreads();
}
__device__ __forceinline__ void finish(const Tile& t, const float* xs, float* y, int pair, int lane) {
  float a0 = 0.f, a1 = 0.f;
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const float* xp = xs + 16 * (u * 64 + lane);
    a0 += dot16(t.q0[u], xp);
    a1 += dot16(t.q1[u], xp);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a0 += __shfl_xor(a0, o);
    a1 += __shfl_xor(a1, o);
  }
  if (lane == 0) {
    y[2 * pair] = a0 * 2.1e-4f;
    y[2 * pair + 1] = a1 * 2.1e-4f;
  }
}

__global__ __launch_bounds__(WG) void k_phase(const char* M, const float* x, float* y) {
  __shared__ __attribute__((aligned(16))) float xs[COLS];
  const int lane = threadIdx.x & 63, pair = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
  Tile t;
  load_tile(t, M, pair, lane);
  stage_x(xs, x);
  finish(t, xs, y, pair, lane);
}

struct Sync { unsigned* master; unsigned* group; int* err; };
#define SPIN_MAX (1 << 22)

template <int MODE /* 1 flat, 2 tree, 3 tree+prefetch, 4 nosync */>
__global__ __launch_bounds__(WG) void k_persist(const char* M0, int nslab, float* b0, float* b1, int P, Sync s) {
  __shared__ __attribute__((aligned(16))) float xs[COLS];
  const int lane = threadIdx.x & 63, pair = blockIdx.x * (WG / 64) + (threadIdx.x >> 6);
  Tile t;
  if (MODE == 3) load_tile(t, M0, pair, lane);
  for (int ph = 0; ph < P; ++ph) {
    const char* M = M0 + (size_t)(ph % nslab) * SLAB;
    const float* x = (ph & 1) ? b1 : b0;
    float* y = (ph & 1) ? b0 : b1;
    if (MODE != 3) load_tile(t, M, pair, lane);
    stage_x(xs, x);
    finish(t, xs, y, pair, lane);
    if (MODE == 3 && ph + 1 < P) load_tile(t, M0 + (size_t)((ph + 1) % nslab) * SLAB, pair, lane);
    if (MODE == 4) { __syncthreads(); continue; }
    // ---- grid barrier: stores drained by every wave, one lane releases and takes a ticket ----------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      unsigned target;
      if (MODE == 1) {
        __hip_atomic_fetch_add(s.master, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        target = (unsigned)(ph + 1) * NWG;
      } else {
        const int g = blockIdx.x & 7;
        const unsigned old = __hip_atomic_fetch_add(s.group + 32 * g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (old == (unsigned)(ph + 1) * (NWG / 8) - 1)
          __hip_atomic_fetch_add(s.master, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        target = (unsigned)(ph + 1) * 8;
      }
      int spins = 0;
      while (__hip_atomic_load(s.master, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SPIN_MAX) { *s.err = 1; break; }
      }
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    __syncthreads();
  }
}

__global__ void k_fill(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    p[i] = x;
  }
}

int main() {
  const int NSLAB = 32, P = 64;
  char* M;
  CK(hipMalloc(&M, SLAB * NSLAB));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)M, SLAB * NSLAB / 4, 12345u);
  float *b0, *b1;
  CK(hipMalloc(&b0, ROWS * 4));
  CK(hipMalloc(&b1, ROWS * 4));
  std::vector<float> x0(ROWS);
  for (int i = 0; i < ROWS; ++i) x0[i] = (float)((i * 37) % 101 - 50) / 50.f;
  unsigned* cnt;
  CK(hipMalloc(&cnt, 4096));
  int* err;
  CK(hipMalloc(&err, 4));
  Sync s{cnt, cnt + 64, err};
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  // graph of P launches
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
  for (int ph = 0; ph < P; ++ph)
    hipLaunchKernelGGL(k_phase, dim3(NWG), dim3(WG), 0, st, M + (size_t)(ph % NSLAB) * SLAB, (ph & 1) ? b1 : b0, (ph & 1) ? b0 : b1);
  CK(hipStreamEndCapture(st, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  auto reset = [&] {
    CK(hipMemcpyAsync(b0, x0.data(), ROWS * 4, hipMemcpyHostToDevice, st));
    CK(hipMemsetAsync(b1, 0, ROWS * 4, st));
    CK(hipMemsetAsync(cnt, 0, 4096, st));
    CK(hipMemsetAsync(err, 0, 4, st));
  };
  auto report = [&](const char* name, float ms) {
    std::vector<float> y(ROWS);
    int herr = 0;
    CK(hipMemcpy(y.data(), (P & 1) ? b1 : b0, ROWS * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    double cs = 0;
    for (int i = 0; i < ROWS; ++i) cs += (double)y[i] * (1 + i % 7);
    printf("%-10s %8.2f us per phase  (%.1f MB in %.2f us = %.2f TB/s)  checksum %.6e%s\n", name, ms * 1e3 / P, SLAB / 1e6,
           ms * 1e3 / P, SLAB / (ms * 1e-3 / P) / 1e12, cs, herr ? "  SPIN LIMIT HIT" : "");
  };
  for (int rep = 0; rep < 2; ++rep) {
    float best = 1e9f;
    for (int it = 0; it < 5; ++it) {
      reset();
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      best = ms < best ? ms : best;
    }
    report("A graph", best);
    auto run = [&](const char* name, auto kern) {
      float b = 1e9f;
      for (int it = 0; it < 5; ++it) {
        reset();
        CK(hipEventRecord(e0, st));
        hipLaunchKernelGGL(kern, dim3(NWG), dim3(WG), 0, st, (const char*)M, NSLAB, b0, b1, P, s);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        b = ms < b ? ms : b;
      }
      report(name, b);
    };
    run("B flat", k_persist<1>);
    run("C tree", k_persist<2>);
    run("D tree+pf", k_persist<3>);
    run("E nosync", k_persist<4>);
  }
  return 0;
}
