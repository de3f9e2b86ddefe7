// Microbenchmark: does the 256 MiB Infinity Cache (MALL) retain a weight slab between two
// kernels, and do non-temporal loads hit in it?  Decides whether prefetching the next GEMV's
// weights during the latency-bound attention kernel can pay.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_mall.hip -o kuiperllama_amd/lib/mb_mall
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT>
__global__ __launch_bounds__(256) void read_sum(const f32x4* __restrict__ p, size_t n4, float* out) {
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nw = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
  // each wave walks 8 KiB tiles (8 x 1 KiB loads in flight), tiles interleaved across waves
  for (size_t t = wave; t * 512 < n4; t += nw) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      size_t idx = t * 512 + u * 64 + lane;
      if (idx >= n4) idx = 0;
      v[u] = NT ? __builtin_nontemporal_load(p + idx) : p[idx];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u].x + v[u].y + v[u].z + v[u].w;
  }
  if (acc == 123.456f) out[0] = acc;  // keep the loads
}

static float run(int nt, const f32x4* p, size_t bytes, float* out, hipStream_t s, hipEvent_t e0, hipEvent_t e1, int grid) {
  hipEventRecord(e0, s);
  if (nt) hipLaunchKernelGGL(read_sum<1>, dim3(grid), dim3(256), 0, s, p, bytes / 16, out);
  else hipLaunchKernelGGL(read_sum<0>, dim3(grid), dim3(256), 0, s, p, bytes / 16, out);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0; hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t FLUSH = (size_t)2 << 30;
  char *flush, *buf; float* out;
  CK(hipMalloc(&flush, FLUSH)); CK(hipMalloc(&buf, (size_t)1 << 30)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(flush, 1, FLUSH)); CK(hipMemset(buf, 1, (size_t)1 << 30));
  CK(hipDeviceSynchronize());
  const size_t sizes[] = {16u << 20, 32u << 20, 64u << 20, 128u << 20, 192u << 20, 256u << 20, 512u << 20};
  printf("# size_MB first_load second_load : GB/s of pass1(cold) pass2 pass3 (same kind), then cross: prefetch kind -> consumer kind\n");
  for (int grid : {1024, 2048}) {
    for (size_t sz : sizes) {
      for (int nt = 0; nt < 2; ++nt) {
        run(0, (const f32x4*)flush, FLUSH, out, s, e0, e1, 2048);  // evict
        float a = run(nt, (const f32x4*)buf, sz, out, s, e0, e1, grid);
        float b = run(nt, (const f32x4*)buf, sz, out, s, e0, e1, grid);
        float c = run(nt, (const f32x4*)buf, sz, out, s, e0, e1, grid);
        printf("grid %d size %4zu MB %s  cold %7.1f  again %7.1f  again %7.1f GB/s  (%.1f us)\n", grid, sz >> 20,
               nt ? "nt   " : "plain", sz / a / 1e6, sz / b / 1e6, sz / c / 1e6, c * 1e3);
      }
      // cross: plain prefetch then nt consumer, and nt prefetch then nt consumer measured above
      run(0, (const f32x4*)flush, FLUSH, out, s, e0, e1, 2048);
      run(0, (const f32x4*)buf, sz, out, s, e0, e1, grid);
      float d = run(1, (const f32x4*)buf, sz, out, s, e0, e1, grid);
      printf("grid %d size %4zu MB plain->nt consumer %7.1f GB/s\n", grid, sz >> 20, sz / d / 1e6);
    }
  }
  // launch floor: empty-ish kernels back to back
  {
    hipEventRecord(e0, s);
    for (int i = 0; i < 200; ++i) hipLaunchKernelGGL(read_sum<0>, dim3(1), dim3(256), 0, s, (const f32x4*)buf, (size_t)64, out);
    hipEventRecord(e1, s); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("tiny kernel chain: %.2f us per launch (eager)\n", ms * 1e3 / 200);
  }
  return 0;
}
