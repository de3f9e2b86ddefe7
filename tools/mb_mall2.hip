// mb_mall2.hip - does a weight subset read with the default cache policy survive, in the 256-MiB Infinity Cache, a
// 4.7-GB stream of non-temporal loads (the rest of a token's weights)?  If it did, a resident subset of the model
// would be served from the memory-side cache in every token.  Per size S: A (S bytes) is read with plain loads, then
// B (4.7 GB) with nt (or plain) loads, five rounds; the rate of A in rounds 2-5 against its cold rate.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_mall2.hip -o kuiperllama_amd/lib/mb_mall2
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int NT>
__global__ __launch_bounds__(256) void read_sum(const f32x4* __restrict__ p, size_t n4, float* out) {
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const size_t nw = (size_t)gridDim.x * 4;
  const int lane = threadIdx.x & 63;
  float acc = 0.f;
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
  if (acc == 123.456f) out[0] = acc;
}
static hipStream_t s;
static hipEvent_t e0, e1;
static float run(int nt, const void* p, size_t bytes, float* out) {
  hipEventRecord(e0, s);
  if (nt) hipLaunchKernelGGL(read_sum<1>, dim3(2048), dim3(256), 0, s, (const f32x4*)p, bytes / 16, out);
  else hipLaunchKernelGGL(read_sum<0>, dim3(2048), dim3(256), 0, s, (const f32x4*)p, bytes / 16, out);
  hipEventRecord(e1, s);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}
int main() {
  CK(hipStreamCreate(&s)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t BIG = (size_t)4700 << 20;
  char *big, *a; float* out;
  CK(hipMalloc(&big, BIG)); CK(hipMalloc(&a, (size_t)256 << 20)); CK(hipMalloc(&out, 4096));
  CK(hipMemset(big, 1, BIG)); CK(hipMemset(a, 1, (size_t)256 << 20)); CK(hipDeviceSynchronize());
  printf("# A = subset read with plain loads; B = 4.7 GB stream between two reads of A.  GB/s of A per round (round 1 = cold)\n");
  for (int bnt = 1; bnt >= 0; --bnt)
    for (size_t mb : {16, 32, 64, 128, 192}) {
      const size_t S = mb << 20;
      run(0, big, BIG, out);  // evict
      printf("A %3zu MB, B read with %s loads:", mb, bnt ? "nt   " : "plain");
      for (int r = 0; r < 5; ++r) {
        const float ta = run(0, a, S, out);
        const float tb = run(bnt, big, BIG, out);
        printf("  %7.0f (B %5.0f)", S / ta / 1e6, BIG / tb / 1e6);
      }
      printf("\n");
    }
  // control: A re-read back to back (nothing in between)
  for (size_t mb : {16, 32, 64, 128, 192}) {
    const size_t S = mb << 20;
    run(0, big, BIG, out);
    printf("A %3zu MB, back to back:", mb);
    for (int r = 0; r < 4; ++r) printf("  %7.0f", S / run(0, a, S, out) / 1e6);
    printf("\n");
  }
  return 0;
}
