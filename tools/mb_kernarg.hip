// mb_kernarg.hip — what does the first `s_load kernarg ; s_waitcnt lgkmcnt(0)` of a short kernel cost, and does the
// gfx950 kernel-argument PRELOAD (arguments delivered in user SGPRs at wave launch,
// -mllvm -amdgpu-kernarg-preload-count=N) remove it?
//
// The decode step is 82 short launches per token; each begins by fetching its arguments through the scalar cache
// before it can form its first address.  A by-value struct argument (what every fused kernel takes) cannot be
// preloaded; individual scalar / pointer arguments can (up to 16 SGPRs).  Two kernels with the same body - one dependent
// load, one add, one store per thread, the shape of the latency-bound attention launch - chained in a graph:
//     k_struct(Args a)                                    arguments through s_load
//     k_scalar(const int* in, int* out, int a0 .. a5)     arguments preloaded (when the flag is given)
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-kernarg-preload-count=16 tools/mb_kernarg.hip \
//         -o kuiperllama_amd/lib/mb_kernarg
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

struct Args {
  const int* in;
  int* out;
  int a0, a1, a2, a3, a4, a5;
};

__global__ __launch_bounds__(256) void k_struct(const Args a) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  a.out[i] = a.in[i] + a.a0 + a.a1 + a.a2 + a.a3 + a.a4 + a.a5;
}
__global__ __launch_bounds__(256) void k_scalar(const int* in, int* out, int a0, int a1, int a2, int a3, int a4,
                                                int a5) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  out[i] = in[i] + a0 + a1 + a2 + a3 + a4 + a5;
}
// no arguments needed before the first memory access: the floor of a launch of this shape
__global__ __launch_bounds__(256) void k_empty() {}

int main(int argc, char** argv) {
  const int chain = argc > 1 ? atoi(argv[1]) : 400;
  const int reps = argc > 2 ? atoi(argv[2]) : 20;
  const int grid = argc > 3 ? atoi(argv[3]) : 256;
  const int n = grid * 256;
  int *a, *b;
  CK(hipMalloc(&a, n * 4));
  CK(hipMalloc(&b, n * 4));
  CK(hipMemset(a, 0, n * 4));
  CK(hipMemset(b, 0, n * 4));
  hipStream_t st;
  CK(hipStreamCreate(&st));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  auto run = [&](const char* name, int which) {
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int k = 0; k < chain; ++k) {
      const int* in = (k & 1) ? b : a;
      int* out = (k & 1) ? a : b;
      if (which == 0) {
        Args x{in, out, 1, 2, 3, 4, 5, k};
        hipLaunchKernelGGL(k_struct, dim3(grid), dim3(256), 0, st, x);
      } else if (which == 1) {
        hipLaunchKernelGGL(k_scalar, dim3(grid), dim3(256), 0, st, in, out, 1, 2, 3, 4, 5, k);
      } else {
        hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st);
      }
    }
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st));
    CK(hipStreamSynchronize(st));
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, st));
      CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%-34s grid %4d: %7.3f us per launch (chain of %d, best of %d)\n", name, grid, best * 1e3 / chain, chain,
           reps);
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(g));
  };
  for (int round = 0; round < 2; ++round) {
    run("k_empty (no arguments)", 2);
    run("k_struct (by-value struct, s_load)", 0);
    run("k_scalar (scalar args, preload)", 1);
  }
  return 0;
}
