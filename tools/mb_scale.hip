// mb_scale.hip — what do the int8 scale loads cost in a pure streaming kernel?  One wave streams
// "row pairs" of 2 x 4 KiB int8 (U = 4 dwordx4 per row per lane, like k_ffn13<true,4,..> on dim 4096)
// from a 96 MB slab per launch, 32 launches over distinct slabs in one graph:
//   mode 0  weights only
//   mode 1  + one dword scale load per dwordx4 (4 lanes share an address)      [shipped layout]
//   mode 2  + ONE coalesced dword scale load per row per chunk (lane l <- scale[pi(l)])
//   mode 3  + scale loads of mode 1 issued by one lane per quad only (exec-masked)
//   hipcc --offload-arch=gfx950 -O3 tools/mb_scale.hip -o kuiperllama_amd/lib/mb_scale
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const i32x4* __restrict__ w, const float* __restrict__ sc, int pairs, float* out) {
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nw = gridDim.x * 4, lane = threadIdx.x & 63;
  float acc = 0.f;
  for (int p = wave; p < pairs; p += nw) {
    const i32x4* r0 = w + (size_t)(2 * p) * 256;      // row = 4096 B = 256 x 16 B
    const i32x4* r1 = r0 + 256;
    const float* s0 = sc + (size_t)(2 * p) * 64;      // 64 groups per row
    const float* s1 = s0 + 64;
    i32x4 q0[4], q1[4];
    float g0[4], g1[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { q0[u] = __builtin_nontemporal_load(r0 + u * 64 + lane); q1[u] = __builtin_nontemporal_load(r1 + u * 64 + lane); }
    if (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { g0[u] = s0[u * 16 + (lane >> 2)]; g1[u] = s1[u * 16 + (lane >> 2)]; }
    } else if (MODE == 2) {
      const int pi = (lane & 3) * 16 + (lane >> 2);
      g0[0] = s0[pi]; g1[0] = s1[pi];
      g0[1] = g0[2] = g0[3] = g0[0]; g1[1] = g1[2] = g1[3] = g1[0];
    } else if (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u) { g0[u] = 0.f; g1[u] = 0.f; }
      if ((lane & 3) == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { g0[u] = s0[u * 16 + (lane >> 2)]; g1[u] = s1[u * 16 + (lane >> 2)]; }
      }
    } else {
#pragma unroll
      for (int u = 0; u < 4; ++u) g0[u] = g1[u] = 1.f;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)
      acc += g0[u] * (float)(q0[u].x ^ q0[u].y ^ q0[u].z ^ q0[u].w) + g1[u] * (float)(q1[u].x ^ q1[u].y ^ q1[u].z ^ q1[u].w);
  }
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  hipStream_t S; CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int pairs = 11008;                       // ffn13 of Llama-2-7B: 11008 (w1,w3) row pairs
  const size_t slab = (size_t)pairs * 2 * 4096, sslab = (size_t)pairs * 2 * 64 * 4;
  const int NL = 32;
  char* w; float* sc; float* out;
  CK(hipMalloc(&w, slab * NL)); CK(hipMalloc(&sc, sslab * NL)); CK(hipMalloc(&out, 64));
  CK(hipMemset(w, 1, slab * NL)); CK(hipMemset(sc, 0, sslab * NL)); CK(hipDeviceSynchronize());
  for (int grid : {512, 768, 1024}) for (int mode = 0; mode < 4; ++mode) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < NL; ++l) {
      const i32x4* wp = (const i32x4*)(w + slab * l); const float* sp = (const float*)((char*)sc + sslab * l);
      if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, S, wp, sp, pairs, out);
      else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, S, wp, sp, pairs, out);
      else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, S, wp, sp, pairs, out);
      else hipLaunchKernelGGL(k<3>, dim3(grid), dim3(256), 0, S, wp, sp, pairs, out);
    }
    CK(hipStreamEndCapture(S, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, S)); CK(hipStreamSynchronize(S));
    float best = 1e9f;
    for (int r = 0; r < 5; ++r) {
      CK(hipEventRecord(e0, S)); CK(hipGraphLaunch(ge, S)); CK(hipEventRecord(e1, S)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    const double us = best * 1e3 / NL, bytes = (double)slab + (mode ? (double)sslab : 0.0);
    printf("grid %4d mode %d: %.2f us per launch, %.2f TB/s of (weights%s)\n", grid, mode, us, bytes / us / 1e6, mode ? "+scales" : "");
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
