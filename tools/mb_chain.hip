// mb_chain.hip — microbenchmarks behind the round-2 design decisions (run on the GPU box):
//   Q1  does a slab read by kernel A stay in the XCD L2s / the Infinity Cache for kernel B?
//   Q2  slope / intercept of a hipGraph chain of dependent streaming kernels (what one hop costs)
//   Q3  prefetching the wo slab on a parallel graph branch while a latency-bound "attention"
//       kernel runs: does the consumer get faster?
//   Q4  VALU issue rates that price the int8 dequant inner loop (v_fma_f32, v_pk_fma_f32,
//       v_cvt_f32_ubyteN, sdwa sign-extending convert)
//   hipcc --offload-arch=gfx950 -O3 tools/mb_chain.hip -o kuiperllama_amd/lib/mb_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// GEMV-shaped streaming read: one wave = one 2*U KiB tile (two "rows" of U KiB), tiles handed out
// wave-strided like kh_gemv.h::gemv_pairs.  dep: a word written by the predecessor (read first).
template <int NT, int U>
__global__ __launch_bounds__(512) void k_stream(const f32x4* __restrict__ p, size_t n4, const float* dep, float* out) {
  const int wpb = blockDim.x >> 6;
  const size_t wave = (size_t)blockIdx.x * wpb + (threadIdx.x >> 6);
  const size_t nw = (size_t)gridDim.x * wpb;
  const int lane = threadIdx.x & 63;
  const float d = dep ? dep[threadIdx.x & 63] : 0.f;
  float acc = 0.f;
  const size_t tile = (size_t)2 * U * 64;  // float4 per tile
  for (size_t t = wave; t * tile < n4; t += nw) {
    f32x4 v[2 * U];
#pragma unroll
    for (int u = 0; u < 2 * U; ++u) {
      size_t idx = t * tile + (size_t)u * 64 + lane;
      if (idx >= n4) idx = 0;
      v[u] = NT ? __builtin_nontemporal_load(p + idx) : p[idx];
    }
#pragma unroll
    for (int u = 0; u < 2 * U; ++u) acc += v[u].x * d + v[u].y + v[u].z + v[u].w;
  }
  acc += __shfl_xor(acc, 1);
  if (lane == 0 && (acc == 123.456f || wave == 0)) out[wave & 63] = acc;  // tiny dependent output
}

// latency-bound stand-in for decode attention: 32 workgroups, a chain of `hops` dependent loads
__global__ __launch_bounds__(512) void k_latency(const int* __restrict__ chase, int hops, const float* dep, float* out) {
  int i = (int)(dep[0] * 0.f) + blockIdx.x * 64 + (threadIdx.x & 63);
  for (int h = 0; h < hops; ++h) i = __builtin_nontemporal_load(chase + i);
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x & 63] = (float)i;
}

template <int MODE>  // 0 v_fma_f32, 1 v_pk_fma_f32, 2 cvt_ubyte + fma, 3 sext byte cvt + fma
__global__ __launch_bounds__(256) void k_alu(float* out, int iters, unsigned seed) {
  float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f;
  f32x2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a1, a0}, p3 = {a3, a2};
  const f32x2 m = {1.0001f, 0.9999f}, c = {1e-6f, -1e-6f};
  unsigned w = seed + threadIdx.x;
  for (int i = 0; i < iters; ++i) {
    if (MODE == 0) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        a0 = __builtin_fmaf(a0, 1.0001f, 1e-6f); a1 = __builtin_fmaf(a1, 0.9999f, 1e-6f);
        a2 = __builtin_fmaf(a2, 1.0001f, -1e-6f); a3 = __builtin_fmaf(a3, 0.9999f, -1e-6f);
      }
    } else if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c);
        p2 = __builtin_elementwise_fma(p2, m, c); p3 = __builtin_elementwise_fma(p3, m, c);
      }
    } else if (MODE == 2) {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        a0 = __builtin_fmaf((float)(w & 0xff), a1, a0); a1 = __builtin_fmaf((float)((w >> 8) & 0xff), a2, a1);
        a2 = __builtin_fmaf((float)((w >> 16) & 0xff), a3, a2); a3 = __builtin_fmaf((float)(w >> 24), a0, a3);
        w = w * 1664525u + 1013904223u;
      }
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) {
        const int s = (int)w;
        a0 = __builtin_fmaf((float)(signed char)(s & 0xff), a1, a0); a1 = __builtin_fmaf((float)(signed char)((s >> 8) & 0xff), a2, a1);
        a2 = __builtin_fmaf((float)(signed char)((s >> 16) & 0xff), a3, a2); a3 = __builtin_fmaf((float)(s >> 24), a0, a3);
        w = w * 1664525u + 1013904223u;
      }
    }
  }
  out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y;
}

static hipStream_t S, S2;
static hipEvent_t E0, E1;

template <class F>
static float time_graph(F&& body, int replays = 5) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
  body();
  CK(hipStreamEndCapture(S, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, S));
  CK(hipStreamSynchronize(S));
  float best = 1e30f;
  for (int r = 0; r < replays; ++r) {
    CK(hipEventRecord(E0, S)); CK(hipGraphLaunch(ge, S)); CK(hipEventRecord(E1, S)); CK(hipEventSynchronize(E1));
    float ms; CK(hipEventElapsedTime(&ms, E0, E1));
    if (ms < best) best = ms;
  }
  CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1e3f;  // us
}

static void stream(int nt, int u, const char* p, size_t bytes, const float* dep, float* out, int grid, int wg, hipStream_t s) {
  const f32x4* q = (const f32x4*)p;
#define L(NT, UU) hipLaunchKernelGGL((k_stream<NT, UU>), dim3(grid), dim3(wg), 0, s, q, bytes / 16, dep, out)
  if (nt) { if (u == 8) L(1, 8); else if (u == 4) L(1, 4); else L(1, 2); }
  else { if (u == 8) L(0, 8); else if (u == 4) L(0, 4); else L(0, 2); }
#undef L
}

int main(int argc, char** argv) {
  CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&S2, hipStreamNonBlocking));
  CK(hipEventCreate(&E0)); CK(hipEventCreate(&E1));
  const size_t BUF = (size_t)6 << 30;
  char* buf; float* out; int* chase;
  CK(hipMalloc(&buf, BUF)); CK(hipMalloc(&out, 1 << 16)); CK(hipMalloc(&chase, 1 << 22));
  CK(hipMemset(buf, 1, BUF)); CK(hipMemset(out, 0, 1 << 16));
  {
    std::vector<int> h(1 << 20);
    for (int i = 0; i < (1 << 20); ++i) h[i] = (int)(((size_t)i * 40503u + 12345u) & ((1 << 20) - 1));
    CK(hipMemcpy(chase, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  }
  CK(hipDeviceSynchronize());

  // ---- Q2: chain of N dependent streaming kernels over distinct slabs -------------------------
  printf("## Q2 chain of 16 dependent streaming kernels (distinct slabs), us per kernel\n");
  for (int wg : {256, 512}) for (int grid : {256, 512, 1024, 2048}) for (int u : {4, 8}) {
    if ((long)grid * wg > 1024L * 512) continue;
    printf("wg %d grid %4d u %d nt :", wg, grid, u);
    float t[6]; int k = 0; const size_t mbs[] = {4, 16, 32, 64, 128, 256};
    for (size_t mb : mbs) {
      const size_t sz = mb << 20;
      float us = time_graph([&] { for (int i = 0; i < 16; ++i) stream(1, u, buf + (size_t)i * sz, sz, out, out, grid, wg, S); });
      t[k++] = us / 16;
      printf("  %zuMB %.2f", mb, us / 16);
    }
    const float slope = (t[5] - t[3]) / 192.f;  // us per MB between 64 and 256 MB
    printf("   | slope %.2f TB/s, intercept(64MB) %.2f us\n", 1.048576f / slope, t[3] - slope * 64);
  }

  // ---- Q1: L2 / MALL retention across a kernel boundary ----------------------------------------
  printf("## Q1 consumer (nt, grid 1024x256) of a slab right after a producer pass; us\n");
  for (size_t mb : {4, 8, 16, 24, 32, 64}) {
    const size_t sz = mb << 20;
    char* slab = buf + ((size_t)5 << 30);
    auto flush = [&] { stream(0, 8, buf, (size_t)2 << 30, nullptr, out, 2048, 256, S); };
    // cold
    flush(); CK(hipStreamSynchronize(S));
    float cold = time_graph([&] { stream(1, 8, slab, sz, out, out, 1024, 256, S); }, 1);
    // after plain producer with the same block->address map, timed alone
    float warm_same = 1e30f, warm_other = 1e30f, warm_nt = 1e30f;
    for (int r = 0; r < 3; ++r) {
      flush(); stream(0, 8, slab, sz, out, out, 1024, 256, S); CK(hipStreamSynchronize(S));
      float a = time_graph([&] { stream(1, 8, slab, sz, out, out, 1024, 256, S); }, 1); if (a < warm_same) warm_same = a;
      flush(); stream(0, 8, slab, sz, out, out, 1000, 256, S); CK(hipStreamSynchronize(S));  // different map
      a = time_graph([&] { stream(1, 8, slab, sz, out, out, 1024, 256, S); }, 1); if (a < warm_other) warm_other = a;
      flush(); stream(1, 8, slab, sz, out, out, 1024, 256, S); CK(hipStreamSynchronize(S));  // nt producer
      a = time_graph([&] { stream(1, 8, slab, sz, out, out, 1024, 256, S); }, 1); if (a < warm_nt) warm_nt = a;
    }
    printf("slab %3zu MB: cold %.2f | after plain pass, same map %.2f | other map %.2f | after nt pass %.2f\n", mb, cold, warm_same, warm_other, warm_nt);
  }
  printf("(time_graph replays the consumer once after a warm-up launch: 'cold' is itself second-touch; see Q3 for the in-graph numbers)\n");

  // ---- Q3: layer-shaped chain, wo slab prefetched beside the latency kernel --------------------
  printf("## Q3 layer chain  qkv(25MB) -> latency kernel -> wo(16.8MB) -> ffn13(134MB) -> w2(67MB), 16 layers, us per layer\n");
  {
    const size_t q = 25u << 20, wo = (size_t)(16.8 * 1048576) & ~(size_t)1023, f = 134u << 20, w2 = 67u << 20;
    const size_t per = q + wo + f + w2;
    hipEvent_t fork, join; CK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    for (int hops : {2, 3}) for (int mode = 0; mode < 4; ++mode) {
      // mode 0: no prefetch; 1: plain prefetch of wo beside the latency kernel (same map as consumer);
      // 2: same with nt prefetch; 3: prefetch with a different block map (MALL only)
      float us = time_graph([&] {
        for (int l = 0; l < 16; ++l) {
          char* base = buf + (size_t)l * per;
          stream(1, 4, base, q, out, out, 512, 256, S);
          if (mode) {
            CK(hipEventRecord(fork, S)); CK(hipStreamWaitEvent(S2, fork, 0));
            stream(mode == 2, 4, base + q, wo, nullptr, out + 4096, mode == 3 ? 500 : 512, 256, S2);
            CK(hipEventRecord(join, S2));
          }
          hipLaunchKernelGGL(k_latency, dim3(32), dim3(512), 0, S, chase, hops, out, out);
          if (mode) CK(hipStreamWaitEvent(S, join, 0));
          stream(1, 4, base + q, wo, out, out, 512, 256, S);
          stream(1, 8, base + q + wo, f, out, out, 1024, 256, S);
          stream(1, 8, base + q + wo + f, w2, out, out, 1024, 512, S);
        }
      });
      printf("latency hops %d mode %d: %.2f us/layer\n", hops, mode, us / 16);
    }
    // the pieces alone, chained over the 16 layers
    float tq = time_graph([&] { for (int l = 0; l < 16; ++l) stream(1, 4, buf + (size_t)l * per, q, out, out, 512, 256, S); }) / 16;
    float tl2 = time_graph([&] { for (int l = 0; l < 16; ++l) hipLaunchKernelGGL(k_latency, dim3(32), dim3(512), 0, S, chase, 2, out, out); }) / 16;
    float tl3 = time_graph([&] { for (int l = 0; l < 16; ++l) hipLaunchKernelGGL(k_latency, dim3(32), dim3(512), 0, S, chase, 3, out, out); }) / 16;
    float two = time_graph([&] { for (int l = 0; l < 16; ++l) stream(1, 4, buf + (size_t)l * per + q, wo, out, out, 512, 256, S); }) / 16;
    printf("alone: qkv %.2f  latency2 %.2f  latency3 %.2f  wo %.2f us\n", tq, tl2, tl3, two);
  }

  // ---- Q4: VALU rates ------------------------------------------------------------------------------
  printf("## Q4 VALU: ns per 64 inner ops per wave-lane-set (grid 1024x256 = 16 waves/CU), lower = faster\n");
  for (int mode = 0; mode < 4; ++mode) {
    const int iters = 4096;
    auto go = [&] {
      if (mode == 0) hipLaunchKernelGGL(k_alu<0>, dim3(1024), dim3(256), 0, S, out, iters, 7u);
      else if (mode == 1) hipLaunchKernelGGL(k_alu<1>, dim3(1024), dim3(256), 0, S, out, iters, 7u);
      else if (mode == 2) hipLaunchKernelGGL(k_alu<2>, dim3(1024), dim3(256), 0, S, out, iters, 7u);
      else hipLaunchKernelGGL(k_alu<3>, dim3(1024), dim3(256), 0, S, out, iters, 7u);
    };
    float us = time_graph(go);
    // per SIMD: 4 waves, each iters*64 "slots" (fma: 64 v_fma; pk: 64 v_pk_fma = 128 fma; cvt modes: 64 cvt+64 fma + lcg)
    const double slots = (double)iters * 64 * 4;  // wave-instruction slots per SIMD (nominal)
    printf("mode %d (%s): %.1f us -> %.2f cycles@2.4GHz per nominal slot per SIMD\n", mode,
           mode == 0 ? "v_fma_f32" : mode == 1 ? "v_pk_fma_f32 (2 fma/slot)" : mode == 2 ? "cvt_ubyte+fma (2 ops/slot)" : "sext-byte cvt+fma",
           us, us * 1e-6 * 2.4e9 / slots);
  }
  return 0;
}
