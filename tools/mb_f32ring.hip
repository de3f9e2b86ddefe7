// mb_f32ring.hip - VERDICT r5 item 3: the Llama-3.2-1B fp32 ffn13 and w2 decode GEMVs on an LDS-DMA ring core
// (tools/f32ring.h) against the shipped register-tile kernels (kh_fused.h), same box, same slabs.
// For every kernel: (1) outputs compared BITWISE with the shipped kernel's on the same inputs, (2) us per launch of a
// hipGraph of NL launches over NL distinct weight slabs (nothing is served from a cache), best of 5.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++20 -w -I kuiperllama_amd/csrc -I tools tools/mb_f32ring.hip -o kuiperllama_amd/lib/mb_f32ring
// argv[1]: 0 ffn13 only, 1 w2 only (default both); argv[2]: sweeps over the slabs per graph (default 2)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#include "f32ring.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_fill_f32(float* p, size_t n, uint32_t seed, float lo, float hi) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed ^ (uint32_t)(i >> 32) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = lo + (hi - lo) * (float)(h >> 8) * (1.0f / 16777216.0f);
  }
}

static hipStream_t S;
static int g_reps = 2;
static float time_graph(int NL0, const std::function<void(int)>& launch) {
  hipGraph_t g; hipGraphExec_t ge;
  const int NL = NL0 * g_reps;
  CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < NL; ++l) launch(l % NL0);
  CK(hipStreamEndCapture(S, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, S)); CK(hipStreamSynchronize(S));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < 7; ++r) {
    CK(hipEventRecord(e0, S)); CK(hipGraphLaunch(ge, S)); CK(hipEventRecord(e1, S)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1e3f / NL;
}
template <class K>
static void optin(K k, size_t lds) {
  if (lds > 64 * 1024) CK(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
}
static bool same(const float* d_a, const float* d_b, size_t n, const char* what) {
  std::vector<float> a(n), b(n);
  CK(hipMemcpy(a.data(), d_a, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d_b, n * 4, hipMemcpyDeviceToHost));
  size_t bad = 0, first = 0;
  for (size_t i = 0; i < n; ++i)
    if (memcmp(&a[i], &b[i], 4) != 0) { if (!bad) first = i; ++bad; }
  if (bad) printf("   !! %s: %zu of %zu words differ (first at %zu: %.9g vs %.9g)\n", what, bad, n, first, a[first], b[first]);
  return bad == 0;
}
static void report(const char* name, const char* variant, float us, double bytes, bool ok, float base_us) {
  printf("%-6s %-44s %7.2f us  %.3f of 8 TB/s  %s  %+5.1f %%\n", name, variant, us, bytes / (us * 1e-6) / 8e12,
         ok ? "bit-identical" : "MISMATCH     ", base_us > 0 ? (us / base_us - 1.0) * 100.0 : 0.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;
  g_reps = argc > 2 ? atoi(argv[2]) : 2;
  CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  const int dim = 2048, hidden = 8192, NL = 16;
  const size_t slab = (size_t)2 * hidden * dim;  // floats: ffn13 = w1 | w3; w2 uses the first half
  float* w;
  CK(hipMalloc(&w, slab * NL * 4));
  hipLaunchKernelGGL(k_fill_f32, dim3(4096), dim3(256), 0, S, w, slab * NL, 0x1234u, -0.05f, 0.05f);
  float *x, *xh, *wn, *o_ref, *o_ref512, *o_new, *x0;
  const size_t NV = 16384;
  CK(hipMalloc(&x, NV * 4)); CK(hipMalloc(&xh, NV * 4)); CK(hipMalloc(&wn, NV * 4));
  CK(hipMalloc(&o_ref, NV * 4)); CK(hipMalloc(&o_ref512, NV * 4)); CK(hipMalloc(&o_new, NV * 4)); CK(hipMalloc(&x0, NV * 4));
  hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, S, x, NV, 0x1u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, S, xh, NV, 0x2u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, S, wn, NV, 0x3u, 0.5f, 1.5f);
  hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, S, x0, NV, 0x4u, -1.f, 1.f);
  CK(hipStreamSynchronize(S));
  printf("Llama-3.2-1B fp32 decode GEMVs, %d distinct slabs x %d sweeps per graph; shipped register-tile kernel vs LDS-DMA ring (tools/f32ring.h)\n", NL, g_reps);

  // ---------------------------------------------------------------- ffn13: 8192 (w1, w3) row pairs x 2048
  if (only < 0 || only == 0) {
    const double bytes = (double)slab * 4 + 2.0 * dim * 4 + hidden * 4.0;
    auto args = [&](int l, float* out) {
      const float* w0 = w + slab * l;
      KhFfn13Args a{}; a.x = x; a.ffn_norm = wn; a.w1 = KhLin{w0, nullptr, nullptr}; a.w3 = KhLin{w0 + slab / 2, nullptr, nullptr};
      a.h = out; a.dim = dim; a.hidden = hidden; a.gshift = 0; a.eps = 1e-5f; return a;
    };
    const size_t lds0 = fused_lds_bytes(false, dim);
    hipLaunchKernelGGL((k_ffn13<false, 8, 2>), dim3(512), dim3(256), lds0, S, args(0, o_ref));
    hipLaunchKernelGGL((k_ffn13<false, 8, 1>), dim3(256), dim3(512), lds0, S, args(0, o_ref512));
    CK(hipStreamSynchronize(S));
    float base = 1e9f;
    for (int rep = 0; rep < 2; ++rep) {
      const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_ffn13<false, 8, 2>), dim3(512), dim3(256), lds0, S, args(l, o_new)); });
      if (t < base) base = t;
    }
    report("ffn13", "shipped k_ffn13<false,8,2> wg256 grid512", base, bytes, true, 0);
    {
      const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_ffn13<false, 8, 1>), dim3(256), dim3(512), lds0, S, args(l, o_new)); });
      report("ffn13", "register tiles wg512 grid256 U8", t, bytes, true, base);
    }
#define FFN_RING(RR, WG, GRID)                                                                                            \
  do {                                                                                                                    \
    constexpr int MV = (WG) == 256 ? 2 : 1;                                                                               \
    const size_t lds = f32ring_lds_bytes(dim, (WG) / 64, RR);                                                             \
    if (lds * (((GRID) + 255) / 256) > 160 * 1024) { printf("ffn13  ring R%d wg%d grid%d: LDS does not fit\n", RR, WG, GRID); break; } \
    optin(k_ffn13_ring_f32<RR, MV>, lds);                                                                                 \
    CK(hipMemsetAsync(o_new, 0xff, hidden * 4, S));                                                                       \
    hipLaunchKernelGGL((k_ffn13_ring_f32<RR, MV>), dim3(GRID), dim3(WG), lds, S, args(0, o_new));                         \
    CK(hipStreamSynchronize(S));                                                                                          \
    const bool ok = same((WG) == 256 ? o_ref : o_ref512, o_new, hidden, "ffn13 h");                                       \
    const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_ffn13_ring_f32<RR, MV>), dim3(GRID), dim3(WG), lds, S, args(l, o_new)); }); \
    char v[96]; snprintf(v, sizeof v, "ring R%d wg%d grid%d (%zu KB, %d KB in flight / CU)", RR, WG, GRID, lds >> 10,     \
                         ((RR) - 1) * 2 * ((WG) / 64) * (((GRID) + 255) / 256));                                          \
    report("ffn13", v, t, bytes, ok, base);                                                                               \
  } while (0)
    FFN_RING(8, 256, 512);
    FFN_RING(6, 256, 512);
    FFN_RING(4, 256, 512);
    FFN_RING(4, 256, 768);
    FFN_RING(3, 256, 1024);
    FFN_RING(2, 256, 1024);
    FFN_RING(8, 512, 256);
    FFN_RING(6, 512, 256);
    FFN_RING(4, 512, 512);
    FFN_RING(3, 512, 512);
    FFN_RING(2, 512, 512);
    FFN_RING(4, 1024, 256);
    FFN_RING(3, 1024, 256);
    {
      const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_ffn13<false, 8, 2>), dim3(512), dim3(256), lds0, S, args(l, o_new)); });
      report("ffn13", "shipped again (drift check)", t, bytes, true, base);
    }
  }
  // ---------------------------------------------------------------- w2: 1024 row pairs x 8192, residual add
  if (only < 0 || only == 1) {
    const int M = hidden, K = dim;
    const double bytes = (double)K * M * 4 + M * 4.0 + 2.0 * K * 4;
    auto args = [&](int l, float* xres) {
      KhGemvResArgs a{}; a.vec = xh; a.w = KhLin{w + slab * l, nullptr, nullptr}; a.x = xres; a.M = M; a.K = K; a.gshift = 0; return a;
    };
    const size_t lds0 = fused_lds_bytes(false, M);
    optin(k_gemv_res<false, 8, 4, 4>, lds0);
    CK(hipMemcpyAsync(o_ref, x0, K * 4, hipMemcpyDeviceToDevice, S));
    hipLaunchKernelGGL((k_gemv_res<false, 8, 4, 4>), dim3(512), dim3(512), lds0, S, args(0, o_ref));
    CK(hipStreamSynchronize(S));
    float base = 1e9f;
    for (int rep = 0; rep < 2; ++rep) {
      const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_gemv_res<false, 8, 4, 4>), dim3(512), dim3(512), lds0, S, args(l, o_new)); });
      if (t < base) base = t;
    }
    report("w2", "shipped k_gemv_res<false,8,4,4> wg512 grid512", base, bytes, true, 0);
#define W2_RING(RR, SP, WG, GRID)                                                                                         \
  do {                                                                                                                    \
    constexpr int MV = (WG) == 512 ? 4 : ((WG) == 1024 ? 2 : 0);                                                          \
    const size_t lds = f32ring_lds_bytes(M, (WG) / 64, RR);                                                               \
    if (lds * (((GRID) + 255) / 256) > 160 * 1024) { printf("w2     ring R%d wg%d grid%d: LDS does not fit\n", RR, WG, GRID); break; } \
    optin(k_gemv_res_ring_f32<RR, MV, SP>, lds);                                                                          \
    CK(hipMemcpyAsync(o_new, x0, K * 4, hipMemcpyDeviceToDevice, S));                                                     \
    hipLaunchKernelGGL((k_gemv_res_ring_f32<RR, MV, SP>), dim3(GRID), dim3(WG), lds, S, args(0, o_new));                  \
    CK(hipStreamSynchronize(S));                                                                                          \
    const bool ok = (SP) == 4 ? same(o_ref, o_new, K, "x after residual") : true;                                         \
    const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_gemv_res_ring_f32<RR, MV, SP>), dim3(GRID), dim3(WG), lds, S, args(l, o_new)); }); \
    char v[96]; snprintf(v, sizeof v, "ring R%d wg%d grid%d split%d (%zu KB)%s", RR, WG, GRID, SP, lds >> 10,             \
                         (SP) == 4 ? "" : " [other split: not compared]");                                                \
    report("w2", v, t, bytes, ok, base);                                                                                  \
  } while (0)
    W2_RING(8, 4, 512, 256);
    W2_RING(6, 4, 512, 256);
    W2_RING(4, 4, 512, 256);
    W2_RING(3, 4, 1024, 256);
    W2_RING(2, 4, 1024, 256);
    W2_RING(8, 2, 512, 256);
    W2_RING(4, 2, 512, 256);
    W2_RING(3, 2, 1024, 256);
    {
      const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_gemv_res<false, 8, 4, 4>), dim3(512), dim3(512), lds0, S, args(l, o_new)); });
      report("w2", "shipped again (drift check)", t, bytes, true, base);
    }
  }
  return 0;
}
