// mb_q8ring.hip — same-box A/B of the int8 decode GEMVs: the shipped register-tile kernels (kh_fused.h) against the
// LDS-DMA ring kernels (kh_fused_ring.h) on the Llama-2-7B int8 shapes (group 64), random weights.
// For every kernel: (1) outputs compared BITWISE with the shipped kernel's on the same inputs, (2) us per launch of
// a hipGraph of NL launches over NL distinct weight slabs (nothing is served from a cache), best of 5.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++20 -I kuiperllama_amd/csrc -I tools tools/mb_q8ring.hip -o kuiperllama_amd/lib/mb_q8ring
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <algorithm>
#include <vector>

#include "ring_variants.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_fill_i8(int8_t* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed ^ (uint32_t)(i >> 32) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (int8_t)(h & 0xff);
  }
}
__global__ void k_fill_f32(float* p, size_t n, uint32_t seed, float lo, float hi) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = lo + (hi - lo) * (float)(h >> 8) * (1.0f / 16777216.0f);
  }
}

static hipStream_t S;
static int g_more = 0;  // argv[4]: also the work-distribution controls of the ffn13 section
static int g_reps = 1;  // sweeps over the NL slabs per graph (argv[3]): long graphs show the sustained clock
static float time_graph(int NL0, const std::function<void(int)>& launch) {
  hipGraph_t g; hipGraphExec_t ge;
  const int NL = NL0 * g_reps;
  CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < NL; ++l) launch(l % NL0);
  CK(hipStreamEndCapture(S, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, S)); CK(hipStreamSynchronize(S));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, S)); CK(hipGraphLaunch(ge, S)); CK(hipEventRecord(e1, S)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1e3f / NL;
}

// -DKH_TRACE: phase stamps of one launch per slab (kh_common.h::KH_STAMP): 0 entry, 1 vector staged, 2 first work item
// finished, 3 thread 0's wave done, 4 kernel end; 8 + w: wave w done.  Launches are NOT back to back here: every traced
// launch starts on an idle chip with cold weights.
#ifdef KH_TRACE
static unsigned long long* g_tr = nullptr;
static void trace_report(const char* name, const char* variant, int grid, int nslab, const std::function<void(int)>& launch) {
  double late[5] = {0}, med[5] = {0}, spread = 0, lastwave = 0, firstdone = 0, sub[3] = {0};
  std::vector<unsigned long long> h((size_t)grid * 32);
  for (int l = 0; l < nslab; ++l) {
    CK(hipMemsetAsync(g_tr, 0, (size_t)grid * 256, S));
    launch(l);
    CK(hipStreamSynchronize(S));
    CK(hipMemcpy(h.data(), g_tr, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull;
    for (int b = 0; b < grid; ++b) if (h[(size_t)b * 32]) t0 = std::min(t0, h[(size_t)b * 32]);
    std::vector<double> ph[5], sp, sb[3];
    double lt[5] = {0}, lw = 0, fd = 1e30;
    for (int b = 0; b < grid; ++b) {
      const unsigned long long* s = &h[(size_t)b * 32];
      if (!s[0]) continue;
      unsigned long long prev = s[0];
      for (int i = 0; i < 5; ++i) {
        if (!s[i]) continue;
        lt[i] = std::max(lt[i], (double)(s[i] - t0) * 0.01);
        if (i) ph[i].push_back((double)(s[i] - prev) * 0.01);
        prev = s[i];
      }
      if (s[5]) sb[0].push_back((double)(s[5] - s[0]) * 0.01);          // entry -> everything requested
      if (s[5] && s[6]) sb[1].push_back((double)(s[6] - s[5]) * 0.01);  // -> vector arrived
      if (s[6] && s[7]) sb[2].push_back((double)(s[7] - s[6]) * 0.01);  // -> block sum done
      unsigned long long lo = ~0ull, hi = 0;
      for (int w = 0; w < 8; ++w) if (s[8 + w]) { lo = std::min(lo, s[8 + w]); hi = std::max(hi, s[8 + w]); }
      if (hi) { sp.push_back((double)(hi - lo) * 0.01); lw = std::max(lw, (double)(hi - t0) * 0.01); fd = std::min(fd, (double)(lo - t0) * 0.01); }
    }
    for (int i = 0; i < 5; ++i) {
      late[i] += lt[i] / nslab;
      if (!ph[i].empty()) { std::sort(ph[i].begin(), ph[i].end()); med[i] += ph[i][ph[i].size() / 2] / nslab; }
    }
    if (!sp.empty()) { std::sort(sp.begin(), sp.end()); spread += sp[sp.size() / 2] / nslab; }
    for (int i = 0; i < 3; ++i)
      if (!sb[i].empty()) { std::sort(sb[i].begin(), sb[i].end()); sub[i] += sb[i][sb[i].size() / 2] / nslab; }
    lastwave += lw / nslab;
    firstdone += fd / nslab;
  }
  printf("   trace %-6s %-28s latest workgroup: entry %.2f | staged %.2f | item1 %.2f | wave0 done %.2f | end %.2f   "
         "median phases: stage %.2f | item1 %.2f | rest %.2f | tail %.2f   waves: first done %.2f, last done %.2f, spread inside a workgroup %.2f\n",
         name, variant, late[0], late[1], late[2], late[3], late[4], med[1], med[2], med[3], med[4], firstdone, lastwave, spread);
  if (sub[0] > 0)
    printf("         inside the stage (register-tile staging only): entry -> all requested %.2f | -> vector arrived %.2f | -> block sum %.2f\n",
           sub[0], sub[1], sub[2]);
  fflush(stdout);
}
#define TRACE(name, variant, grid, ...) trace_report(name, variant, grid, 8, __VA_ARGS__)
#else
#define TRACE(name, variant, grid, ...) do { } while (0)
#endif
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
  printf("%-6s %-34s %7.2f us  %.3f of 8 TB/s  %s  %+5.1f %%\n", name, variant, us, bytes / (us * 1e-6) / 8e12,
         ok ? "bit-identical" : "MISMATCH     ", base_us > 0 ? (us / base_us - 1.0) * 100.0 : 0.0);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const int only = argc > 1 ? atoi(argv[1]) : -1;  // 0 ffn13, 1 cls, 2 wo, 3 w2, 4 qkv
  g_reps = argc > 3 ? atoi(argv[3]) : 1;
  g_more = argc > 4 ? atoi(argv[4]) : 0;
  const int shift = argc > 2 ? atoi(argv[2]) : 0;  // bytes the weight / scale arrays sit off a 4-KiB boundary (a device image
                                                   // that still carries its 32-byte header puts every row 32 B off a line)
  CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
#ifdef KH_TRACE  // every kernel of this build flushes its stamps: the buffer exists before the first launch
  CK(hipMalloc(&g_tr, (size_t)4096 * 32 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(kh_trace_buf), &g_tr, sizeof(g_tr)));
#endif
  const int dim = 4096, hidden = 11008, vocab = 32000, NL = 32, gshift = 6;
  const size_t slab_max = (size_t)vocab * dim;  // cls is the largest matrix (131 MB); ffn13 = 2 * 11008 * 4096 = 90 MB
  int8_t* w; float* sc;
  CK(hipMalloc(&w, slab_max * NL + 4096)); CK(hipMalloc(&sc, slab_max / 64 * 4 * NL + 4096));
  w += shift; sc += shift / 4;
  hipLaunchKernelGGL(k_fill_i8, dim3(4096), dim3(256), 0, S, w, slab_max * NL, 0x1234u);
  hipLaunchKernelGGL(k_fill_f32, dim3(4096), dim3(256), 0, S, sc, slab_max / 64 * NL, 0x77u, 0.001f, 0.01f);
  float *x, *xh, *wn, *o_ref, *o_ref512, *o_new, *x0;
  const size_t NV = 32768;
  CK(hipMalloc(&x, NV * 4)); CK(hipMalloc(&xh, NV * 4)); CK(hipMalloc(&wn, NV * 4));
  CK(hipMalloc(&o_ref, NV * 4)); CK(hipMalloc(&o_ref512, NV * 4)); CK(hipMalloc(&o_new, NV * 4)); CK(hipMalloc(&x0, NV * 4));
  hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, S, x, NV, 0x1u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, S, xh, NV, 0x2u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, S, wn, NV, 0x3u, 0.5f, 1.5f);
  hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, S, x0, NV, 0x4u, -1.f, 1.f);
  float* pv; int32_t* pi;
  CK(hipMalloc(&pv, 4096 * 4)); CK(hipMalloc(&pi, 4096 * 4));
  CK(hipStreamSynchronize(S));
  printf("Llama-2-7B int8 (group 64) decode GEMVs, %d distinct slabs per graph; shipped register-tile kernel vs LDS-DMA ring; arrays %d B off alignment\n", NL, shift);

  // ---------------------------------------------------------------- ffn13
  if (only < 0 || only == 0) {
    const size_t wb = (size_t)2 * hidden * dim, sb = wb / 64;  // bytes / floats per slab
    const double bytes = (double)wb + sb * 4.0 + 2.0 * dim * 4 + hidden * 4.0;
    auto args = [&](int l, float* out) {
      const int8_t* w0 = w + wb * l; const float* s0 = sc + sb * l;
      KhFfn13Args a{}; a.x = x; a.ffn_norm = wn; a.w1 = KhLin{w0, s0, nullptr}; a.w3 = KhLin{w0 + wb / 2, s0 + sb / 2, nullptr};
      a.h = out; a.dim = dim; a.hidden = hidden; a.gshift = gshift; a.eps = 1e-5f; return a;
    };
    const size_t lds0 = fused_lds_bytes(true, dim);
    hipLaunchKernelGGL((k_ffn13<true, 4, 4>), dim3(512), dim3(256), lds0, S, args(0, o_ref));
    hipLaunchKernelGGL((k_ffn13<true, 4, 4>), dim3(256), dim3(512), lds0, S, args(0, o_ref512));  // the norm's block sum depends on the width
    CK(hipStreamSynchronize(S));
    const float base = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_ffn13<true, 4, 4>), dim3(512), dim3(256), lds0, S, args(l, o_new)); });
    report("ffn13", "shipped wg256 grid512 U4", base, bytes, true, 0);
    TRACE("ffn13", "shipped wg256 grid512", 512, [&](int l) { hipLaunchKernelGGL((k_ffn13<true, 4, 4>), dim3(512), dim3(256), lds0, S, args(l, o_new)); });
    {
      const size_t ldsr = ring_lds_bytes(dim, false, 4, 2);
      TRACE("ffn13", "ring R2 wg256 grid512 x:regs", 512, [&](int l) { hipLaunchKernelGGL((k_ffn13_ring<2, 4>), dim3(512), dim3(256), ldsr, S, args(l, o_new)); });
    }
#define FFN_RINGX(RR, MV, WG, GRID, BL, VT) FFN_RINGS(RR, MV, WG, GRID, BL, VT, 0)
#define FFN_RINGS(RR, MV, WG, GRID, BL, VT, STG)                                                                          \
  do {                                                                                                                    \
    const size_t lds = ring_lds_off(dim, (STG) == 0) + (size_t)((WG) / 64) * (RR) * KH_RING_SLOT;                         \
    if (lds * (((GRID) + 255) / 256) > 160 * 1024) { printf("ffn13  ring R%d wg%d grid%d: LDS does not fit\n", RR, WG, GRID); break; } \
    optin(k_ffn13_ring<RR, MV, BL, RingStager<true, MV, VT, STG>>, lds);                                                                        \
    CK(hipMemsetAsync(o_new, 0xff, hidden * 4, S));                                                                       \
    hipLaunchKernelGGL((k_ffn13_ring<RR, MV, BL, RingStager<true, MV, VT, STG>>), dim3(GRID), dim3(WG), lds, S, args(0, o_new));                \
    CK(hipStreamSynchronize(S));                                                                                          \
    const bool ok = same(((WG) == 256 || (VT) == 256) ? o_ref : o_ref512, o_new, hidden, "ffn13 h");                      \
    const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_ffn13_ring<RR, MV, BL, RingStager<true, MV, VT, STG>>), dim3(GRID), dim3(WG), lds, S, args(l, o_new)); }); \
    char v[96]; snprintf(v, sizeof v, "ring R%d wg%d grid%d%s %s (%zu KB)", RR, WG, GRID, (BL) ? " blocked" : "",         \
                         (STG) ? "x:regs" : "x:dma", lds >> 10);                                                          \
    report("ffn13", v, t, bytes, ok, base);                                                                               \
  } while (0)
    FFN_RINGS(2, 4, 256, 512, false, 0, 0);
    FFN_RINGS(3, 4, 256, 512, false, 0, 0);
    FFN_RINGS(2, 4, 256, 512, false, 0, 1);
    FFN_RINGS(3, 4, 256, 512, false, 0, 1);
    FFN_RINGS(4, 4, 256, 512, false, 0, 1);
    FFN_RINGS(2, 4, 256, 768, false, 0, 1);
    FFN_RINGS(3, 4, 256, 768, false, 0, 1);
    FFN_RINGS(2, 4, 256, 1024, false, 0, 1);
    FFN_RINGS(1, 4, 256, 1024, false, 0, 1);
    FFN_RINGS(2, 4, 512, 256, false, 0, 1);
    FFN_RINGS(2, 4, 512, 512, false, 0, 1);
    // [r6] four waves per CU: 11008 pairs over 1024 waves = 10 or 11 items each (2.3 % static imbalance instead of the
    // 11.6 % of 5 or 6 items at eight waves per CU, profiles/r6_mb_skew.txt), the bytes in flight made up by the ring depth
    FFN_RINGS(4, 4, 256, 256, false, 0, 1);
    FFN_RINGS(6, 4, 256, 256, false, 0, 1);
    FFN_RINGS(8, 4, 256, 256, false, 0, 1);
    FFN_RINGS(12, 4, 256, 256, false, 0, 1);
    FFN_RINGS(4, 4, 256, 256, true, 0, 1);
    FFN_RINGS(8, 4, 256, 256, true, 0, 1);
    if (g_more) {
      FFN_RINGX(2, 4, 256, 512, true, 0);     // control: blocked mapping alone, same 8 waves per CU (43 items -> 6 rounds)
      FFN_RINGX(2, 4, 704, 256, true, 256);   // 11 waves per CU: 43 items = 10 x 4 + 3
      FFN_RINGX(2, 4, 704, 256, false, 256);  // control: 11 waves, interleaved mapping
    }
  }
  // ---------------------------------------------------------------- cls
  if (only < 0 || only == 1) {
    const size_t wb = (size_t)vocab * dim, sb = wb / 64;
    const double bytes = (double)wb + sb * 4.0 + 2.0 * dim * 4 + vocab * 4.0;
    auto args = [&](int l, float* out) {
      KhClsArgs a{}; a.x = x; a.final_norm = wn; a.wcls = KhLin{w + wb * l, sc + sb * l, nullptr}; a.logits = out;
      a.part_val = pv; a.part_idx = pi; a.dim = dim; a.vocab = vocab; a.gshift = gshift; a.eps = 1e-5f; return a;
    };
    const size_t lds0 = cls_lds_bytes(true, dim);
    hipLaunchKernelGGL((k_cls<true, 4, 4>), dim3(512), dim3(256), lds0, S, args(0, o_ref));
    hipLaunchKernelGGL((k_cls<true, 4, 4>), dim3(256), dim3(512), lds0, S, args(0, o_ref512));
    CK(hipStreamSynchronize(S));
    const float base = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_cls<true, 4, 4>), dim3(512), dim3(256), lds0, S, args(l, o_new)); });
    report("cls", "shipped wg256 grid512 U4", base, bytes, true, 0);
    TRACE("cls", "shipped wg256 grid512", 512, [&](int l) { hipLaunchKernelGGL((k_cls<true, 4, 4>), dim3(512), dim3(256), lds0, S, args(l, o_new)); });
    {
      const size_t ldsr = ring_lds_bytes(dim, false, 4, 2);
      TRACE("cls", "ring R2 wg256 grid512 x:regs", 512, [&](int l) { hipLaunchKernelGGL((k_cls_ring<2, 4>), dim3(512), dim3(256), ldsr, S, args(l, o_new)); });
    }
#define CLS_RINGS(RR, MV, WG, GRID, STG)                                                                                  \
  do {                                                                                                                    \
    const size_t lds = ring_lds_off(dim, (STG) == 0) + (size_t)((WG) / 64) * (RR) * KH_RING_SLOT;                         \
    if (lds * (((GRID) + 255) / 256) > 160 * 1024) { printf("cls    ring R%d wg%d grid%d: LDS does not fit\n", RR, WG, GRID); break; } \
    optin(k_cls_ring<RR, MV, false, RingStager<true, MV, 0, STG>>, lds);                                                                        \
    CK(hipMemsetAsync(o_new, 0xff, vocab * 4, S));                                                                        \
    hipLaunchKernelGGL((k_cls_ring<RR, MV, false, RingStager<true, MV, 0, STG>>), dim3(GRID), dim3(WG), lds, S, args(0, o_new));                \
    CK(hipStreamSynchronize(S));                                                                                          \
    const bool ok = same((WG) == 256 ? o_ref : o_ref512, o_new, vocab, "cls logits");                                     \
    const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_cls_ring<RR, MV, false, RingStager<true, MV, 0, STG>>), dim3(GRID), dim3(WG), lds, S, args(l, o_new)); }); \
    char v[96]; snprintf(v, sizeof v, "ring R%d wg%d grid%d %s (%zu KB)", RR, WG, GRID, (STG) ? "x:regs" : "x:dma", lds >> 10); \
    report("cls", v, t, bytes, ok, base);                                                                                 \
  } while (0)
    CLS_RINGS(2, 4, 256, 512, 0);
    CLS_RINGS(2, 4, 256, 512, 1);
    CLS_RINGS(3, 4, 256, 512, 1);
    CLS_RINGS(2, 4, 256, 768, 1);
    CLS_RINGS(2, 4, 256, 1024, 1);
  }
  // ---------------------------------------------------------------- wo (K = 4096, M = 4096, split 1) and w2 (K = 4096, M = 11008, split 2)
  for (int which = 2; which <= 3; ++which) {
    if (!(only < 0 || only == which)) continue;
    const int M = which == 2 ? dim : hidden, K = dim;
    const char* name = which == 2 ? "wo" : "w2";
    const size_t wb = (size_t)K * M, sb = wb / 64;
    const double bytes = (double)wb + sb * 4.0 + M * 4.0 + 2.0 * K * 4;
    float* vec = which == 2 ? x : xh;
    auto args = [&](int l, float* xres) {
      KhGemvResArgs a{}; a.vec = vec; a.w = KhLin{w + wb * l, sc + sb * l, nullptr}; a.x = xres; a.M = M; a.K = K; a.gshift = gshift; return a;
    };
    const size_t lds0 = fused_lds_bytes(true, M);
    optin(k_gemv_res<true, 2, 6, 2>, lds0);
    auto ship = [&](int l, float* xres) {
      if (which == 2) hipLaunchKernelGGL((k_gemv_res<true, 4, 4, 1>), dim3(512), dim3(256), lds0, S, args(l, xres));
      else hipLaunchKernelGGL((k_gemv_res<true, 2, 6, 2>), dim3(512), dim3(512), lds0, S, args(l, xres));
    };
    CK(hipMemcpyAsync(o_ref, x0, K * 4, hipMemcpyDeviceToDevice, S));
    ship(0, o_ref);
    CK(hipStreamSynchronize(S));
    const float base = time_graph(NL, [&](int l) { ship(l, o_new); });
    report(name, which == 2 ? "shipped wg256 grid512 U4 split1" : "shipped wg512 grid512 U2 split2", base, bytes, true, 0);
    TRACE(name, "shipped", 512, [&](int l) { ship(l, o_new); });
#define RES_RINGS(RR, MV, SP, WG, GRID, STG)                                                                              \
  do {                                                                                                                    \
    const size_t lds = ring_lds_off(M, false) + (size_t)((WG) / 64) * (RR) * KH_RING_SLOT;                                \
    if (lds * (((GRID) + 255) / 256) > 160 * 1024) { printf("%-6s ring R%d wg%d grid%d: LDS does not fit\n", name, RR, WG, GRID); break; } \
    if (kh_stage_maxv(M, WG) != MV) break;                                                                                \
    optin(k_gemv_res_ring<RR, MV, SP, STG>, lds);                                                                         \
    CK(hipMemcpyAsync(o_new, x0, K * 4, hipMemcpyDeviceToDevice, S));                                                     \
    hipLaunchKernelGGL((k_gemv_res_ring<RR, MV, SP, STG>), dim3(GRID), dim3(WG), lds, S, args(0, o_new));                 \
    CK(hipStreamSynchronize(S));                                                                                          \
    const bool ok = same(o_ref, o_new, K, "x after residual");                                                            \
    const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_gemv_res_ring<RR, MV, SP, STG>), dim3(GRID), dim3(WG), lds, S, args(l, o_new)); }); \
    char v[96]; snprintf(v, sizeof v, "ring R%d wg%d grid%d split%d %s (%zu KB)", RR, WG, GRID, SP, (STG) ? "x:regs" : "x:dma", lds >> 10); \
    report(name, v, t, bytes, ok, base);                                                                                  \
  } while (0)
    if (which == 2) {
      RES_RINGS(2, 4, 1, 256, 512, 0);
      RES_RINGS(2, 4, 1, 256, 512, 1);
      RES_RINGS(4, 4, 1, 256, 512, 1);
      RES_RINGS(2, 4, 1, 256, 1024, 1);
      RES_RINGS(2, 4, 1, 512, 256, 1);
    } else {
      RES_RINGS(2, 6, 2, 512, 256, 0);
      RES_RINGS(2, 6, 2, 512, 256, 1);
      RES_RINGS(3, 6, 2, 512, 256, 1);
      RES_RINGS(1, 6, 2, 512, 512, 1);
      RES_RINGS(2, 6, 2, 512, 512, 1);
    }
  }
  // ---------------------------------------------------------------- qkv (6144 pairs x 4096, RoPE epilogue, pos 5)
  if (only < 0 || only == 4) {
    const int hs = 128, cache_len = 64, pos = 5;
    const size_t wb = (size_t)3 * dim * dim, sb = wb / 64;
    const double bytes = (double)wb + sb * 4.0 + 2.0 * dim * 4 + 3.0 * dim * 4;
    float *kc, *vc, *sinc, *cosc; int32_t* d_pos;
    CK(hipMalloc(&kc, (size_t)cache_len * dim * 4)); CK(hipMalloc(&vc, (size_t)cache_len * dim * 4));
    CK(hipMalloc(&sinc, (size_t)cache_len * hs * 4)); CK(hipMalloc(&cosc, (size_t)cache_len * hs * 4)); CK(hipMalloc(&d_pos, 4));
    hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, S, sinc, (size_t)cache_len * hs, 0x5u, -1.f, 1.f);
    hipLaunchKernelGGL(k_fill_f32, dim3(64), dim3(256), 0, S, cosc, (size_t)cache_len * hs, 0x6u, -1.f, 1.f);
    CK(hipMemcpyAsync(d_pos, &pos, 4, hipMemcpyHostToDevice, S));
    CK(hipStreamSynchronize(S));
    // outputs: q -> out[0 .. dim), k row -> out[dim .. 2 dim), v row -> out[2 dim .. 3 dim) by pointing the caches at out
    auto args = [&](int l, float* out) {
      const int8_t* w0 = w + wb * l; const float* s0 = sc + sb * l;
      KhQkvArgs a{}; a.x = x; a.att_norm = wn;
      a.wq = KhLin{w0, s0, nullptr}; a.wk = KhLin{w0 + wb / 3, s0 + sb / 3, nullptr}; a.wv = KhLin{w0 + 2 * (wb / 3), s0 + 2 * (sb / 3), nullptr};
      a.q_out = out; a.kcache_layer = out + dim - (size_t)pos * dim; a.vcache_layer = out + 2 * dim - (size_t)pos * dim;
      a.d_pos = d_pos; a.sin_cache = sinc; a.cos_cache = cosc; a.dim = dim; a.kv_dim = dim; a.head_size = hs;
      a.rope_mode = KH_ROPE_INTERLEAVED; a.gshift = gshift; a.eps = 1e-5f; return a;
    };
    const size_t lds0 = fused_lds_bytes(true, dim);
    hipLaunchKernelGGL((k_qkv<true, 4, 4, 1>), dim3(512), dim3(256), lds0, S, args(0, o_ref));
    hipLaunchKernelGGL((k_qkv<true, 4, 4, 1>), dim3(256), dim3(512), lds0, S, args(0, o_ref512));
    CK(hipStreamSynchronize(S));
    const float base = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_qkv<true, 4, 4, 1>), dim3(512), dim3(256), lds0, S, args(l, o_new)); });
    report("qkv", "shipped wg256 grid512 U4", base, bytes, true, 0);
    TRACE("qkv", "shipped wg256 grid512", 512, [&](int l) { hipLaunchKernelGGL((k_qkv<true, 4, 4, 1>), dim3(512), dim3(256), lds0, S, args(l, o_new)); });
#define QKV_RINGS(RR, MV, WG, GRID, STG)                                                                                  \
  do {                                                                                                                    \
    const size_t lds = ring_lds_off(dim, (STG) == 0) + (size_t)((WG) / 64) * (RR) * KH_RING_SLOT;                         \
    if (lds * (((GRID) + 255) / 256) > 160 * 1024) { printf("qkv    ring R%d wg%d grid%d: LDS does not fit\n", RR, WG, GRID); break; } \
    optin(k_qkv_ring<RR, MV, 1, STG>, lds);                                                                               \
    CK(hipMemsetAsync(o_new, 0xff, 3 * dim * 4, S));                                                                      \
    hipLaunchKernelGGL((k_qkv_ring<RR, MV, 1, STG>), dim3(GRID), dim3(WG), lds, S, args(0, o_new));                       \
    CK(hipStreamSynchronize(S));                                                                                          \
    const bool ok = same((WG) == 256 ? o_ref : o_ref512, o_new, 3 * dim, "q | k row | v row");                            \
    const float t = time_graph(NL, [&](int l) { hipLaunchKernelGGL((k_qkv_ring<RR, MV, 1, STG>), dim3(GRID), dim3(WG), lds, S, args(l, o_new)); }); \
    char v[96]; snprintf(v, sizeof v, "ring R%d wg%d grid%d %s (%zu KB)", RR, WG, GRID, (STG) ? "x:regs" : "x:dma", lds >> 10); \
    report("qkv", v, t, bytes, ok, base);                                                                                 \
  } while (0)
    QKV_RINGS(2, 4, 256, 768, 0);
    QKV_RINGS(2, 4, 256, 512, 1);
    QKV_RINGS(3, 4, 256, 512, 1);
    QKV_RINGS(2, 4, 256, 768, 1);
    QKV_RINGS(3, 4, 256, 768, 1);
    QKV_RINGS(2, 4, 256, 1024, 1);
  }
  return 0;
}
