// mb_skew.hip - when do the waves of a decode GEMV launch finish?  The Llama-3.2-1B fp32 ffn13 and w2 launches and the
// Llama-2-7B int8 ffn13 (ring) launch, built with -DKH_TRACE (kh_common.h phase stamps), launched BACK TO BACK over
// distinct weight slabs as inside a decode step; the stamps of the last launch of the chain are read back: per wave
// "last work item finished", per workgroup entry / end.  Reports percentiles of the wave finish times and the per-XCD
// medians.  A wide spread at equal static work would mean a dynamic work distribution has something to win.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -std=c++20 -w -DKH_TRACE -I kuiperllama_amd/csrc -I tools tools/mb_skew.hip -o kuiperllama_amd/lib/mb_skew
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "kh_fused_ring.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_fill_f32(float* p, size_t n, uint32_t seed, float lo, float hi) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed ^ (uint32_t)(i >> 32) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
    p[i] = lo + (hi - lo) * (float)(h >> 8) * (1.0f / 16777216.0f);
  }
}
__global__ void k_fill_i8(int8_t* p, size_t n, uint32_t seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed ^ (uint32_t)(i >> 32) * 40503u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    p[i] = (int8_t)(h & 0xff);
  }
}
static hipStream_t S;
static unsigned long long* g_tr;

static void report(const char* name, int grid, int wpw, int reps, const std::function<void(int)>& launch, int NL) {
  std::vector<unsigned long long> h((size_t)grid * 32);
  std::vector<double> pct[7];
  double dur = 0, xcd_med[8] = {0};
  for (int r = 0; r < reps; ++r) {
    CK(hipMemsetAsync(g_tr, 0, (size_t)grid * 256, S));
    for (int l = 0; l < NL; ++l) launch(l);  // back to back; the stamps of the last launch survive
    CK(hipStreamSynchronize(S));
    CK(hipMemcpy(h.data(), g_tr, h.size() * 8, hipMemcpyDeviceToHost));
    unsigned long long t0 = ~0ull, t1 = 0;
    for (int b = 0; b < grid; ++b) { if (h[(size_t)b * 32]) t0 = std::min(t0, h[(size_t)b * 32]); t1 = std::max(t1, h[(size_t)b * 32 + 4]); }
    std::vector<double> done, per_xcd[8];
    for (int b = 0; b < grid; ++b)
      for (int w = 0; w < wpw; ++w)
        if (h[(size_t)b * 32 + 8 + w]) { const double t = (double)(h[(size_t)b * 32 + 8 + w] - t0) * 0.01; done.push_back(t); per_xcd[b & 7].push_back(t); }
    std::sort(done.begin(), done.end());
    const double q[7] = {0.0, 0.05, 0.25, 0.5, 0.75, 0.95, 1.0};
    for (int i = 0; i < 7; ++i) pct[i].push_back(done[(size_t)(q[i] * (done.size() - 1))]);
    dur += (double)(t1 - t0) * 0.01 / reps;
    for (int x = 0; x < 8; ++x) { std::sort(per_xcd[x].begin(), per_xcd[x].end()); xcd_med[x] += per_xcd[x][per_xcd[x].size() / 2] / reps; }
  }
  printf("%-34s entry->end %6.2f us | wave finish times: min %5.2f  p5 %5.2f  p25 %5.2f  median %5.2f  p75 %5.2f  p95 %5.2f  max %5.2f", name, dur,
         pct[0][reps / 2], pct[1][reps / 2], pct[2][reps / 2], pct[3][reps / 2], pct[4][reps / 2], pct[5][reps / 2], pct[6][reps / 2]);
  printf(" | median per XCD (workgroup %% 8):");
  for (int x = 0; x < 8; ++x) printf(" %5.2f", xcd_med[x]);
  printf("\n");
  fflush(stdout);
}

int main() {
  CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  CK(hipMalloc(&g_tr, (size_t)4096 * 32 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(kh_trace_buf), &g_tr, sizeof(g_tr)));
  const int NL = 8;
  float *x, *xh, *wn, *out;
  CK(hipMalloc(&x, 65536)); CK(hipMalloc(&xh, 65536)); CK(hipMalloc(&wn, 65536)); CK(hipMalloc(&out, 65536));
  hipLaunchKernelGGL(k_fill_f32, dim3(16), dim3(256), 0, S, x, 16384, 0x1u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_f32, dim3(16), dim3(256), 0, S, xh, 16384, 0x2u, -1.f, 1.f);
  hipLaunchKernelGGL(k_fill_f32, dim3(16), dim3(256), 0, S, wn, 16384, 0x3u, 0.5f, 1.5f);
  printf("wave finish times (us after the launch's first workgroup entry), last launch of %d back-to-back launches over distinct slabs, median of 9 chains\n", NL);
  {  // Llama-3.2-1B fp32
    const int dim = 2048, hidden = 8192;
    const size_t slab = (size_t)2 * hidden * dim;
    float* w; CK(hipMalloc(&w, slab * NL * 4));
    hipLaunchKernelGGL(k_fill_f32, dim3(4096), dim3(256), 0, S, w, slab * NL, 0x1234u, -0.05f, 0.05f);
    CK(hipStreamSynchronize(S));
    auto fa = [&](int l) { KhFfn13Args a{}; a.x = x; a.ffn_norm = wn; a.w1 = KhLin{w + slab * l, nullptr, nullptr}; a.w3 = KhLin{w + slab * l + slab / 2, nullptr, nullptr};
                           a.h = out; a.dim = dim; a.hidden = hidden; a.gshift = 0; a.eps = 1e-5f; return a; };
    const size_t lds0 = fused_lds_bytes(false, dim);
    report("1B fp32 ffn13 <false,8,2> 512x256", 512, 4, 9, [&](int l) { hipLaunchKernelGGL((k_ffn13<false, 8, 2>), dim3(512), dim3(256), lds0, S, fa(l)); }, NL);
    auto wa = [&](int l) { KhGemvResArgs a{}; a.vec = xh; a.w = KhLin{w + slab * l, nullptr, nullptr}; a.x = out; a.M = hidden; a.K = dim; a.gshift = 0; return a; };
    const size_t lds1 = fused_lds_bytes(false, hidden);
    CK(hipFuncSetAttribute((const void*)k_gemv_res<false, 8, 4, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1));
    report("1B fp32 w2 <false,8,4,4> 512x512", 512, 8, 9, [&](int l) { hipLaunchKernelGGL((k_gemv_res<false, 8, 4, 4>), dim3(512), dim3(512), lds1, S, wa(l)); }, NL);
    CK(hipFree(w));
  }
  {  // Llama-2-7B int8 ffn13 on the ring and on register tiles
    const int dim = 4096, hidden = 11008, gshift = 6;
    const size_t wb = (size_t)2 * hidden * dim, sb = wb / 64;
    int8_t* w; float* sc;
    CK(hipMalloc(&w, wb * NL)); CK(hipMalloc(&sc, sb * 4 * NL));
    hipLaunchKernelGGL(k_fill_i8, dim3(4096), dim3(256), 0, S, w, wb * NL, 0x1234u);
    hipLaunchKernelGGL(k_fill_f32, dim3(4096), dim3(256), 0, S, sc, sb * NL, 0x77u, 0.001f, 0.01f);
    CK(hipStreamSynchronize(S));
    auto fa = [&](int l) { KhFfn13Args a{}; a.x = x; a.ffn_norm = wn; a.w1 = KhLin{w + wb * l, sc + sb * l, nullptr}; a.w3 = KhLin{w + wb * l + wb / 2, sc + sb * l + sb / 2, nullptr};
                           a.h = out; a.dim = dim; a.hidden = hidden; a.gshift = gshift; a.eps = 1e-5f; return a; };
    const size_t ldsr = ring_lds_bytes(dim, false, 4, 2);
    report("7B int8 ffn13 ring<2,4> 512x256", 512, 4, 9, [&](int l) { hipLaunchKernelGGL((k_ffn13_ring<2, 4>), dim3(512), dim3(256), ldsr, S, fa(l)); }, NL);
    const size_t lds0 = fused_lds_bytes(true, dim);
    report("7B int8 ffn13 <true,4,4> 512x256", 512, 4, 9, [&](int l) { hipLaunchKernelGGL((k_ffn13<true, 4, 4>), dim3(512), dim3(256), lds0, S, fa(l)); }, NL);
  }
  return 0;
}
