// mb_attn_phases.hip - where the microseconds of a decode-attention launch go (round 5).
//
// Builds the PRODUCT launch (kh_attn.h::k_attn_decode through launch_attn_decode, device-side position, the model's
// split plan) with -DKH_ATTN_TRACE: thread 0 of every workgroup stamps the 100-MHz constant clock at
//   0 kernel entry   1 position known, split geometry done   2 first K/V batch consumed   3 last batch consumed
//   4 lane groups / waves folded (LDS)   5 partial published (stores drained, barrier)   6 ticket returned
//   7 merged output stored (last arriver only)
// and lane 0 of every wave stamps "first batch consumed" / "last batch consumed" (the skew between the waves of a
// workgroup ends up in the fold phase of thread 0, which waits at the barrier).
// Llama-3.2-1B geometry, 16 rotating layer caches of 131072 rows, a 128-MB streaming launch between two attention
// launches (in the model the K/V rows of a layer were last touched a token ago, 5 GB of weights earlier).  Per
// position: event-timed launch (hipGraph of the alternation, filler subtracted) and, from one traced launch per layer,
// the critical path = per stamp the LATEST workgroup relative to the EARLIEST entry, and per-workgroup medians of the
// phase lengths.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -DKH_ATTN_TRACE tools/mb_attn_phases.hip -o kuiperllama_amd/lib/mb_attn_phases
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "../kuiperllama_amd/csrc/kh_attn.h"

namespace khm {
const char* dbg(const char* k) { return getenv(k); }
}  // namespace khm
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void k_fill(float* d, size_t n, unsigned seed) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)(i * 2654435761u) ^ seed;
    s = s * 1664525u + 1013904223u;
    s ^= s >> 15;
    s = s * 1664525u + 1013904223u;
    d[i] = ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.f;
  }
}
__global__ __launch_bounds__(256) void k_stream(const f32x4* __restrict__ src, size_t n4, float* sink) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    const f32x4 v = ld_nt(src + i);
    acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
  }
  if (acc.x + acc.y + acc.z + acc.w == 12345.678f) sink[0] = acc.x;
}

int main(int argc, char** argv) {
  const int heads = 32, kvh = 8, hs = 64, kv_dim = kvh * hs, kv_mul = 4, LAYERS = 16, wg = 512;
  const int cache_len = 131072;
  const int tlong = argc > 1 ? atoi(argv[1]) : -1;  // KH_ATTN_TLONG-style override: 0 = per-head path at every position
  // argv[2]: 0 = no streaming launch between the attention launches, and every launch on the SAME layer's rows (the
  // instruction cache, the L2 and the TLB stay warm: what the phases cost when nothing has to come from far away)
  const bool cold = argc > 2 ? atoi(argv[2]) != 0 : true;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *kc, *vc, *q, *out, *filler, *sink;
  const size_t layer_elems = (size_t)cache_len * kv_dim;
  CK(hipMalloc(&kc, LAYERS * layer_elems * 4));
  CK(hipMalloc(&vc, LAYERS * layer_elems * 4));
  CK(hipMalloc(&q, heads * hs * 4));
  CK(hipMalloc(&out, heads * hs * 4));
  const size_t fill_bytes = (size_t)128 << 20;
  CK(hipMalloc(&filler, fill_bytes));
  CK(hipMalloc(&sink, 64));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, kc, LAYERS * layer_elems, 1u);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, vc, LAYERS * layer_elems, 7u);
  hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, st, q, (size_t)heads * hs, 3u);
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, st, filler, fill_bytes / 4, 9u);
  const AttnPlan plan = attn_plan(heads, kv_mul, hs, cache_len, wg, tlong);
  const size_t wsb = attn_ws_bytes(heads, hs, plan.stride);
  void* ws;
  CK(hipMalloc(&ws, wsb));
  CK(hipMemsetAsync(ws, 0, wsb, st));
  int32_t* d_pos;
  CK(hipMalloc(&d_pos, 4));
  const int max_grid = 4096;
  unsigned long long* tr;
  CK(hipMalloc(&tr, (size_t)max_grid * 32 * 8));
  CK(hipMemcpyToSymbol(HIP_SYMBOL(kh_attn_trace_buf), &tr, sizeof(tr)));
  auto args = [&](int l) {
    KhAttnArgs a{};
    a.q = q;
    a.kcache_layer = kc + l * layer_elems;
    a.vcache_layer = vc + l * layer_elems;
    a.out = out;
    a.d_pos = d_pos;
    a.kv_dim = kv_dim;
    a.kv_mul = kv_mul;
    a.head_size = hs;
    a.kv_heads = kvh;
    a.nsplit = plan.ns;
    a.ws = ws;
    a.ws_stride = plan.stride;
    a.nsplit_g = plan.ns_g;
    a.t_long = plan.t_long;
    a.ts_shift = plan.ts_shift;
    a.defer = 0;
    a.fenced = 0;
    a.tok_stride = 0;
    a.ws_tok_bytes = 0;
    return a;
  };
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%s\n", cold ? "COLD: a 128-MB streaming launch before every attention launch, 16 rotating layers"
                        : "WARM: back-to-back launches on one layer's rows");
  printf("Llama-3.2-1B geometry; plan: %d splits per head (quantum %d), %d per KV group from position %d on; stamps in us "
         "after the earliest workgroup's entry\n", plan.ns, 1 << plan.ts_shift, plan.ns_g, plan.t_long - 1);
  std::vector<int> poss = {63, 1023, 4094, 4095, 4096, 8191, 16383, 32768, 131071};
  for (int pos : poss) {
    CK(hipMemcpyAsync(d_pos, &pos, 4, hipMemcpyHostToDevice, st));
    CK(hipStreamSynchronize(st));
    // ---- event timing: graph of (filler, attention) x LAYERS against (filler) x LAYERS
    float us[2];
    for (int which = 0; which < 2; ++which) {
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int l = 0; l < LAYERS; ++l) {
        if (cold) hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, st, (const f32x4*)filler, fill_bytes / 16, sink);
        if (which == 0) launch_attn_decode(args(cold ? l : 0), 0, wg, st);
      }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      float best = 1e9f;
      for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      us[which] = best * 1e3f / LAYERS;
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    }
    // ---- traced launches, one per layer
    double crit[8] = {0}, med_phase[8] = {0}, skew1 = 0, skew3 = 0, lastwave3 = 0;
    int active = 0, grid_seen = 0;
    for (int l = 0; l < LAYERS; ++l) {
      CK(hipMemsetAsync(tr, 0, (size_t)max_grid * 256, st));
      if (cold) hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, st, (const f32x4*)filler, fill_bytes / 16, sink);
      else launch_attn_decode(args(0), 0, wg, st);  // warm: the traced launch follows an identical one
      launch_attn_decode(args(cold ? l : 0), 0, wg, st);
      CK(hipStreamSynchronize(st));
      std::vector<unsigned long long> h((size_t)max_grid * 32);
      CK(hipMemcpy(h.data(), tr, h.size() * 8, hipMemcpyDeviceToHost));
      unsigned long long t0 = ~0ull;
      int nb = 0;
      for (int b = 0; b < max_grid; ++b)
        if (h[(size_t)b * 32]) {
          t0 = std::min(t0, h[(size_t)b * 32]);
          nb = b + 1;
        }
      grid_seen = nb;
      std::vector<double> ph[8], sk1, sk3;
      double late[8] = {0}, late_w3 = 0;
      int act = 0;
      for (int b = 0; b < nb; ++b) {
        const unsigned long long* s = &h[(size_t)b * 32];
        if (!s[0]) continue;
        if (s[1]) ++act;
        if (s[1]) {  // per-wave stamps: first batch consumed (8..15), last batch consumed (16..23)
          unsigned long long lo1 = ~0ull, hi1 = 0, lo3 = ~0ull, hi3 = 0;
          for (int w = 0; w < 8; ++w) {
            if (s[8 + w]) { lo1 = std::min(lo1, s[8 + w]); hi1 = std::max(hi1, s[8 + w]); }
            if (s[16 + w]) { lo3 = std::min(lo3, s[16 + w]); hi3 = std::max(hi3, s[16 + w]); }
          }
          if (hi1) sk1.push_back((double)(hi1 - lo1) * 0.01);
          if (hi3) {
            sk3.push_back((double)(hi3 - lo3) * 0.01);
            late_w3 = std::max(late_w3, (double)(hi3 - t0) * 0.01);
          }
        }
        unsigned long long prev = s[0];
        for (int i = 0; i < 8; ++i) {
          if (!s[i]) continue;
          late[i] = std::max(late[i], (double)(s[i] - t0) * 0.01);
          if (i) ph[i].push_back((double)(s[i] - prev) * 0.01);
          prev = s[i];
        }
      }
      active = act;
      if (!sk1.empty()) { std::sort(sk1.begin(), sk1.end()); skew1 += sk1[sk1.size() / 2] / LAYERS; }
      if (!sk3.empty()) { std::sort(sk3.begin(), sk3.end()); skew3 += sk3[sk3.size() / 2] / LAYERS; }
      lastwave3 += late_w3 / LAYERS;
      for (int i = 0; i < 8; ++i) {
        crit[i] += late[i] / LAYERS;
        if (!ph[i].empty()) {
          std::sort(ph[i].begin(), ph[i].end());
          med_phase[i] += ph[i][ph[i].size() / 2] / LAYERS;
        }
      }
    }
    const double kvb = 2.0 * (pos + 1) * kv_dim * 4;
    printf("pos %6d  launch %6.2f us (K/V %.1f MB = %.2f of 8 TB/s)  grid %d, %d workgroups own timesteps\n", pos,
           us[0] - us[1], kvb / 1e6, kvb / ((us[0] - us[1]) * 1e-6) / 8e12, grid_seen, active);
    printf("    latest workgroup at stamp:   entry %.2f | pos %.2f | batch1 %.2f | batches %.2f | folded %.2f | published %.2f | ticket %.2f | merged %.2f\n",
           crit[0], crit[1], crit[2], crit[3], crit[4], crit[5], crit[6], crit[7]);
    printf("    median phase per workgroup:  pos %.2f | batch1 %.2f | rest %.2f | fold %.2f | publish %.2f | ticket %.2f | merge %.2f\n",
           med_phase[1], med_phase[2], med_phase[3], med_phase[4], med_phase[5], med_phase[6], med_phase[7]);
    printf("    waves of a workgroup: first batch consumed within %.2f us of each other (median), last batch within %.2f; "
           "latest wave of the launch done with its batches at %.2f\n", skew1, skew3, lastwave3);
  }
  return 0;
}
