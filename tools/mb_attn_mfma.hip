// mb_attn_mfma.hip — decode attention of a GQA group on the matrix cores, measured.
//
// The north star names the QK^T / .V contractions "on MFMA".  For a batch-1 DECODE step the query side of both
// products is kv_mul (4 for Llama-3.2-1B) vectors, so on v_mfma_f32_16x16x4_f32 - the only exact-fp32 shape - they
// fill kv_mul of the 16 N columns.  DESIGN §7 argued that this loses; this is the same-box A/B:
//   VALU  the shipped GQA group core (kh_attn.h::attn_group_partial<16, 4>: one workgroup per (kv group, time split),
//         16 lanes per timestep, four query heads per pass over the K/V rows, online softmax per lane group)
//   MFMA  the same workgroup decomposition; a wave owns 16-timestep tiles: S[t, head] = K[t, :] . Q[head, :]^T as
//         16 MFMAs (K rows as 16-byte loads with the k-quad trick of kh_gemm.h, heads on N padded to 16), online
//         softmax on the D registers (their layout - lane = head column, registers = 4 consecutive timesteps - is the
//         B operand of the second product), O^T[d, head] += V^T[d, t] . P[t, head] as 16 more MFMAs (V rows as
//         16-byte loads, the head dimension permuted d = 4 i + c so that a float4 feeds four d-tiles).
// Both write the (M, L, o) partial of every (head, split); the host merges them and compares the two outputs.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mb_attn_mfma.hip -o kuiperllama_amd/lib/mb_attn_mfma
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../kuiperllama_amd/csrc/kh_attn.h"

namespace khm {
const char* dbg(const char*) { return nullptr; }
}  // namespace khm
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

struct PartArgs {
  const float* q;       // [heads * hs]
  const float* kc;      // layer base
  const float* vc;
  float* part;          // [heads][NS][hs + 2]: o[hs] | M | L
  int pos, kv_dim, kv_heads, hs, NS;
};

template <int KVM>
__global__ __launch_bounds__(512) void k_part_valu(const PartArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int g = blockIdx.x % a.kv_heads, s = blockIdx.x / a.kv_heads;
  const int nT = a.pos + 1, TS = attn_split_len(nT, a.NS), nact = (nT + TS - 1) / TS;
  if (s >= nact) return;
  const int t_begin = s * TS, t_end = t_begin + TS < nT ? t_begin + TS : nT;
  float r, L, M;
  attn_group_partial<16, KVM>(a.q + (size_t)g * KVM * a.hs, a.kc + (size_t)g * a.hs, a.vc + (size_t)g * a.hs, a.kv_dim,
                              a.hs, t_begin, t_end, (float*)smem_raw, r, L, M);
  const int tid = threadIdx.x;
  if (tid < KVM * a.hs) {
    const int j = tid / a.hs, e = tid - j * a.hs;
    float* p = a.part + ((size_t)(g * KVM + j) * a.NS + s) * (a.hs + 2);
    p[e] = r;
    if (e == 0) {
      p[a.hs] = M;
      p[a.hs + 1] = L;
    }
  }
}

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// hs == 64.  LDS: red_m[8][KVM] | red_l[8][KVM] | opart[8][KVM][64]
template <int KVM>
__global__ __launch_bounds__(512) void k_part_mfma(const PartArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  float* red_m = (float*)smem_raw;
  float* red_l = red_m + 8 * KVM;
  float* opart = red_l + 8 * KVM;
  const int g = blockIdx.x % a.kv_heads, s = blockIdx.x / a.kv_heads;
  const int nT = a.pos + 1, TS = attn_split_len(nT, a.NS), nact = (nT + TS - 1) / TS;
  if (s >= nact) return;
  const int t_begin = s * TS, t_end = t_begin + TS < nT ? t_begin + TS : nT;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (int)(blockDim.x >> 6);
  const int i = lane & 15, h = lane >> 4;
  const int stride4 = a.kv_dim >> 2;
  const f32x4* K4 = (const f32x4*)(a.kc + (size_t)g * 64);
  const f32x4* V4 = (const f32x4*)(a.vc + (size_t)g * 64);
  // B operand of the first product: Q[head = lane & 15][16 b + 4 (lane >> 4) + s], zero for the padded columns
  float qf[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[b][c] = i < KVM ? a.q[(size_t)(g * KVM + i) * 64 + 16 * b + 4 * h + c] : 0.f;
  const float sc2 = 1.4426950408889634f / sqrtf(64.f);
  float m_run = -INFINITY, l_run = 0.f;
  f32x4 oacc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) oacc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
  for (int t0 = t_begin + 16 * wave; t0 < t_end; t0 += 16 * nw) {
    f32x4 kf[4], vf[4];
    const int row = t0 + i < t_end ? t0 + i : t_end - 1;
#pragma unroll
    for (int b = 0; b < 4; ++b) kf[b] = ld_nt(K4 + (size_t)row * stride4 + 4 * b + h);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tr = t0 + 4 * h + r < t_end ? t0 + 4 * h + r : t_end - 1;
      vf[r] = ld_nt(V4 + (size_t)tr * stride4 + i);
    }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      acc = mfma4(kf[b].x, qf[b][0], acc);
      acc = mfma4(kf[b].y, qf[b][1], acc);
      acc = mfma4(kf[b].z, qf[b][2], acc);
      acc = mfma4(kf[b].w, qf[b][3], acc);
    }
    // D: lane (column = head i, h) holds timesteps t0 + 4 h + r
    float sv[4], mt = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sv[r] = t0 + 4 * h + r < t_end ? acc[r] * sc2 : -INFINITY;
      mt = fmaxf(mt, sv[r]);
    }
    mt = across_groups_max<16>(mt);
    const float mn = fmaxf(m_run, mt);
    const float alpha = __builtin_amdgcn_exp2f(m_run - mn);
    float p[4], ps = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      p[r] = __builtin_amdgcn_exp2f(sv[r] - mn);
      ps += p[r];
    }
    l_run = l_run * alpha + ps;
    m_run = mn;
#pragma unroll
    for (int c = 0; c < 4; ++c) oacc[c] = oacc[c] * alpha;
    // O^T[d = 4 m + c][head] += sum_t V[t][4 m + c] P[t][head]; MFMA #r contracts the timesteps t0 + 4 kk + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      oacc[0] = mfma4(vf[r].x, p[r], oacc[0]);
      oacc[1] = mfma4(vf[r].y, p[r], oacc[1]);
      oacc[2] = mfma4(vf[r].z, p[r], oacc[2]);
      oacc[3] = mfma4(vf[r].w, p[r], oacc[3]);
    }
  }
  const float l_tot = across_groups_sum<16>(l_run);
  if (i < KVM) {
    if (h == 0) {
      red_m[wave * KVM + i] = m_run;
      red_l[wave * KVM + i] = l_tot;
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) opart[(size_t)(wave * KVM + i) * 64 + 4 * (4 * h + rr) + c] = oacc[c][rr];
  }
  __syncthreads();
  if (tid < KVM * 64) {
    const int j = tid >> 6, e = tid & 63;
    float M = -INFINITY;
    for (int w = 0; w < nw; ++w) M = fmaxf(M, red_m[w * KVM + j]);
    float r = 0.f, L = 0.f;
    for (int w = 0; w < nw; ++w) {
      const float mw = red_m[w * KVM + j];
      const float f = mw == -INFINITY ? 0.f : __builtin_amdgcn_exp2f(mw - M);
      r += opart[(size_t)(w * KVM + j) * 64 + e] * f;
      L += red_l[w * KVM + j] * f;
    }
    float* p = a.part + ((size_t)(g * KVM + j) * a.NS + s) * (64 + 2);
    p[e] = r;
    if (e == 0) {
      p[64] = M * 0.6931471805599453f;
      p[65] = L;
    }
  }
}

static void fill_rand(float* d, size_t n, float amp, uint32_t seed) {
  std::vector<float> h(n);
  uint32_t s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = ((float)(s >> 8) * (1.0f / 16777216.0f) - 0.5f) * 2.f * amp;
  }
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
}

int main() {
  const int heads = 32, kvh = 8, hs = 64, kv_dim = kvh * hs, KVM = 4, NS = 32, LAYERS = 4;
  const int cache_len = 131072;
  hipStream_t st;
  CK(hipStreamCreate(&st));
  float *kc, *vc, *q, *part_v, *part_m;
  const size_t layer_elems = (size_t)cache_len * kv_dim;
  CK(hipMalloc(&kc, LAYERS * layer_elems * 4));
  CK(hipMalloc(&vc, LAYERS * layer_elems * 4));
  CK(hipMalloc(&q, heads * hs * 4));
  const size_t pn = (size_t)heads * NS * (hs + 2);
  CK(hipMalloc(&part_v, pn * 4));
  CK(hipMalloc(&part_m, pn * 4));
  for (int l = 0; l < LAYERS; ++l) {
    for (size_t off = 0; off < layer_elems; off += (size_t)16384 * kv_dim) {
      const size_t n = layer_elems - off < (size_t)16384 * kv_dim ? layer_elems - off : (size_t)16384 * kv_dim;
      fill_rand(kc + l * layer_elems + off, n, 1.0f, (uint32_t)(l * 100 + off / 4096 + 1));
      fill_rand(vc + l * layer_elems + off, n, 1.0f, (uint32_t)(l * 100 + off / 4096 + 7));
    }
  }
  fill_rand(q, heads * hs, 1.0f, 3);
  const size_t lds_v = attn_group_lds_bytes(hs, KVM), lds_m = (size_t)(16 * KVM + 8 * KVM * 64) * 4;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("Llama-3.2-1B geometry (32 heads, 8 KV heads, head size 64), %d splits per KV group, 512-thread workgroups\n", NS);
  for (int pos : {4095, 32767, 131071}) {
    auto args = [&](int l, float* part) {
      PartArgs a{q, kc + l * layer_elems, vc + l * layer_elems, part, pos, kv_dim, kvh, hs, NS};
      return a;
    };
    // correctness: merge the partials of both kernels on the host, compare the outputs
    hipLaunchKernelGGL(k_part_valu<4>, dim3(kvh * NS), dim3(512), lds_v, st, args(0, part_v));
    hipLaunchKernelGGL(k_part_mfma<4>, dim3(kvh * NS), dim3(512), lds_m, st, args(0, part_m));
    CK(hipStreamSynchronize(st));
    std::vector<float> hv(pn), hm(pn);
    CK(hipMemcpy(hv.data(), part_v, pn * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hm.data(), part_m, pn * 4, hipMemcpyDeviceToHost));
    const int nT = pos + 1, TS = attn_split_len(nT, NS), nact = (nT + TS - 1) / TS;
    double worst = 0;
    for (int hd = 0; hd < heads; ++hd) {
      auto merged = [&](const std::vector<float>& P, int e) {
        double M = -1e300;
        for (int s = 0; s < nact; ++s) M = fmax(M, (double)P[((size_t)hd * NS + s) * (hs + 2) + hs]);
        double num = 0, den = 0;
        for (int s = 0; s < nact; ++s) {
          const float* p = &P[((size_t)hd * NS + s) * (hs + 2)];
          const double f = exp((double)p[hs] - M);
          num += p[e] * f;
          den += p[hs + 1] * f;
        }
        return num / den;
      };
      for (int e = 0; e < hs; ++e) worst = fmax(worst, fabs(merged(hv, e) - merged(hm, e)));
    }
    // timing: LAYERS rotating caches per graph so consecutive launches do not hit the same lines
    float us[2];
    for (int which = 0; which < 2; ++which) {
      hipGraph_t g;
      hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      for (int rep = 0; rep < 4; ++rep)
        for (int l = 0; l < LAYERS; ++l) {
          if (which == 0) hipLaunchKernelGGL(k_part_valu<4>, dim3(kvh * NS), dim3(512), lds_v, st, args(l, part_v));
          else hipLaunchKernelGGL(k_part_mfma<4>, dim3(kvh * NS), dim3(512), lds_m, st, args(l, part_m));
        }
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st));
      CK(hipStreamSynchronize(st));
      float best = 1e9f;
      for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0, st));
        CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
      }
      us[which] = best * 1e3f / (4 * LAYERS);
      CK(hipGraphExecDestroy(ge));
      CK(hipGraphDestroy(g));
    }
    const double mb = 2.0 * (pos + 1) * kv_dim * 4 / 1e6;
    printf("pos %6d (%7.2f MB of K/V): VALU %7.2f us (%.2f TB/s)   MFMA %7.2f us (%.2f TB/s)   MFMA/VALU %.3f   max |out diff| %.2e %s\n",
           pos, mb, us[0], mb / us[0], us[1], mb / us[1], us[1] / us[0], worst, worst < 2e-5 ? "ok" : "BAD");
  }
  return 0;
}
