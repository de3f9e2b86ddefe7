// mb_gemv_mfma.hip — the batch-1 decode GEMV on the matrix cores, measured (VERDICT r5 item 8 / missing #2).
//
// The north star says "GEMM and QK^T / .V contractions on MFMA".  The batch-1 decode matmul (reference kernel:
// kuiper/source/op/kernels/cuda/matmul_kernel.cu:7-54) is y[K] = W[K, M] . x[M]: one activation vector, i.e. ONE useful
// column of an MFMA's N dimension, 0.5 flop per weight byte.  DESIGN 7 argued that the matrix cores cannot win there and
// gave the reason (the operand layout dictates the access shape); decode ATTENTION got a real A/B in round 4
// (tools/mb_attn_mfma.hip: 10-23 % slower).  This is the same A/B for the dominant GEMV of the headline workload, the
// Llama-3.2-1B ffn13 launch (w1, w3: [8192, 2048] fp32 each + RMSNorm staging + SwiGLU, 134.3 MB per launch):
//
//   VALU    the shipped kernel, k_ffn13<false, 8, 2> (kh_fused.h): a wave streams one (w1 row, w3 row) pair, every
//           load instruction = 1 KiB contiguous of one row, v_fma_f32 from registers, one DPP butterfly per pair
//   MFMA16  v_mfma_f32_16x16x4_f32, x broadcast over the 16 N columns (15/16 of every MFMA is padding): a workgroup
//           owns 16 rows of w1 and of w3, its four waves split K; lane (m = l % 16, kq = l / 16) supplies A[m][kq],
//           so a dwordx4 load feeds four MFMAs and one load instruction touches 16 rows x 64 B
//   MFMA4   v_mfma_f32_4x4x1_16B_f32, the 16 blocks mapped to 16 consecutive k-quads of FOUR rows: lane (i = l % 4,
//           b = l / 4) supplies A_b[i], one load instruction = 4 rows x 256 B contiguous - the best access shape the
//           MFMA operand layouts allow for a GEMV; the 16 per-block partial sums are added across lanes at the end
// Same staging of the vector (Stager<true, false, 2>: g = w_norm * x in LDS, RMS scale in the epilogue), same bytes
// in flight per wave (16 x dwordx4 per matrix pair tile), same launch geometry class (512 x 256 threads), the same
// distinct weight slabs per launch of a captured graph (nothing is served from a cache).  Outputs are compared with
// the VALU kernel's (summation order differs: tolerance, not bits).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mb_gemv_mfma.hip -o kuiperllama_amd/lib/mb_gemv_mfma
//   mb_gemv_mfma [slabs per graph = 16] [replays = 20] [only = 0 | 1 valu | 2 mfma16 | 3 mfma4]   (rocprofv3 passes)
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../kuiperllama_amd/csrc/kh_fused.h"

namespace khm {
const char* dbg(const char*) { return nullptr; }
}  // namespace khm
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ float comp(const f32x4& v, int s) { return s == 0 ? v.x : s == 1 ? v.y : s == 2 ? v.z : v.w; }

constexpr int U = 8;  // dwordx4 loads per matrix in flight per lane: 2 x 8 KiB per wave, as the VALU kernel's tile

// ---- MFMA16: 16 rows per workgroup, K split over the 4 waves -------------------------------------------------------
// lds: xs[dim] | red[8] | comb[4 waves][2 matrices][16 rows]
__global__ __launch_bounds__(256) void k_ffn13_mfma16(const KhFfn13Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  const int dim = a.dim;
  float* red = (float*)(xs + (dim >> 2));
  float* comb = red + 8;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int m = lane & 15, kq = lane >> 4;
  Stager<true, false, 2> st(a.x, a.ffn_norm, dim);
  st.issue();
  __builtin_amdgcn_sched_barrier(0);
  const int items = a.hidden >> 4;
  const int cpw4 = dim >> 4;       // float4 per row per wave (K split 4 ways)
  const int nt = cpw4 / (4 * U);   // tiles of U steps, 4 float4 (16 floats) of a row per step
  f32x4 v1[U], v3[U];
  auto load_tile = [&](int p, int t) __attribute__((always_inline)) {
    const f32x4* r1 = (const f32x4*)((const float*)a.w1.w + (size_t)(16 * p + m) * dim) + wave * cpw4 + t * 4 * U + kq;
    const f32x4* r3 = (const f32x4*)((const float*)a.w3.w + (size_t)(16 * p + m) * dim) + wave * cpw4 + t * 4 * U + kq;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v1[u] = ld_nt(r1 + 4 * u);
      v3[u] = ld_nt(r3 + 4 * u);
    }
  };
  int p = blockIdx.x;
  load_tile(p < items ? p : 0, 0);  // unconditional: a branch around the loads costs the exact vmcnt ladder (kh_gemv.h)
  const float rs = st.finish(xs, a.eps, red);
  if (p >= items) return;  // uniform over the workgroup
  // ONE loop over (item, tile) with scalar control, as gemv_pairs: the next tile - of this item or of the next - is
  // requested in a burst right behind the current tile's MFMAs
  f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c3 = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0;;) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f32x4 xq = xs[wave * cpw4 + t * 4 * U + 4 * u + kq];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        c1 = mfma16(comp(v1[u], s), comp(xq, s), c1);
        c3 = mfma16(comp(v3[u], s), comp(xq, s), c3);
      }
    }
    const int tn = t + 1 < nt ? t + 1 : 0;
    const int pn = tn ? p : p + (int)gridDim.x;
    if (pn < items) load_tile(pn, tn);
    if (tn == 0) {
      // D[i][j]: lane = j + 16 * (i / 4), register i % 4; every column j holds the same value (x was broadcast)
      if (m == 0) {
        float* cw = comb + wave * 32;
        cw[4 * kq + 0] = c1.x; cw[4 * kq + 1] = c1.y; cw[4 * kq + 2] = c1.z; cw[4 * kq + 3] = c1.w;
        cw[16 + 4 * kq + 0] = c3.x; cw[16 + 4 * kq + 1] = c3.y; cw[16 + 4 * kq + 2] = c3.z; cw[16 + 4 * kq + 3] = c3.w;
      }
      __syncthreads();
      if (threadIdx.x < 16) {
        const int r = threadIdx.x;
        const float s1 = ((comb[r] + comb[32 + r]) + comb[64 + r]) + comb[96 + r];
        const float s3 = ((comb[16 + r] + comb[48 + r]) + comb[80 + r]) + comb[112 + r];
        a.h[16 * p + r] = swiglu1(rs * s1, rs * s3);
      }
      __syncthreads();
      if (pn >= items) break;
      c1 = f32x4{0.f, 0.f, 0.f, 0.f};
      c3 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    p = pn;
    t = tn;
  }
}

// ---- MFMA4: 4 rows per wave, the 16 blocks = 16 consecutive k-quads ---------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float add_dpp(float v) { return v + dpp_f32<CTRL>(v); }
// sum over the 16 lanes that share lane % 4 (the 16 blocks): row_ror:4, row_ror:8 inside a row of 16, then rows, then halves
__device__ __forceinline__ float sum_blocks(float v) {
  v = add_dpp<0x124>(v);
  v = add_dpp<0x128>(v);
  v = xor16_sum(v);
  return xor32_sum(v);
}
__global__ __launch_bounds__(256) void k_ffn13_mfma4(const KhFfn13Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  const int dim = a.dim;
  float* red = (float*)(xs + (dim >> 2));
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int i = lane & 3, b = lane >> 2;
  Stager<true, false, 2> st(a.x, a.ffn_norm, dim);
  st.issue();
  __builtin_amdgcn_sched_barrier(0);
  const int items = a.hidden >> 2;
  const int nt = (dim >> 2) / (16 * U);  // a step = 16 float4 (64 floats) of each of the 4 rows
  f32x4 v1[U], v3[U];
  auto load_tile = [&](int p, int t) __attribute__((always_inline)) {
    const f32x4* r1 = (const f32x4*)((const float*)a.w1.w + (size_t)(4 * p + i) * dim) + t * 16 * U + b;
    const f32x4* r3 = (const f32x4*)((const float*)a.w3.w + (size_t)(4 * p + i) * dim) + t * 16 * U + b;
#pragma unroll
    for (int u = 0; u < U; ++u) {
      v1[u] = ld_nt(r1 + 16 * u);
      v3[u] = ld_nt(r3 + 16 * u);
    }
  };
  const int np = gridDim.x * 4;
  int p = blockIdx.x * 4 + wave;
  load_tile(p < items ? p : 0, 0);
  const float rs = st.finish(xs, a.eps, red);
  if (p >= items) return;  // uniform per wave; no barrier below
  f32x4 c1 = {0.f, 0.f, 0.f, 0.f}, c3 = {0.f, 0.f, 0.f, 0.f};
  for (int t = 0;;) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const f32x4 xq = xs[t * 16 * U + 16 * u + b];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        c1 = mfma4(comp(v1[u], s), comp(xq, s), c1);
        c3 = mfma4(comp(v3[u], s), comp(xq, s), c3);
      }
    }
    const int tn = t + 1 < nt ? t + 1 : 0;
    const int pn = tn ? p : p + np;
    if (pn < items) load_tile(pn, tn);
    if (tn == 0) {
      // D_b[r][j]: lane = j + 4 b, register r; equal over j.  Sum the 16 blocks' partials of every row.
      const float s10 = sum_blocks(c1.x), s11 = sum_blocks(c1.y), s12 = sum_blocks(c1.z), s13 = sum_blocks(c1.w);
      const float s30 = sum_blocks(c3.x), s31 = sum_blocks(c3.y), s32 = sum_blocks(c3.z), s33 = sum_blocks(c3.w);
      if (lane == 0) {
        f32x4 o;
        o.x = swiglu1(rs * s10, rs * s30);
        o.y = swiglu1(rs * s11, rs * s31);
        o.z = swiglu1(rs * s12, rs * s32);
        o.w = swiglu1(rs * s13, rs * s33);
        *(f32x4*)(a.h + 4 * p) = o;
      }
      if (pn >= items) break;
      c1 = f32x4{0.f, 0.f, 0.f, 0.f};
      c3 = f32x4{0.f, 0.f, 0.f, 0.f};
    }
    p = pn;
    t = tn;
  }
}

// ---------------------------------------------------------------------------------------------------------------------
static float frand(uint32_t& s) {
  s = s * 1664525u + 1013904223u;
  return ((s >> 8) & 0xffff) / 65536.0f - 0.5f;
}

int main(int argc, char** argv) {
  const int NL = argc > 1 ? atoi(argv[1]) : 16, REPLAYS = argc > 2 ? atoi(argv[2]) : 20, ONLY = argc > 3 ? atoi(argv[3]) : 0;
  const int dim = 2048, hidden = 8192;
  const size_t wn = (size_t)hidden * dim;
  hipStream_t S;
  CK(hipStreamCreate(&S));
  float *x, *g, *w, *h[3];
  CK(hipMalloc(&x, dim * 4));
  CK(hipMalloc(&g, dim * 4));
  CK(hipMalloc(&w, (size_t)NL * 2 * wn * 4));
  for (auto& q : h) CK(hipMalloc(&q, (size_t)NL * hidden * 4));
  {
    std::vector<float> hx(dim), hg(dim), hw(2 * wn);
    uint32_t s = 12345;
    for (auto& v : hx) v = 2.f * frand(s);
    for (auto& v : hg) v = 1.f + 0.2f * frand(s);
    for (auto& v : hw) v = 0.1f * frand(s);
    CK(hipMemcpy(x, hx.data(), dim * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(g, hg.data(), dim * 4, hipMemcpyHostToDevice));
    for (int l = 0; l < NL; ++l) {  // every slab its own values (slab l = slab 0 rotated by l rows)
      const size_t rot = (size_t)l * dim;
      CK(hipMemcpy(w + (size_t)l * 2 * wn, hw.data() + rot, (2 * wn - rot) * 4, hipMemcpyHostToDevice));
      if (rot) CK(hipMemcpy(w + (size_t)l * 2 * wn + (2 * wn - rot), hw.data(), rot * 4, hipMemcpyHostToDevice));
    }
  }
  auto args = [&](int l, float* out) {
    KhFfn13Args a;
    a.x = x;
    a.ffn_norm = g;
    a.w1 = KhLin{w + (size_t)l * 2 * wn, nullptr, nullptr};
    a.w3 = KhLin{w + (size_t)l * 2 * wn + wn, nullptr, nullptr};
    a.h = out + (size_t)l * hidden;
    a.dim = dim;
    a.hidden = hidden;
    a.gshift = 0;
    a.eps = 1e-5f;
    return a;
  };
  const size_t lds_valu = fused_lds_bytes(false, dim);
  const size_t lds16 = (size_t)dim * 4 + 8 * 4 + 4 * 32 * 4, lds4 = (size_t)dim * 4 + 8 * 4;
  auto launch = [&](int which, int l) {
    if (which == 0)
      hipLaunchKernelGGL((k_ffn13<false, 8, 2>), dim3(512), dim3(256), lds_valu, S, args(l, h[0]));
    else if (which == 1)
      hipLaunchKernelGGL(k_ffn13_mfma16, dim3(hidden / 16), dim3(256), lds16, S, args(l, h[1]));
    else
      hipLaunchKernelGGL(k_ffn13_mfma4, dim3(hidden / 16), dim3(256), lds4, S, args(l, h[2]));
  };
  auto time_graph = [&](int which) {
    hipGraph_t gr;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
    for (int l = 0; l < NL; ++l) launch(which, l);
    CK(hipStreamEndCapture(S, &gr));
    CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    CK(hipGraphLaunch(ge, S));
    CK(hipStreamSynchronize(S));
    float best = 1e30f;
    for (int r = 0; r < REPLAYS; ++r) {
      CK(hipEventRecord(e0, S));
      CK(hipGraphLaunch(ge, S));
      CK(hipEventRecord(e1, S));
      CK(hipEventSynchronize(e1));
      float ms;
      CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    CK(hipGraphExecDestroy(ge));
    CK(hipGraphDestroy(gr));
    return best * 1e3f / NL;
  };
  const double bytes = 2.0 * wn * 4 + 2.0 * dim * 4 + hidden * 4.0;
  const char* names[3] = {"VALU   k_ffn13<false, 8, 2> 512 x 256 (shipped)", "MFMA16 v_mfma_f32_16x16x4_f32, 16 rows / workgroup, K over 4 waves",
                          "MFMA4  v_mfma_f32_4x4x1_16B_f32, 4 rows / wave, blocks = k-quads"};
  printf("Llama-3.2-1B ffn13: w1, w3 [%d, %d] fp32, %.1f MB per launch; %d distinct slabs per graph, best of %d replays\n", hidden, dim,
         bytes / 1e6, NL, REPLAYS);
  std::vector<float> ref((size_t)NL * hidden), out((size_t)NL * hidden);
  for (int which = 0; which < 3; ++which) {
    if (ONLY && ONLY != which + 1) continue;
    const float us = time_graph(which);
    CK(hipMemcpy(out.data(), h[which], out.size() * 4, hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    if (which == 0) ref = out;
    for (size_t k = 0; k < out.size(); ++k) {
      maxerr = fmax(maxerr, fabs((double)out[k] - ref[k]));
      maxref = fmax(maxref, fabs((double)ref[k]));
    }
    printf("%-70s %7.2f us  %5.3f of 8 TB/s", names[which], us, bytes / (us * 1e-6) / 8e12);
    if (which == 0 || ONLY)
      printf("\n");
    else
      printf("   max |h - h_valu| %.2e (max |h| %.2e)\n", maxerr, maxref);
  }
  return 0;
}
