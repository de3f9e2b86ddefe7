// kh_gemv.h — batch-1 GEMV cores for gfx950.
//
// Shape of the problem: y[K] = W[K,M] . x[M], W row-major as exported in the
// .bin (tools/export.py:79-131), streamed once per token => HBM-bound, 0.5 flop/B (fp32).
//
// Mapping: one WAVE owns a pair of rows at a time and streams them with 16-byte-per-lane
// non-temporal loads (64 lanes x 16 B = 1 KiB contiguous per instruction), U instructions per
// row in flight before the first FMA (2*U KiB per wave).  The activation vector lives in LDS
// (staged once per workgroup, optionally RMS-normalised on the way in); LDS read bandwidth is
// ~25x the per-CU HBM rate so re-reading x per row is free.  No LDS round trip for the
// weights (MI355X guide: "GEMV / M <= 16 decode weights ... load straight to VGPRs, deep
// unroll, late vmcnt").  The only cross-lane step is one 6-stage wave reduction per row pair.
//
// Row PAIRS are the unit because every fused epilogue consumes two outputs together:
// RoPE rotates (v0,v1), SwiGLU combines (w1.x, w3.x), and plain rows just take (2i, 2i+1).
#pragma once
#include "kh_common.h"

// ---------------------------------------------------------------------------------------------
// fp32 rows.  w0/w1: row base pointers (16-B aligned), xs: x in LDS as float4[M4].
// Returns the two dot products in every lane.
template <int U>
__device__ __forceinline__ void dot2_f32(const f32x4* __restrict__ w0,
                                         const f32x4* __restrict__ w1, const f32x4* xs, int M4,
                                         int lane, float& s0, float& s1) {
  float a0 = 0.f, a1 = 0.f;
  for (int c0 = 0; c0 < M4; c0 += KH_WAVE * U) {
    f32x4 v0[U], v1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = c0 + u * KH_WAVE + lane;
      const int cidx = idx < M4 ? idx : 0;  // clamped address, masked below
      v0[u] = ld_nt(w0 + cidx);
      v1[u] = ld_nt(w1 + cidx);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = c0 + u * KH_WAVE + lane;
      if (idx < M4) {
        const f32x4 xv = xs[idx];
        a0 = fma4(v0[u], xv, a0);
        a1 = fma4(v1[u], xv, a1);
      }
    }
  }
  s0 = wave_sum(a0);
  s1 = wave_sum(a1);
}

// ---------------------------------------------------------------------------------------------
// int8 group-quantised rows (tools/export.py:134-210: int8[K*M] then fp32 scales[K*M/g]).
// w0/w1: row base pointers as dwordx4 (16 weights per lane per load); sc0/sc1: pointer to the
// scale of the row's first group (valid because M % group == 0 on this path); gshift =
// log2(group_size) (group is a power of two >= 16 on this path, so one lane's 16 weights
// share one scale).  xs: LDS in q8_slot() layout.  Dequant factored per 16-weight run:
//   sum_i x_i * s_g * w_i  ==  s_g * sum_i x_i * w_i      (reference: cuda/matmul_kernel.cu:73)
template <int U>
__device__ __forceinline__ void dot2_q8(const i32x4* __restrict__ w0,
                                        const i32x4* __restrict__ w1,
                                        const float* __restrict__ sc0,
                                        const float* __restrict__ sc1, int gshift,
                                        const f32x4* xs, int M16, int lane, float& s0,
                                        float& s1) {
  float a0 = 0.f, a1 = 0.f;
  const int plane = M16 + 1;
  for (int c0 = 0; c0 < M16; c0 += KH_WAVE * U) {
    i32x4 q0[U], q1[U];
    float g0[U], g1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = c0 + u * KH_WAVE + lane;
      const int cidx = idx < M16 ? idx : 0;
      q0[u] = ld_nt(w0 + cidx);
      q1[u] = ld_nt(w1 + cidx);
      const int gi = (cidx << 4) >> gshift;
      g0[u] = sc0[gi];
      g1[u] = sc1[gi];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int idx = c0 + u * KH_WAVE + lane;
      if (idx < M16) {
        const f32x4 x0 = xs[idx], x1 = xs[plane + idx], x2 = xs[2 * plane + idx],
                    x3 = xs[3 * plane + idx];
        float t0 = 0.f, t1 = 0.f;
        t0 = dot4_i8(q0[u].x, x0, t0);
        t0 = dot4_i8(q0[u].y, x1, t0);
        t0 = dot4_i8(q0[u].z, x2, t0);
        t0 = dot4_i8(q0[u].w, x3, t0);
        t1 = dot4_i8(q1[u].x, x0, t1);
        t1 = dot4_i8(q1[u].y, x1, t1);
        t1 = dot4_i8(q1[u].z, x2, t1);
        t1 = dot4_i8(q1[u].w, x3, t1);
        a0 = __builtin_fmaf(g0[u], t0, a0);
        a1 = __builtin_fmaf(g1[u], t1, a1);
      }
    }
  }
  s0 = wave_sum(a0);
  s1 = wave_sum(a1);
}

// ---------------------------------------------------------------------------------------------
// Stage a vector into LDS (all 256 threads), optionally RMS-normalising it on the way:
//   xs = w_norm * (x * 1/sqrt(mean(x^2)+eps))           (cpu/rmsnorm_kernel.cpp:24-32)
// Every workgroup recomputes the norm redundantly from the L2-resident x (8-16 KiB): cheaper
// than a separate single-block launch + kernel boundary (reference: row_rmsnorm_f32, 1 block).
// LAYOUT_Q8 selects the q8_slot() arrangement.  red = LDS float[4].
template <bool NORM, bool LAYOUT_Q8>
__device__ __forceinline__ void stage_vec(const float* __restrict__ x,
                                          const float* __restrict__ wnorm, f32x4* xs, int M,
                                          float eps, float* red) {
  const int M4 = M >> 2;
  const f32x4* x4 = (const f32x4*)x;
  float rs = 1.f;
  if (NORM) {
    float ss = 0.f;
    for (int i = threadIdx.x; i < M4; i += KH_WG) {
      const f32x4 v = x4[i];
      ss = fma4(v, v, ss);
    }
    ss = block_sum(ss, red);
    const float mean = ss / (float)M + eps;
    rs = 1.0f / sqrtf(mean);
  }
  const f32x4* w4 = (const f32x4*)wnorm;
  const int M16 = M >> 4;
  for (int i = threadIdx.x; i < M4; i += KH_WG) {
    f32x4 v = x4[i];
    if (NORM) {
      const f32x4 w = w4[i];
      v.x = w.x * (rs * v.x);
      v.y = w.y * (rs * v.y);
      v.z = w.z * (rs * v.z);
      v.w = w.w * (rs * v.w);
    }
    xs[LAYOUT_Q8 ? q8_slot(i, M16) : i] = v;
  }
  __syncthreads();
}
