// kh_fused_ring.h — the int8 decode-step GEMV kernels of kh_fused.h on the LDS-DMA ring core (kh_q8ring.h).
// Same argument structs, same work decomposition, same per-lane arithmetic and epilogues as k_ffn13 / k_cls /
// k_gemv_res / k_qkv<QUANT = true>: outputs are bit-identical, only the path of the weight bytes differs
// (HBM -> LDS ring by DMA -> ds_read_b128 instead of HBM -> VGPR tiles).
//   R        ring slots per wave (piece pairs of 2 x 1 KiB + scales)
//   MAXV     float4 of the input vector per staging thread (as Stager)
//   BLOCKED  contiguous items per workgroup instead of gemv_pairs' interleaved mapping (kh_q8ring.h)
//   ST       the staging of the input vector (StagerAsm; tools/ring_variants.h plugs in StagerDma)
// The product launches k_ffn13_ring<2, 4> and k_cls_ring<2, 4> (kh_model_step.hip); the ring forms of wo / w2 / qkv,
// which tie or lose against the register tiles, live with the microbenchmark in tools/ring_variants.h.
#pragma once
#include "kh_fused.h"
#include "kh_q8ring.h"

template <int R, int MAXV, bool BLOCKED = false, class ST = StagerAsm<true, MAXV, 0>>
__global__ __launch_bounds__(1024) void k_ffn13_ring(const KhFfn13Args a) {
  extern __shared__ __attribute__((aligned(256))) char smem_raw[];
  KH_STAMP_INIT();
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<true>(xs, a.dim);
  const int lane = threadIdx.x & 63;
  const int dim = a.dim;
  const void *w1 = a.w1.w, *w3 = a.w3.w;
  const float *s1p = a.w1.scales, *s3p = a.w3.scales;
  float* const h = a.h;
  const float eps = a.eps;
  const Gemv<true, 1> g(dim, a.gshift);
  ST st(a.x, a.ffn_norm, xs, smem_raw + ring_lds_wraw_off(dim), dim);
  auto pair = [&](int r) __attribute__((always_inline)) { return g.rows(w1, r, w3, r, s1p, s3p, dim); };
  float rs = 1.f;  // RMS scale of x: set by the staging, applied in the epilogue (as k_ffn13)
  auto epi = [&](int r, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (lane == 0) h[r] = swiglu1(rs * s0, rs * s1);
  };
  ring_pairs<1, R, BLOCKED>(
      dim, a.gshift, xs, a.hidden, lane, nullptr, smem_raw + ring_lds_off(dim, ST::kRawArea), pair,
      [](int) __attribute__((always_inline)) { return NoAux{}; },
      [&]() __attribute__((always_inline)) { st.issue(); },
      [&](bool exact) __attribute__((always_inline)) { rs = st.template finish<R * 4>(eps, red, exact); }, epi);
  KH_STAMP_FLUSH();
}

template <int R, int MAXV, bool BLOCKED = false, class ST = StagerAsm<true, MAXV, 0>>
__global__ __launch_bounds__(1024) void k_cls_ring(const KhClsArgs a) {
  extern __shared__ __attribute__((aligned(256))) char smem_raw[];
  KH_STAMP_INIT();
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<true>(xs, a.dim);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int dim = a.dim, vocab = a.vocab;
  const void* const w = a.wcls.w;
  const float* const sc = a.wcls.scales;
  float* const logits = a.logits;
  const float eps = a.eps;
  const Gemv<true, 1> g(dim, a.gshift);
  ST st(a.x, a.final_norm, xs, smem_raw + ring_lds_wraw_off(dim), dim);
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  auto r1_of = [&](int p) __attribute__((always_inline)) { return 2 * p + 1 < vocab ? 2 * p + 1 : 2 * p; };
  auto pair = [&](int p) __attribute__((always_inline)) { return g.rows(w, 2 * p, w, r1_of(p), sc, sc, dim); };
  float rs = 1.f;  // RMS scale of x: set by the staging, applied in the epilogue (as k_cls)
  auto epi = [&](int p, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (lane != 0) return;
    const int r0 = 2 * p, r1 = r1_of(p);
    s0 *= rs;
    s1 *= rs;
    logits[r0] = s0;
    amax_merge(bv, bi, s0, r0);
    if (r1 != r0) {
      logits[r1] = s1;
      amax_merge(bv, bi, s1, r1);
    }
  };
  ring_pairs<1, R, BLOCKED>(
      dim, a.gshift, xs, (vocab + 1) >> 1, lane, nullptr, smem_raw + ring_lds_off(dim, ST::kRawArea), pair,
      [](int) __attribute__((always_inline)) { return NoAux{}; },
      [&]() __attribute__((always_inline)) { st.issue(); },
      [&](bool exact) __attribute__((always_inline)) { rs = st.template finish<R * 4>(eps, red, exact); }, epi);
  // stage-1 argmax: one partial per workgroup (ties -> lowest index), as k_cls; red[] / comb[] are free again
  int* redi = (int*)(red + KH_WAVES_MAX);
  __syncthreads();
  if (lane == 0) {
    red[wave] = bv;
    redi[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = red[0];
    int i = redi[0];
    for (int w2 = 1, nw = kh_nwaves(); w2 < nw; ++w2) amax_merge(v, i, red[w2], redi[w2]);
    a.part_val[blockIdx.x] = v;
    a.part_idx[blockIdx.x] = i;
  }
  KH_STAMP_FLUSH();
}
