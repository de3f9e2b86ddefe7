// kh_fused_ring.h — the int8 decode-step GEMV kernels of kh_fused.h on the LDS-DMA ring core (kh_q8ring.h).
// Same argument structs, same work decomposition, same per-lane arithmetic and epilogues as k_ffn13 / k_cls /
// k_gemv_res / k_qkv<QUANT = true>: outputs are bit-identical, only the path of the weight bytes differs
// (HBM -> LDS ring by DMA -> ds_read_b128 instead of HBM -> VGPR tiles).
//   R        ring slots per wave (piece pairs of 2 x 1 KiB + scales)
//   MAXV     float4 of the input vector per staging thread (as Stager)
//   BLOCKED  contiguous items per workgroup instead of gemv_pairs' interleaved mapping (kh_q8ring.h)
//   VT       width of the workgroup whose staging is reproduced (0: the real width)
#pragma once
#include "kh_fused.h"
#include "kh_q8ring.h"

// STG: 0 = the vector staged through the DMA path (StagerDma), 1 = register loads by asm (StagerAsm)
template <bool NORM, int MAXV, int VT, int STG>
using RingStager = typename std::conditional<STG == 1, StagerAsm<NORM, MAXV, VT>, StagerDma<NORM, MAXV, VT>>::type;

template <int R, int MAXV, bool BLOCKED = false, int VT = 0, int STG = 0>
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
  RingStager<true, MAXV, VT, STG> st(a.x, a.ffn_norm, xs, smem_raw + ring_lds_wraw_off(dim), dim);
  auto pair = [&](int r) __attribute__((always_inline)) { return g.rows(w1, r, w3, r, s1p, s3p, dim); };
  auto epi = [&](int r, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (lane == 0) h[r] = swiglu1(s0, s1);
  };
  ring_pairs<1, R, BLOCKED>(
      dim, a.gshift, xs, a.hidden, lane, nullptr, smem_raw + ring_lds_off(dim, STG == 0), pair,
      [](int) __attribute__((always_inline)) { return NoAux{}; },
      [&]() __attribute__((always_inline)) { st.issue(); },
      [&](bool exact) __attribute__((always_inline)) { st.template finish<R * 4>(eps, red, exact); }, epi);
  KH_STAMP_FLUSH();
}

template <int R, int MAXV, bool BLOCKED = false, int VT = 0, int STG = 0>
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
  RingStager<true, MAXV, VT, STG> st(a.x, a.final_norm, xs, smem_raw + ring_lds_wraw_off(dim), dim);
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  auto r1_of = [&](int p) __attribute__((always_inline)) { return 2 * p + 1 < vocab ? 2 * p + 1 : 2 * p; };
  auto pair = [&](int p) __attribute__((always_inline)) { return g.rows(w, 2 * p, w, r1_of(p), sc, sc, dim); };
  auto epi = [&](int p, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (lane != 0) return;
    const int r0 = 2 * p, r1 = r1_of(p);
    logits[r0] = s0;
    amax_merge(bv, bi, s0, r0);
    if (r1 != r0) {
      logits[r1] = s1;
      amax_merge(bv, bi, s1, r1);
    }
  };
  ring_pairs<1, R, BLOCKED>(
      dim, a.gshift, xs, (vocab + 1) >> 1, lane, nullptr, smem_raw + ring_lds_off(dim, STG == 0), pair,
      [](int) __attribute__((always_inline)) { return NoAux{}; },
      [&]() __attribute__((always_inline)) { st.issue(); },
      [&](bool exact) __attribute__((always_inline)) { st.template finish<R * 4>(eps, red, exact); }, epi);
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

// y = W . v ; x += y (wo, w2).  The residual words x[2p], x[2p+1] come through the scalar cache (ld_uniform).
template <int R, int MAXV, int SPLIT, int STG = 0>
__global__ __launch_bounds__(1024) void k_gemv_res_ring(const KhGemvResArgs a) {
  extern __shared__ __attribute__((aligned(256))) char smem_raw[];
  KH_STAMP_INIT();
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<true>(xs, a.M);
  const int lane = threadIdx.x & 63;
  const int M = a.M;
  const void* const w = a.w.w;
  const float* const scales = a.w.scales;
  float* const x = a.x;
  const Gemv<true, 1> g(M, a.gshift);
  RingStager<false, MAXV, 0, STG> st(a.vec, nullptr, xs, nullptr, M);
  auto pair = [&](int p) __attribute__((always_inline)) { return g.rows(w, 2 * p, w, 2 * p + 1, scales, scales, M); };
  struct Aux {
    float x0, x1;
  };
  auto auxf = [&](int p) __attribute__((always_inline)) { return Aux{ld_uniform(x + 2 * p), ld_uniform(x + 2 * p + 1)}; };
  auto epi = [&](int p, float s0, float s1, const Aux& r) __attribute__((always_inline)) {
    if (lane != 0) return;
    x[2 * p] = r.x0 + s0;
    x[2 * p + 1] = r.x1 + s1;
  };
  ring_pairs<SPLIT, R, false>(
      M, a.gshift, xs, a.K >> 1, lane, red + KH_WAVES_MAX, smem_raw + ring_lds_off(M, false), pair, auxf,
      [&]() __attribute__((always_inline)) { st.issue(); },
      [&](bool exact) __attribute__((always_inline)) { st.template finish<R * 4>(0.f, red, exact); }, epi);
  KH_STAMP_FLUSH();
}

// RMSNorm(x) -> [wq|wk|wv] row pairs -> +bias -> RoPE -> q / cache row `pos` (k_qkv<true, ...>).  The epilogue
// operands of a pair - sin, cos of its cache column and the two bias values - come through the scalar cache.
template <int R, int MAXV, int SPLIT, int STG = 0>
__global__ __launch_bounds__(1024) void k_qkv_ring(const KhQkvArgs a) {
  extern __shared__ __attribute__((aligned(256))) char smem_raw[];
  KH_STAMP_INIT();
  const void *wq_w = a.wq.w, *wk_w = a.wk.w, *wv_w = a.wv.w;
  const float *wq_s = a.wq.scales, *wk_s = a.wk.scales, *wv_s = a.wv.scales;
  const float *wq_b = a.wq.bias, *wk_b = a.wk.bias, *wv_b = a.wv.bias;
  float* const q_out = a.q_out;
  float* const kc = a.kcache_layer;
  float* const vc = a.vcache_layer;
  const float* const sin_cache = a.sin_cache;
  const float* const cos_cache = a.cos_cache;
  const int dim = a.dim, kv_dim = a.kv_dim, rope_mode = a.rope_mode;
  const float eps = a.eps;
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<true>(xs, dim);
  const int lane = threadIdx.x & 63;
  const int hs = a.head_size, half = hs >> 1;
  const bool half_pow2 = (half & (half - 1)) == 0;
  const int half_sh = __builtin_ctz((unsigned)half | 0x40000000u);
  const int npq = dim >> 1, npk = kv_dim >> 1;
  const int total = npq + 2 * npk;
  const Gemv<true, 1> g(dim, a.gshift);
  RingStager<true, MAXV, 0, STG> st(a.x, a.att_norm, xs, smem_raw + ring_lds_wraw_off(dim), dim);
  const int pos = *a.d_pos;  // scalar load: lgkmcnt, not vmcnt

  auto decode = [&](int p, int& which, int& r0, int& r1, int& cidx) __attribute__((always_inline)) {
    int pp;
    if (p < npq) {
      which = 0;
      pp = p;
    } else if (p < npq + npk) {
      which = 1;
      pp = p - npq;
    } else {
      which = 2;
      pp = p - npq - npk;
    }
    if (which < 2 && rope_mode == KH_ROPE_HALF) {
      const int head = half_pow2 ? pp >> half_sh : pp / half, j = pp - head * half;
      r0 = head * hs + j;
      r1 = r0 + half;
      cidx = 2 * j;
    } else {
      r0 = 2 * pp;
      r1 = r0 + 1;
      cidx = half_pow2 ? r0 & (hs - 1) : r0 % hs;
    }
  };
  auto pair = [&](int p) __attribute__((always_inline)) {
    int which, r0, r1, cidx;
    decode(p, which, r0, r1, cidx);
    const void* w = sel3(which, wq_w, wk_w, wv_w);
    const float* sc = sel3(which, wq_s, wk_s, wv_s);
    return g.rows(w, r0, w, r1, sc, sc, dim);
  };
  struct Aux {
    float fci, fcr, b0, b1;
  };
  auto auxf = [&](int p) __attribute__((always_inline)) {
    int which, r0, r1, cidx;
    decode(p, which, r0, r1, cidx);
    const float* bias = sel3(which, wq_b, wk_b, wv_b);
    Aux x;
    x.fci = ld_uniform(sin_cache + (size_t)pos * hs + cidx);
    x.fcr = ld_uniform(cos_cache + (size_t)pos * hs + cidx);
    x.b0 = bias ? ld_uniform(bias + r0) : 0.f;
    x.b1 = bias ? ld_uniform(bias + r1) : 0.f;
    return x;
  };
  auto epi = [&](int p, float s0, float s1, const Aux& x) __attribute__((always_inline)) {
    if (lane != 0) return;
    int which, r0, r1, cidx;
    decode(p, which, r0, r1, cidx);
    s0 = s0 + x.b0;
    s1 = s1 + x.b1;
    float* dst = sel3(which, q_out, kc + (size_t)pos * kv_dim, vc + (size_t)pos * kv_dim);
    if (which < 2) {
      const float v0 = s0, v1 = s1;
      s0 = v0 * x.fcr - v1 * x.fci;
      s1 = v0 * x.fci + v1 * x.fcr;
    }
    dst[r0] = s0;
    dst[r1] = s1;
  };
  ring_pairs<SPLIT, R, false>(
      dim, a.gshift, xs, total, lane, red + KH_WAVES_MAX, smem_raw + ring_lds_off(dim, STG == 0), pair, auxf,
      [&]() __attribute__((always_inline)) { st.issue(); },
      [&](bool exact) __attribute__((always_inline)) { st.template finish<R * 4>(eps, red, exact); }, epi);
  KH_STAMP_FLUSH();
}
