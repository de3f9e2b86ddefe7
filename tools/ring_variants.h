// ring_variants.h - the parts of the round-5 LDS-DMA ring work that the product does NOT launch, kept with the
// microbenchmark that measures them (tools/mb_q8ring.hip; profiles/r5_int8_ring_ab.txt):
//   StagerDma        the input vector staged through the DMA path instead of asm register loads (measured equal)
//   k_gemv_res_ring  wo / w2 on the ring core   } bit-identical to the register-tile kernels, tie or lose by
//   k_qkv_ring       qkv on the ring core       } 0.1-0.3 us inside the model: the product keeps k_gemv_res / k_qkv
// The ring core itself (kh_q8ring.h) and the two adopted kernels (kh_fused_ring.h: k_ffn13_ring, k_cls_ring) are
// product headers.
#pragma once
#include <type_traits>

#include "kh_fused_ring.h"

// the same with the default cache policy (the activation vector: every workgroup reads it, it lives in L2)
__device__ __forceinline__ void dma_x4_keep(const void* base, unsigned voff, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
               :
               : "v"(voff), "s"(base), "s"(dst)
               : "memory", "m0");  // see kh_q8ring.h::dma_x4 (built with -w: the reserved-register remark is off)
}

// Input vector staging through the DMA path - the alternative to StagerAsm, measured equal within 0.5 % on all five
// kernels (profiles/r5_int8_ring_ab.txt section 6) and not used by the product: the waves pull the raw vector into the
// xs area (linear) and the norm weight into its own area with 1-KiB pieces, issued BEFORE the ring prologue, finish()
// waits for exactly the prologue's operation count, then permutes (and normalises) in place through registers: the
// element -> thread mapping and the arithmetic are Stager<NORM, true, MAXV>'s, so the staged vector is bit-identical.
// Costs two more barriers, an LDS round trip and M floats of LDS for the norm weight; needs M % 256 == 0 (whole
// 1-KiB pieces).  (It was written when the first ring version seemed to stage wrong vectors through register loads;
// that was the microbenchmark overwriting its own reference output - tools/mb_vmcnt_order.hip shows that register
// loads and LDS-DMA operations do retire in issue order through vmcnt, and StagerAsm is bit-identical.)
// VT: the element -> thread mapping and the norm's reduction tree are those of a VT-thread workgroup (0: the real
// width); threads past VT only take part in the barriers.  Lets a workgroup of any width (11 waves, ...) stage
// exactly what the 256-thread kernels stage.
template <bool NORM, int MAXV, int VT = 0>
struct StagerDma {
  static constexpr bool kRawArea = NORM;  // the raw norm weight sits between the reduction words and the rings
  const float* x;
  const float* wnorm;
  f32x4* xs;         // q8-layout area; receives the raw vector first
  const f32x4* wraw;  // NORM: raw norm weight area (M floats)
  int M;
  __device__ __forceinline__ StagerDma(const float* x_, const float* wnorm_, f32x4* xs_, const void* wraw_, int M_)
      : x(x_), wnorm(wnorm_), xs(xs_), wraw((const f32x4*)wraw_), M(M_) {}
  __device__ __forceinline__ void issue() {
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = kh_nwaves();
    const unsigned lane16 = (threadIdx.x & 63u) << 4;
    const unsigned xs0 = (unsigned)__builtin_amdgcn_readfirstlane((int)kh_lds_addr(xs));
    const unsigned wr0 = (unsigned)__builtin_amdgcn_readfirstlane((int)kh_lds_addr(wraw));
    const int npieces = M >> 8;
    for (int pc = wave; pc < npieces; pc += nw) {
      dma_x4_keep(x, (unsigned)pc * 1024u + lane16, xs0 + (unsigned)pc * 1024u);
      if (NORM) dma_x4_keep(wnorm, (unsigned)pc * 1024u + lane16, wr0 + (unsigned)pc * 1024u);
    }
  }
  // [r6] as Stager / StagerAsm: g = w_norm * x into LDS, the RMS scale returned for the epilogue (kh_gemv.h)
  template <int YOUNGER>
  __device__ __forceinline__ float finish(float eps, float* red, bool exact) {
    if (exact)
      wait_vm<YOUNGER>();
    else
      wait_vm<0>();
    __syncthreads();  // every wave's pieces have landed
    const int M4 = M >> 2, M16 = M >> 4;
    const int wgv = VT ? VT : kh_wg();
    const bool act = (int)threadIdx.x < wgv;
    f32x4 xv[MAXV], wv[NORM ? MAXV : 1];
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int i = threadIdx.x + v * wgv;
      const int ci = (act && i < M4) ? i : 0;
      xv[v] = xs[ci];
      if (NORM) wv[v] = wraw[ci];
    }
    float ss = 0.f;
    if (NORM) {
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const float t = fma4(xv[v], xv[v], 0.f);
        ss += (act && (int)threadIdx.x + v * wgv < M4) ? t : 0.f;
      }
    }
    __syncthreads();  // the raw reads above are done before the permuted writes below
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int i = threadIdx.x + v * wgv;
      if (act && i < M4) {
        f32x4 t = xv[v];
        if (NORM) {
          t.x = wv[v].x * t.x;
          t.y = wv[v].y * t.y;
          t.z = wv[v].z * t.z;
          t.w = wv[v].w * t.w;
        }
        xs[q8_slot(i, M16)] = t;
      }
    }
    if (NORM) {
      // stage_rs over the wgv / 64 waves that staged (threads past VT contribute 0 and write nothing)
      ss = wave_sum(ss);
      const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nvw = wgv >> 6;
      if (lane == 0 && wave < nvw) red[wave] = ss;
      __syncthreads();
      float r = 0.f;
#pragma unroll
      for (int w = 0; w < KH_WAVES_MAX; ++w) r += w < nvw ? red[w < nvw ? w : 0] : 0.f;
      return 1.0f / sqrtf(r / (float)M + eps);
    }
    __syncthreads();
    return 1.f;
  }
};


// STG: 0 = the vector staged through the DMA path (StagerDma), 1 = register loads by asm (StagerAsm)
template <bool NORM, int MAXV, int VT, int STG>
using RingStager = typename std::conditional<STG == 1, StagerAsm<NORM, MAXV, VT>, StagerDma<NORM, MAXV, VT>>::type;

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
      [&](bool exact) __attribute__((always_inline)) { (void)st.template finish<R * 4>(0.f, red, exact); }, epi);
  KH_STAMP_FLUSH();
}

// stage g = att_norm * x -> [wq|wk|wv] row pairs -> * rs -> +bias -> RoPE -> q / cache row `pos` (k_qkv<true, ...> on
// the ring core).  The epilogue operands of a pair - sin, cos of its cache column and the two bias values - come
// through the scalar cache (ld_uniform: no vector load while the ring is live).  [r6] with the one-barrier staging this
// kernel beats k_qkv by 2.6-2.8 % in the microbenchmark (profiles/r6_q8_phases.txt: 10.45 vs 10.75 us) - and not inside
// the model (r6_qkv_ring_ab.txt: 11.0 vs 11.1 us back to back, 603.3 vs 604.2 tok/s): built into the product behind
// plan_ring for that A/B (commit history), measured, and taken out again.
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
  float rs = 1.f;
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
    s0 = rs * s0 + x.b0;  // the RMS scale of the staged vector, as k_qkv
    s1 = rs * s1 + x.b1;
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
      [&](bool exact) __attribute__((always_inline)) { rs = st.template finish<R * 4>(eps, red, exact); }, epi);
  KH_STAMP_FLUSH();
}
