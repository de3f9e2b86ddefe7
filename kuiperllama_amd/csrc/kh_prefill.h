// kh_prefill.h — prompt prefill: several prompt tokens per pass over the weights.
//
// The reference feeds the prompt one token per forward pass (demo/main.cpp:20-22), so every
// prompt token streams all weights once.  Here B consecutive prompt tokens share ONE pass: the
// row-pair GEMV core keeps B (4 or 8) activation vectors in LDS and B accumulator pairs per wave, so the
// weight bytes per prompt token drop by B while the kernels stay HBM-bound (B = 4: 4 FMA per
// weight byte-quad, ~10 % of the VALU rate at the HBM rate).  No logits are produced for prompt
// tokens (the reference computes and discards them); what the prompt phase leaves behind — the
// K/V cache rows — is BIT-IDENTICAL to the token-by-token path:
//   * per (row, token) the dot product visits the columns in the same order (lane l takes
//     chunks l, l+64, l+128, ... of its column range whatever U is), with the same SPLIT
//     partition and the same fixed-order combine;
//   * RMS-norm statistics use the same per-thread partition and the same reduction order as
//     Stager of the decode kernels (same workgroup width), the vector is staged as w_norm * x and the
//     scale multiplies the dot product in the epilogue, as there;
//   * epilogues (bias, RoPE, SwiGLU, residual) are the same expressions;
//   * attention is the decode kernel itself, one grid slice per token.
// tests/test_model_gpu.py::test_prefill_* compares cache rows and the following logits bit for bit.
#pragma once
#include "kh_fused.h"

// Tokens per weight pass: up to 8 for the dim-input matrices of fp32 models when 8 vectors leave
// room for two workgroups per CU, else 4; w2 (hidden-sized input) takes the largest of 8/4/2 whose
// vectors fit LDS; int8 stays at 4 (its converted-weight tile already fills the register file).
#define KH_PF_BMAX 8
// 16-byte loads per row in flight per lane: int8 converts the whole tile to floats once per
// chunk (16 floats per load), so its tile is kept to 2 loads per row
#define KH_PF_U(QUANT) ((QUANT) ? 2 : 4)

// ---- B-token FMA of one register chunk --------------------------------------------------------
template <int U, int B>
__device__ __forceinline__ void fma_chunk_b(const RegsF32<U>& r, const f32x4* xs, int xstride,
                                            int c0, int M4, int lane, float (&a0)[B],
                                            float (&a1)[B]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int idx = c0 + u * KH_WAVE + lane;
    if (idx < M4) {
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const f32x4 xv = xs[(size_t)b * xstride + idx];
        a0[b] = fma4(r.v0[u], xv, a0[b]);
        a1[b] = fma4(r.v1[u], xv, a1[b]);
      }
    }
  }
}
// int8: the 16 weights of a load are converted to float ONCE per row and reused by all B tokens
// (the decode kernels convert inside dot4_i8; the value sequence fed to the FMA chain is the same:
// t = fma(w_0, x_0, 0), fma(w_1, x_1, t), ... over the 16 weights, then a = fma(scale, t, a)).
__device__ __forceinline__ void cvt16_i8(const i32x4& q, float (&w)[16]) {
  const int d[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    w[4 * k + 0] = (float)(int8_t)(d[k] & 0xff);
    w[4 * k + 1] = (float)(int8_t)((d[k] >> 8) & 0xff);
    w[4 * k + 2] = (float)(int8_t)((d[k] >> 16) & 0xff);
    w[4 * k + 3] = (float)(d[k] >> 24);
  }
}
__device__ __forceinline__ float dot16(const float (&w)[16], const f32x4& x0, const f32x4& x1,
                                       const f32x4& x2, const f32x4& x3) {
  float t = 0.f;
  t = __builtin_fmaf(w[0], x0.x, t);
  t = __builtin_fmaf(w[1], x0.y, t);
  t = __builtin_fmaf(w[2], x0.z, t);
  t = __builtin_fmaf(w[3], x0.w, t);
  t = __builtin_fmaf(w[4], x1.x, t);
  t = __builtin_fmaf(w[5], x1.y, t);
  t = __builtin_fmaf(w[6], x1.z, t);
  t = __builtin_fmaf(w[7], x1.w, t);
  t = __builtin_fmaf(w[8], x2.x, t);
  t = __builtin_fmaf(w[9], x2.y, t);
  t = __builtin_fmaf(w[10], x2.z, t);
  t = __builtin_fmaf(w[11], x2.w, t);
  t = __builtin_fmaf(w[12], x3.x, t);
  t = __builtin_fmaf(w[13], x3.y, t);
  t = __builtin_fmaf(w[14], x3.z, t);
  t = __builtin_fmaf(w[15], x3.w, t);
  return t;
}
template <int U, int B>
__device__ __forceinline__ void fma_chunk_b(const RegsQ8<U>& r, const f32x4* xs, int xstride,
                                            int c0, int M16, int plane, int lane, float (&a0)[B],
                                            float (&a1)[B]) {
#pragma unroll
  for (int u = 0; u < U; ++u) {
    const int idx = c0 + u * KH_WAVE + lane;
    if (idx < M16) {
      float w0[16], w1[16];
      cvt16_i8(r.q0[u], w0);
      cvt16_i8(r.q1[u], w1);
#pragma unroll
      for (int b = 0; b < B; ++b) {
        const f32x4* xb = xs + (size_t)b * xstride;
        const f32x4 x0 = xb[idx], x1 = xb[plane + idx], x2 = xb[2 * plane + idx],
                    x3 = xb[3 * plane + idx];
        a0[b] = __builtin_fmaf(r.g0[u], dot16(w0, x0, x1, x2, x3), a0[b]);
        a1[b] = __builtin_fmaf(r.g1[u], dot16(w1, x0, x1, x2, x3), a1[b]);
        // keep the next token's LDS reads below this token's FMAs (bounds the live operands)
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}
template <bool QUANT, int U, int B>
__device__ __forceinline__ void gemv_fma_b(const Gemv<QUANT, U>& g,
                                           const typename Gemv<QUANT, U>::Regs& r,
                                           const f32x4* xs, int xstride, int c0, int lim, int lane,
                                           float (&a0)[B], float (&a1)[B]) {
  if constexpr (QUANT)
    fma_chunk_b<U, B>(r, xs, xstride, c0, lim, g.Mc + 1, lane, a0, a1);
  else
    fma_chunk_b<U, B>(r, xs, xstride, c0, lim, lane, a0, a1);
}

// LDS of a B-token GEMV stage: xs[B] | red[KH_WAVES_MAX*B] | comb[2*KH_WAVES_MAX*B]
static inline size_t pf_tok_lds_bytes(bool quant, int M) {
  return quant ? kh_q8_lds_bytes(M) : (size_t)M * 4;
}
static inline size_t pf_lds_bytes(bool quant, int M, int B) {
  return (size_t)B * pf_tok_lds_bytes(quant, M) + (size_t)3 * KH_WAVES_MAX * B * sizeof(float);
}

// ---- the row-pair loop of kh_gemv.h::gemv_pairs with B accumulator sets -----------------------
//   PAIR(p) -> Rows; PRE(p) -> epilogue operands fetched behind the pair's weight loads;
//   STAGE() -> all B activation vectors into LDS (barriers inside);
//   EPI(p, s0[B], s1[B], aux) called by every lane of the owning wave.
// (A double-buffered register tile — loads of chunk i+1 issued before the FMAs of chunk i — was
// measured: no gain, and the int8 kernels ran out of VGPRs; other waves already cover the FMAs.)
template <bool QUANT, int U, int SPLIT, int B, class PairFn, class PreFn, class StageFn,
          class EpiFn>
__device__ __forceinline__ void gemv_pairs_b(const Gemv<QUANT, U>& g, const f32x4* xs, int xstride,
                                             int total, int lane, float* comb, PairFn&& PAIR,
                                             PreFn&& PRE, StageFn&& STAGE, EpiFn&& EPI) {
  static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "SPLIT must be 1, 2 or 4");
  const int PPW = kh_nwaves() / SPLIT;
  const int wave = threadIdx.x >> 6;
  const int part = wave & (SPLIT - 1);
  const int gp = (int)blockIdx.x * PPW + wave / SPLIT;
  const int np = (int)gridDim.x * PPW;
  const int step = KH_WAVE * U;
  const int Q = (((g.Mc + SPLIT - 1) / SPLIT) + 3) & ~3;  // the decode path's column quantum
  const int cb = part * Q;
  const int ce = cb + Q < g.Mc ? cb + Q : g.Mc;
  const int p0 = gp < total ? gp : 0;
  typename Gemv<QUANT, U>::Regs regs;
  typename Gemv<QUANT, U>::Rows cur = PAIR(p0);
  g.load(regs, cur, cb, ce, lane);  // first weight chunk in flight while the vectors are staged
  auto aux = PRE(p0);
  STAGE();
  const int iters = (total + np - 1) / np;  // uniform over the workgroup (barriers below)
  for (int it = 0; it < iters; ++it) {
    const int p = gp + it * np;
    const bool valid = p < total;
    float a0[B], a1[B];
#pragma unroll
    for (int b = 0; b < B; ++b) a0[b] = a1[b] = 0.f;
    if (valid) {
      for (int c0 = cb;;) {
        gemv_fma_b<QUANT, U, B>(g, regs, xs, xstride, c0, ce, lane, a0, a1);
        c0 += step;
        if (c0 >= ce) break;
        g.load(regs, cur, c0, ce, lane);
      }
    }
    const int pn = p + np;
    auto aux_next = aux;
    if (pn < total) {
      cur = PAIR(pn);
      aux_next = PRE(pn);  // ahead of the tile: the youngest loads at the latch are then weight loads
      g.load(regs, cur, cb, ce, lane);  // (kh_gemv.h::gemv_pairs: a vmcnt(0) drain otherwise)
    }
    float s0[B], s1[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
      s0[b] = wave_sum(a0[b]);
      s1[b] = wave_sum(a1[b]);
    }
    if constexpr (SPLIT == 1) {
      if (valid) EPI(p, s0, s1, aux);
    } else {
      if (lane == 0) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
          comb[(2 * wave) * B + b] = s0[b];
          comb[(2 * wave + 1) * B + b] = s1[b];
        }
      }
      __syncthreads();
      if (valid && part == 0) {
#pragma unroll
        for (int b = 0; b < B; ++b) {
          s0[b] = comb[(2 * wave) * B + b];
          s1[b] = comb[(2 * wave + 1) * B + b];
#pragma unroll
          for (int k = 1; k < SPLIT; ++k) {  // same fixed order as the decode path
            s0[b] += comb[(2 * (wave + k)) * B + b];
            s1[b] += comb[(2 * (wave + k) + 1) * B + b];
          }
        }
        EPI(p, s0, s1, aux);
      }
      __syncthreads();
    }
    aux = aux_next;
  }
}

// ---- staging of B vectors -----------------------------------------------------------------------
// NORM: the arithmetic of Stager<true, .., MAXV>::finish, token by token (same per-thread partial sums - a slot
// beyond the vector adds 0 whatever MAXV is -, same wave / LDS reduction order, g = w_norm * x into LDS, the
// scale rs[b] handed to the epilogue); the B sums of squares share ONE barrier with the LDS writes.
// x: [B][src_stride] floats.
template <bool LAYOUT_Q8, int B>
__device__ __forceinline__ void pf_stage_norm(const float* x, size_t src_stride,
                                              const float* wnorm, f32x4* xs, int xstride, int M,
                                              float eps, float* red /*[KH_WAVES_MAX*B]*/, float (&rs)[B]) {
  constexpr int MAXV = 4;
  constexpr int TB = 2;  // tokens staged per round (register footprint: TB*MAXV float4)
  static_assert(B % TB == 0, "B must be a multiple of the staging batch");
  const int M4 = M >> 2, M16 = M >> 4;
  const int wg = kh_wg();
  const int n = kh_nwaves();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const f32x4* w4 = (const f32x4*)wnorm;
  f32x4 wv[MAXV];
#pragma unroll
  for (int v = 0; v < MAXV; ++v) {
    const int i = threadIdx.x + v * wg;
    wv[v] = w4[i < M4 ? i : 0];
  }
#pragma unroll
  for (int b0 = 0; b0 < B; b0 += TB) {
    f32x4 xv[TB][MAXV];
#pragma unroll
    for (int t = 0; t < TB; ++t)
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int i = threadIdx.x + v * wg;
        xv[t][v] = ((const f32x4*)(x + (size_t)(b0 + t) * src_stride))[i < M4 ? i : 0];
      }
#pragma unroll
    for (int t = 0; t < TB; ++t) {
      float s = 0.f;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const float q = fma4(xv[t][v], xv[t][v], 0.f);
        s += (threadIdx.x + v * wg < M4) ? q : 0.f;
      }
      s = wave_sum(s);
      if (lane == 0) red[wave * B + b0 + t] = s;
      f32x4* xb = xs + (size_t)(b0 + t) * xstride;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int i = threadIdx.x + v * wg;
        if (i < M4) {
          f32x4 q = xv[t][v];
          q.x = wv[v].x * q.x;
          q.y = wv[v].y * q.y;
          q.z = wv[v].z * q.z;
          q.w = wv[v].w * q.w;
          xb[LAYOUT_Q8 ? q8_slot(i, M16) : i] = q;
        }
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < B; ++b) {
    float r = 0.f;
#pragma unroll
    for (int w = 0; w < KH_WAVES_MAX; ++w) r += w < n ? red[(w < n ? w : 0) * B + b] : 0.f;
    rs[b] = 1.0f / sqrtf(r / (float)M + eps);
  }
}
// plain copy (inputs of wo / w2): no arithmetic, any length
template <bool LAYOUT_Q8, int B>
__device__ __forceinline__ void pf_stage_copy(const float* x, size_t src_stride, f32x4* xs,
                                              int xstride, int M) {
  const int M4 = M >> 2, M16 = M >> 4;
  const int wg = kh_wg();
  for (int i = threadIdx.x; i < M4; i += wg) {
    f32x4 v[B];
#pragma unroll
    for (int b = 0; b < B; ++b) v[b] = ((const f32x4*)(x + (size_t)b * src_stride))[i];
#pragma unroll
    for (int b = 0; b < B; ++b) xs[(size_t)b * xstride + (LAYOUT_Q8 ? q8_slot(i, M16) : i)] = v[b];
  }
  __syncthreads();
}
template <bool QUANT>
__device__ __forceinline__ int pf_xstride(int M) {  // f32x4 slots per staged token
  return QUANT ? 4 * ((M >> 4) + 1) : (M >> 2);
}

// ---- kernels ------------------------------------------------------------------------------------
struct KhPfTokens {
  int32_t t[KH_PF_BMAX];
};
// embedding rows of the B tokens -> X[B][dim]   (emb_kernel.cu / model.cpp:245-263 fill_input)
static __global__ __launch_bounds__(KH_WG) void k_pf_embed(KhPfTokens tok, const float* __restrict__ emb,
                                                    float* __restrict__ X, int dim) {
  const int b = blockIdx.x;
  const f32x4* src = (const f32x4*)(emb + (size_t)tok.t[b] * dim);
  f32x4* dst = (f32x4*)(X + (size_t)b * dim);
  for (int i = threadIdx.x; i < (dim >> 2); i += KH_WG) dst[i] = src[i];
}

struct KhPfQkvArgs {
  const float* X;         // [B][dim] residual streams
  const float* att_norm;
  KhLin wq, wk, wv;
  float* Q;               // [B][dim]
  float* kcache_layer;
  float* vcache_layer;
  const float* sin_cache;
  const float* cos_cache;
  int dim, kv_dim, head_size, rope_mode, gshift;
  int pos0, nvalid;       // token b sits at position pos0 + b; tokens >= nvalid are padding
  float eps;
};
template <bool QUANT, int SPLIT, int B>
__global__ __launch_bounds__(KH_WG_MAX) void k_pf_qkv(const KhPfQkvArgs a) {
  constexpr int U = KH_PF_U(QUANT);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const void *wq_w = a.wq.w, *wk_w = a.wk.w, *wv_w = a.wv.w;
  const float *wq_s = a.wq.scales, *wk_s = a.wk.scales, *wv_s = a.wv.scales;
  const float *wq_b = a.wq.bias, *wk_b = a.wk.bias, *wv_b = a.wv.bias;
  float* const Qo = a.Q;
  float* const kc = a.kcache_layer;
  float* const vc = a.vcache_layer;
  const float* const sin_cache = a.sin_cache;
  const float* const cos_cache = a.cos_cache;
  const float* const X = a.X;
  const float* const att_norm = a.att_norm;
  const int dim = a.dim, kv_dim = a.kv_dim, rope_mode = a.rope_mode;
  const int pos0 = a.pos0, nvalid = a.nvalid;
  const float eps = a.eps;
  f32x4* xs = (f32x4*)smem_raw;
  const int xstride = pf_xstride<QUANT>(dim);
  float* red = (float*)(xs + (size_t)B * xstride);
  float* comb = red + KH_WAVES_MAX * B;
  const int lane = threadIdx.x & 63;
  const int hs = a.head_size, half = hs >> 1;
  const int npq = dim >> 1, npk = kv_dim >> 1;
  const int total = npq + 2 * npk;
  const Gemv<QUANT, U> g(dim, a.gshift);
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
      const int head = pp / half, j = pp - head * half;
      r0 = head * hs + j;
      r1 = r0 + half;
      cidx = 2 * j;
    } else {
      r0 = 2 * pp;
      r1 = r0 + 1;
      cidx = r0 % hs;
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
    float fci[B], fcr[B], b0, b1;
  };
  auto pre = [&](int p) __attribute__((always_inline)) {
    int which, r0, r1, cidx;
    decode(p, which, r0, r1, cidx);
    const float* bias = sel3(which, wq_b, wk_b, wv_b);
    Aux x;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      const int pos = pos0 + (b < nvalid ? b : nvalid - 1);
      x.fci[b] = sin_cache[(size_t)pos * hs + cidx];
      x.fcr[b] = cos_cache[(size_t)pos * hs + cidx];
    }
    x.b0 = bias ? bias[r0] : 0.f;
    x.b1 = bias ? bias[r1] : 0.f;
    return x;
  };
  float rs[B];  // RMS scale per token: set by the staging, applied in the epilogue (as k_qkv)
  auto epi = [&](int p, const float (&s0)[B], const float (&s1)[B], const Aux& x)
      __attribute__((always_inline)) {
    if (lane != 0) return;
    int which, r0, r1, cidx;
    decode(p, which, r0, r1, cidx);
#pragma unroll
    for (int b = 0; b < B; ++b) {
      if (b >= nvalid) break;
      float v0 = rs[b] * s0[b] + x.b0, v1 = rs[b] * s1[b] + x.b1;
      const size_t row = (size_t)(pos0 + b) * kv_dim;
      float* dst = sel3(which, Qo + (size_t)b * dim, kc + row, vc + row);
      if (which < 2) {
        const float t0 = v0, t1 = v1;
        v0 = t0 * x.fcr[b] - t1 * x.fci[b];
        v1 = t0 * x.fci[b] + t1 * x.fcr[b];
      }
      dst[r0] = v0;
      dst[r1] = v1;
    }
  };
  gemv_pairs_b<QUANT, U, SPLIT, B>(
      g, xs, xstride, total, lane, comb, pair, pre,
      [&]() __attribute__((always_inline)) {
        pf_stage_norm<QUANT, B>(X, (size_t)dim, att_norm, xs, xstride, dim, eps, red, rs);
      },
      epi);
}

struct KhPfFfn13Args {
  const float* X;   // [B][dim]
  const float* ffn_norm;
  KhLin w1, w3;
  float* H;         // [B][hidden]
  int dim, hidden, gshift, nvalid;
  float eps;
};
template <bool QUANT, int B>
__global__ __launch_bounds__(KH_WG_MAX) void k_pf_ffn13(const KhPfFfn13Args a) {
  constexpr int U = KH_PF_U(QUANT);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const void *w1 = a.w1.w, *w3 = a.w3.w;
  const float *s1p = a.w1.scales, *s3p = a.w3.scales;
  const float* const X = a.X;
  const float* const ffn_norm = a.ffn_norm;
  float* const H = a.H;
  const int dim = a.dim, hidden = a.hidden, nvalid = a.nvalid;
  const float eps = a.eps;
  f32x4* xs = (f32x4*)smem_raw;
  const int xstride = pf_xstride<QUANT>(dim);
  float* red = (float*)(xs + (size_t)B * xstride);
  const int lane = threadIdx.x & 63;
  const Gemv<QUANT, U> g(dim, a.gshift);
  auto pair = [&](int r) __attribute__((always_inline)) { return g.rows(w1, r, w3, r, s1p, s3p, dim); };
  float rs[B];  // RMS scale per token: set by the staging, applied in the epilogue (as k_ffn13)
  auto epi = [&](int r, const float (&s0)[B], const float (&s1)[B], const NoAux&)
      __attribute__((always_inline)) {
    if (lane != 0) return;
#pragma unroll
    for (int b = 0; b < B; ++b)
      if (b < nvalid) H[(size_t)b * hidden + r] = swiglu1(rs[b] * s0[b], rs[b] * s1[b]);
  };
  gemv_pairs_b<QUANT, U, 1, B>(
      g, xs, xstride, hidden, lane, nullptr, pair,
      [](int) __attribute__((always_inline)) { return NoAux{}; },
      [&]() __attribute__((always_inline)) {
        pf_stage_norm<QUANT, B>(X, (size_t)dim, ffn_norm, xs, xstride, dim, eps, red, rs);
      },
      epi);
}

struct KhPfGemvResArgs {
  const float* V;  // [B][M] input vectors
  KhLin w;         // [K, M]
  float* X;        // [B][K] residual streams, updated in place
  int M, K, gshift, nvalid;
};
template <bool QUANT, int SPLIT, int B>
__global__ __launch_bounds__(KH_WG_MAX) void k_pf_gemv_res(const KhPfGemvResArgs a) {
  constexpr int U = KH_PF_U(QUANT);
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const void* const w = a.w.w;
  const float* const scales = a.w.scales;
  const float* const V = a.V;
  float* const X = a.X;
  const int M = a.M, K = a.K, nvalid = a.nvalid;
  f32x4* xs = (f32x4*)smem_raw;
  const int xstride = pf_xstride<QUANT>(M);
  float* red = (float*)(xs + (size_t)B * xstride);
  float* comb = red + KH_WAVES_MAX * B;
  const int lane = threadIdx.x & 63;
  const Gemv<QUANT, U> g(M, a.gshift);
  auto pair = [&](int p) __attribute__((always_inline)) { return g.rows(w, 2 * p, w, 2 * p + 1, scales, scales, M); };
  struct Aux {
    float x0[B], x1[B];
  };
  auto pre = [&](int p) __attribute__((always_inline)) {
    Aux r;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      r.x0[b] = X[(size_t)b * K + 2 * p];
      r.x1[b] = X[(size_t)b * K + 2 * p + 1];
    }
    return r;
  };
  auto epi = [&](int p, const float (&s0)[B], const float (&s1)[B], const Aux& r)
      __attribute__((always_inline)) {
    if (lane != 0) return;
#pragma unroll
    for (int b = 0; b < B; ++b) {
      if (b < nvalid) {
        X[(size_t)b * K + 2 * p] = r.x0[b] + s0[b];
        X[(size_t)b * K + 2 * p + 1] = r.x1[b] + s1[b];
      }
    }
  };
  gemv_pairs_b<QUANT, U, SPLIT, B>(
      g, xs, xstride, K >> 1, lane, comb, pair, pre,
      [&]() __attribute__((always_inline)) { pf_stage_copy<QUANT, B>(V, (size_t)M, xs, xstride, M); },
      epi);
}
