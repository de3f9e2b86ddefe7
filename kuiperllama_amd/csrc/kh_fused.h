// kh_fused.h — the fused decode-step kernels (5 per layer + classifier + sampler).
//
// The reference issues 14 launches per layer (17 for Qwen2) from host code that also reads
// `pos` on the CPU.  Here one token = 5L+2 launches whose every
// offset derives from the DEVICE scalar *d_pos, so the whole step is a replayable hipGraph:
//
//   k_qkv    rmsnorm(x) -> LDS ; [wq|wk|wv] row pairs ; +bias ; RoPE ; q -> q_buf,
//            k,v -> cache row `pos`                      (llama3.cpp:600-640, matmul.cpp:74-77)
//   k_attn   decode attention, one WG per head            (llama3.cpp:652-668)
//   k_wo     wo . att + residual add into x               (llama3.cpp:670-686)
//   k_ffn13  rmsnorm(x) -> LDS ; (w1 row, w3 row) pairs ; SwiGLU -> h   (llama3.cpp:688-708)
//   k_w2     w2 . h + residual add into x                 (llama3.cpp:710-719)
//   k_cls    rmsnorm(x) -> LDS ; classifier rows -> logits + per-WG argmax partials (:722-731)
//   k_sample merge partials -> next token (or forced prompt token), append to words,
//            gather its embedding row into x, ++*d_pos    (llama3.cpp:733-745, main.cpp:20-41)
//
// All GEMV kernels share kh_gemv.h::gemv_pairs: a wave streams one ROW PAIR, and the pair is
// chosen so the epilogue has both operands in registers (RoPE partner rows, (w1,w3) rows,
// adjacent rows).
#pragma once
#include "kh_attn.h"
#include "kh_common.h"
#include "kh_gemv.h"

// Refill policy of gemv_pairs per kernel (same-box A/B of all four combinations on four models,
// profiles/r3_roll_vs_bulk_ab.txt): the slot-by-slot ROLLING refill pays only in the int8 QKV kernel
// (10.9 vs 11.7 us; three short items per wave, dequant-heavy); everywhere else requesting the next
// tile in one burst after the FMAs is as fast or faster (fp32 ffn13 21.4 vs 21.9 us, Llama-2-7B fp32
// w2 30.4 vs 32.4) - longer bursts per DRAM row.


template <bool QUANT>
__device__ __forceinline__ float* lds_red_ptr(f32x4* xs, int M) {
  return (float*)(xs + (QUANT ? 4 * ((M >> 4) + 1) : (M >> 2)));
}
// xs | red[KH_WAVES_MAX] | comb[2*KH_WAVES_MAX] (split-row partial sums)
static inline size_t fused_lds_bytes(bool quant, int M) {
  return (quant ? kh_q8_lds_bytes(M) : (size_t)M * 4) + 3 * KH_WAVES_MAX * sizeof(float);
}

// Three-way select on VALUES.  Written as `w == 0 ? a : ...` directly on named variables, the
// conditional operator yields an lvalue, clang selects between the variables' ADDRESSES, and the
// variables (kernel-argument pointers) are forced into scratch memory.
template <class T>
__device__ __forceinline__ T sel3(int w, T a, T b, T c) {
  return w == 0 ? a : (w == 1 ? b : c);
}

// ---------------------------------------------------------------------------------------------
struct KhQkvArgs {
  const float* x;         // residual stream [dim]
  const float* att_norm;  // [dim]
  KhLin wq, wk, wv;
  float* q_out;           // [dim]
  float* kcache_layer;    // cache + layer*cache_len*kv_dim
  float* vcache_layer;
  const int32_t* d_pos;
  const float* sin_cache;  // [cache_len, hs]
  const float* cos_cache;
  int dim, kv_dim, head_size, rope_mode, gshift;
  float eps;
};

// __launch_bounds__(512, 4): at least 4 waves per SIMD, i.e. at most 128 VGPRs.  The long-row shapes run
// two 512-thread workgroups per CU (16 waves); the fp32 U = 8 kernels sit at 120-130 registers and a
// build that crossed 128 lost the second workgroup (w2 fp32 11.8 -> 12.5 us, profiles/r3_aux_first_ab.txt).
template <bool QUANT, int U, int MAXV, int SPLIT>
__global__ __launch_bounds__(KH_WG_MAX, 4) void k_qkv(const KhQkvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  KH_STAMP_INIT();
  // Every kernel argument used inside the lambdas is first copied into a scalar local: a
  // lambda that captures the argument STRUCT by reference keeps the whole struct addressable
  // and the compiler then parks it in scratch memory (seen as 168 B/lane of scratch traffic).
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
  float* red = lds_red_ptr<QUANT>(xs, dim);
  const int lane = threadIdx.x & 63;
  const int hs = a.head_size, half = hs >> 1;
  // head sizes are powers of two in every BASELINE config: shifts instead of the ~25-instruction
  // integer division the work-item decode would otherwise run three times per row pair
  const bool half_pow2 = (half & (half - 1)) == 0;
  const int half_sh = __builtin_ctz((unsigned)half | 0x40000000u);
  const int npq = dim >> 1, npk = kv_dim >> 1;
  const int total = npq + 2 * npk;
  const Gemv<QUANT, U> g(dim, a.gshift);
  Stager<true, QUANT, MAXV> st(a.x, a.att_norm, dim);
  const int pos = *a.d_pos;

  // work item p -> (which projection, row pair, sin/cos column)
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
      // cpu/rope_kernel.cpp:18-42: pair (head*hs + j, + hs/2), cache column 2j
      const int head = half_pow2 ? pp >> half_sh : pp / half, j = pp - head * half;
      r0 = head * hs + j;
      r1 = r0 + half;
      cidx = 2 * j;
    } else {
      // cpu/rope_kernel.cpp:98-121: pair (2i, 2i+1), cache column (2i % hs); v: plain pair
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
  float rs = 1.f;  // RMS scale of x: set by the staging, applied in the epilogue
  auto pre = [&](int p) __attribute__((always_inline)) {
    int which, r0, r1, cidx;
    decode(p, which, r0, r1, cidx);
    const float* bias = sel3(which, wq_b, wk_b, wv_b);
    Aux x;
    x.fci = sin_cache[(size_t)pos * hs + cidx];
    x.fcr = cos_cache[(size_t)pos * hs + cidx];
    x.b0 = bias ? bias[r0] : 0.f;
    x.b1 = bias ? bias[r1] : 0.f;
    return x;
  };
  auto epi = [&](int p, float s0, float s1, const Aux& x) __attribute__((always_inline)) {
    if (lane != 0) return;
    int which, r0, r1, cidx;
    decode(p, which, r0, r1, cidx);
    // the RMS scale of the staged vector (kh_gemv.h::stage_vec), then matmul.cpp:74-77: bias added after
    // the matmul, before RoPE (x + 0.f is exact)
    s0 = rs * s0 + x.b0;
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
  gemv_pairs<SPLIT, /*ROLL=*/QUANT>(g, xs, total, lane, red + KH_WAVES_MAX, pair, pre,
                              [&]() __attribute__((always_inline)) { st.issue(); },
                              [&]() __attribute__((always_inline)) { rs = st.finish(xs, eps, red); }, epi);
  KH_STAMP_FLUSH();
}

// ---------------------------------------------------------------------------------------------
// head_size <= 32 (tiny test models): the generic LDS-score core of the op-level kernel
static __global__ __launch_bounds__(KH_WG_MAX) void k_attn_generic(const KhAttnArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int pos = *a.d_pos;
  const int h = blockIdx.x;
  const size_t head_off = (size_t)(h / a.kv_mul) * a.head_size;
  attn_head_decode(a.q + (size_t)h * a.head_size, a.kcache_layer + head_off,
                   a.vcache_layer + head_off, a.kv_dim, a.head_size, pos,
                   a.out + (size_t)h * a.head_size, nullptr, (float*)smem_raw);
}

// ---------------------------------------------------------------------------------------------
// y = W . v ; x += y      (wo and w2 with their residual adds, llama3.cpp:670-686, 710-719)
struct KhGemvResArgs {
  const float* vec;  // [M]
  KhLin w;           // [K, M]
  float* x;          // [K] residual stream, updated in place
  int M, K, gshift;
};
// STAGER(a, M): the staging of the input vector - Stager<false, ...> over a.vec, or CombStager over the
// split partials of a deferring attention launch (k_wo_comb below).
template <bool QUANT, int U, int SPLIT, class St>
__device__ __forceinline__ void gemv_res_body(const KhGemvResArgs& a, St& st, f32x4* xs, float* red) {
  // scalar locals for everything the lambdas touch (see qkv_body)
  const void* const w = a.w.w;
  const float* const scales = a.w.scales;
  float* const x = a.x;
  const int M = a.M;
  const int lane = threadIdx.x & 63;
  const Gemv<QUANT, U> g(M, a.gshift);
  auto pair = [&](int p) __attribute__((always_inline)) { return g.rows(w, 2 * p, w, 2 * p + 1, scales, scales, M); };
  struct Aux {
    float x0, x1;
  };
  auto pre = [&](int p) __attribute__((always_inline)) { return Aux{x[2 * p], x[2 * p + 1]}; };  // residual, fetched early
  auto epi = [&](int p, float s0, float s1, const Aux& r) __attribute__((always_inline)) {
    if (lane != 0) return;
    x[2 * p] = r.x0 + s0;
    x[2 * p + 1] = r.x1 + s1;
  };
  gemv_pairs<SPLIT, /*ROLL=*/false>(
      g, xs, a.K >> 1 /* K even, checked at model build */, lane, red + KH_WAVES_MAX, pair, pre,
      [&]() __attribute__((always_inline)) { st.issue(); },
      [&]() __attribute__((always_inline)) { (void)st.finish(xs, 0.f, red); }, epi);
}
template <bool QUANT, int U, int MAXV, int SPLIT>
__global__ __launch_bounds__(KH_WG_MAX, 4) void k_gemv_res(const KhGemvResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  KH_STAMP_INIT();
  f32x4* xs = (f32x4*)smem_raw;
  Stager<false, QUANT, MAXV> st(a.vec, nullptr, a.M);
  gemv_res_body<QUANT, U, SPLIT>(a, st, xs, lds_red_ptr<QUANT>(xs, a.M));
  KH_STAMP_FLUSH();
}

// ---------------------------------------------------------------------------------------------
// Deferred merge of the decode-attention time splits (kh_attn.h, defer mode), done by the kernel that
// consumes the attention output anyway: while wo stages its input vector it reads, per head, the
// nact <= 16 partials (M_s, L_s, o_s[hs]) the split workgroups left in the workspace and forms
//     att[h][e] = (sum_s o_s[e] f_s) / (sum_s L_s f_s) ,   f_s = exp(M_s - max_k M_k)
// with both sums as fmaf chains in ascending s - the arithmetic of the in-launch merger
// (kh_attn.h::attn_merge_splits), bit for bit, so the two ways of merging are interchangeable; one
// active split gives o / L, the value the attention kernel itself writes when it is not deferring.
// The factors f_s are computed once per workgroup (16 lanes per head: DPP max) and parked in LDS beside
// the L_s; the o loads do not depend on them, so the (M, L) loads, the first batch of o loads and the
// first weight tile of gemv_pairs all leave before the first wait: one or two L2 round trips under the
// first HBM round trip, against the ~9 us of publish -> ticket -> acquire -> merge the in-launch form
// costs the attention kernel (r3 profile, pos 4095).
// This is a kernel of its own, not a branch inside k_gemv_res: with both stagings behind a uniform
// branch the compiler's s_waitcnt bookkeeping merged their counter states and the weight loop of the
// PLAIN path lost its progressive vmcnt ladder (seen in the ISA).  Which one a step launches is decided
// on the host from the position range of the captured graph (kh_model_step.hip::step_variant).
struct KhCombArgs {
  const float* ml;        // [heads, nsw, 2]  (M, L) per (head, split slot)
  const float* o;         // [heads, nsw, hs] unnormalised outputs
  const int32_t* d_pos;
  int ns;                 // time splits per head carried by the attention grid (<= KH_ATTN_MAX_NS)
  int ts_shift;           // log2 of the split quantum (KhAttnArgs::ts_shift)
  int nsw;                // split slots per head in the workspace
  int heads, hs;
};
// MAXV float4 of the vector per thread (as Stager); SB splits per batch of o loads.
template <bool LAYOUT_Q8, int MAXV, int SB>
struct CombStager {
  static_assert(KH_ATTN_MAX_NS == 16, "factor slots are laid out 16 per head");
  f32x4 ov[SB * MAXV];
  float cm[KH_COMB_CP], cl[KH_COMB_CP];
  const float* ml;
  const f32x4* o4;
  float* fl;  // LDS: f[heads * 16] | L[heads * 16]
  int M, hs4, hsh, nact, nsw, heads;
  __device__ __forceinline__ CombStager(const KhCombArgs& c, int M_, int nact_, float* fl_)
      : ml(c.ml), o4((const f32x4*)c.o), fl(fl_), M(M_), hs4(c.hs >> 2), nact(nact_), nsw(c.nsw), heads(c.heads) {
    hsh = __builtin_ctz((unsigned)hs4);  // float4 index -> head: head sizes are powers of two (comb_supported)
  }
  __device__ __forceinline__ int head_of(int i) const { return i >> hsh; }
  __device__ __forceinline__ void load_batch(int s0) {
    const int M4 = M >> 2;
#pragma unroll
    for (int u = 0; u < SB; ++u) {
      const int s = s0 + u < nact ? s0 + u : nact - 1;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int i = threadIdx.x + v * kh_wg();
        const int ci = i < M4 ? i : 0;
        const int h = head_of(ci);
        ov[u * MAXV + v] = o4[(size_t)(h * nsw + s) * hs4 + (ci - h * hs4)];
      }
    }
  }
  __device__ __forceinline__ void issue() {
#pragma unroll
    for (int c = 0; c < KH_COMB_CP; ++c) {  // (M, L) of (head, split) = (t >> 4, t & 15)
      const int t = threadIdx.x + c * kh_wg();
      const int h = t >> 4, s = t & 15;
      const bool ok = h < heads && s < nact;
      const size_t idx = ok ? (size_t)(h * nsw + s) * 2 : 0;
      cm[c] = ml[idx];
      cl[c] = ml[idx + 1];
    }
    load_batch(0);
  }
  __device__ __forceinline__ void finish(f32x4* xs, float /*eps*/, float* /*red*/) {
    const int nf = heads * 16;
#pragma unroll
    for (int c = 0; c < KH_COMB_CP; ++c) {
      const int t = threadIdx.x + c * kh_wg();
      const int h = t >> 4, s = t & 15;
      const bool ok = h < heads && s < nact;
      const float mm = ok ? cm[c] : -INFINITY;
      const float Mx = group_max<16>(mm);  // all 64 lanes: DPP, no branch
      if (h < heads) {
        fl[t] = ok ? expf(mm - Mx) : 0.f;
        fl[nf + t] = ok ? cl[c] : 0.f;
      }
    }
    __syncthreads();
    const int M4 = M >> 2, M16 = M >> 4;
    f32x4 num[MAXV];
    float den[MAXV];
    int hb[MAXV];
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      num[v] = f32x4{0.f, 0.f, 0.f, 0.f};
      den[v] = 0.f;
      const int i = threadIdx.x + v * kh_wg();
      hb[v] = head_of(i < M4 ? i : 0) << 4;
    }
    for (int s0 = 0;;) {
#pragma unroll
      for (int u = 0; u < SB; ++u) {
        // (the per-split branch stays HERE, unlike in kh_attn.h::attn_merge_lds: without it the compiler hoists every
        // factor read of the batch, the 4-float4 instantiations cross their 128 registers and spill 20-24 bytes of
        // scratch - and the launch gained nothing, profiles/r6_attn_merge_bf_ab.txt)
        if (s0 + u < nact) {  // uniform
#pragma unroll
          for (int v = 0; v < MAXV; ++v) {
            const float f = fl[hb[v] + s0 + u];
            const float l = fl[nf + hb[v] + s0 + u];
            const f32x4 t = ov[u * MAXV + v];
            num[v].x = __builtin_fmaf(t.x, f, num[v].x);
            num[v].y = __builtin_fmaf(t.y, f, num[v].y);
            num[v].z = __builtin_fmaf(t.z, f, num[v].z);
            num[v].w = __builtin_fmaf(t.w, f, num[v].w);
            den[v] = __builtin_fmaf(l, f, den[v]);
          }
        }
      }
      s0 += SB;
      if (s0 >= nact) break;
      load_batch(s0);
    }
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int i = threadIdx.x + v * kh_wg();
      if (i < M4) {
        f32x4 r;
        r.x = num[v].x / den[v];
        r.y = num[v].y / den[v];
        r.z = num[v].z / den[v];
        r.w = num[v].w / den[v];
        xs[LAYOUT_Q8 ? q8_slot(i, M16) : i] = r;
      }
    }
    __syncthreads();
  }
};
// xs | red | comb | f[heads * 16] | L[heads * 16]
static inline size_t comb_lds_bytes(bool quant, int M, int heads) {
  return fused_lds_bytes(quant, M) + (size_t)2 * heads * KH_ATTN_MAX_NS * sizeof(float);
}
struct KhWoCombArgs {
  KhGemvResArgs g;  // g.vec unused
  KhCombArgs cb;
};
// wo behind a deferring attention launch.  MAXV in {2, 4} float4 of the dim-long vector per thread.
template <bool QUANT, int U, int MAXV, int SPLIT>
__global__ __launch_bounds__(KH_WG_MAX, 4) void k_wo_comb(const KhWoCombArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<QUANT>(xs, a.g.M);
  const int nact = attn_active_splits(*a.cb.d_pos, a.cb.ns, a.cb.ts_shift);
  float* fl = red + 3 * KH_WAVES_MAX;
  // The o loads travel beside the first weight tile where both fit the 128 registers of a
  // 4-waves-per-SIMD launch (4 splits x 2 float4 or 2 x 4 per batch).  A 64-register tile (fp32 U = 8) or
  // the int8 tile beside 4 float4 per split does not leave that room (scratch in the first build): there
  // the vector is staged completely, 16 float4 per batch, before gemv_pairs requests its first tile.
  constexpr bool OVERLAP = !(U >= 8 || (QUANT && U >= 4 && MAXV >= 4));
  if constexpr (OVERLAP) {
    CombStager<QUANT, MAXV, (MAXV >= 4 ? 2 : 4)> st(a.cb, a.g.M, nact, fl);
    gemv_res_body<QUANT, U, SPLIT>(a.g, st, xs, red);
  } else {
    CombStager<QUANT, MAXV, 16 / MAXV> st(a.cb, a.g.M, nact, fl);
    st.issue();
    st.finish(xs, 0.f, red);
    struct Staged {
      __device__ __forceinline__ void issue() {}
      __device__ __forceinline__ void finish(f32x4*, float, float*) {}
    } none;
    gemv_res_body<QUANT, U, SPLIT>(a.g, none, xs, red);
  }
}

// ---------------------------------------------------------------------------------------------
struct KhFfn13Args {
  const float* x;
  const float* ffn_norm;
  KhLin w1, w3;
  float* h;  // [hidden]
  int dim, hidden, gshift;
  float eps;
};
template <bool QUANT, int U, int MAXV>
__global__ __launch_bounds__(KH_WG_MAX, 4) void k_ffn13(const KhFfn13Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  KH_STAMP_INIT();
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<QUANT>(xs, a.dim);
  const int lane = threadIdx.x & 63;
  const Gemv<QUANT, U> g(a.dim, a.gshift);
  Stager<true, QUANT, MAXV> st(a.x, a.ffn_norm, a.dim);
  auto pair = [&](int r) __attribute__((always_inline)) { return g.rows(a.w1.w, r, a.w3.w, r, a.w1.scales, a.w3.scales, a.dim); };
  float rs = 1.f;  // RMS scale of x: set by the staging, applied in the epilogue
  auto epi = [&](int r, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (lane == 0) a.h[r] = swiglu1(rs * s0, rs * s1);
  };
  gemv_pairs<1, /*ROLL=*/false>(g, xs, a.hidden, lane, nullptr, pair, [](int) __attribute__((always_inline)) { return NoAux{}; },
                       [&]() __attribute__((always_inline)) { st.issue(); },
                       [&]() __attribute__((always_inline)) { rs = st.finish(xs, a.eps, red); }, epi);
  KH_STAMP_FLUSH();
}

// ---------------------------------------------------------------------------------------------
struct KhClsArgs {
  const float* x;
  const float* final_norm;
  KhLin wcls;
  float* logits;    // [vocab]
  float* part_val;  // [gridDim.x]
  int32_t* part_idx;
  int dim, vocab, gshift;
  float eps;
};
template <bool QUANT, int U, int MAXV>
__global__ __launch_bounds__(KH_WG_MAX, 4) void k_cls(const KhClsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  KH_STAMP_INIT();
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<QUANT>(xs, a.dim);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const Gemv<QUANT, U> g(a.dim, a.gshift);
  Stager<true, QUANT, MAXV> st(a.x, a.final_norm, a.dim);
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  auto r1_of = [&](int p) __attribute__((always_inline)) { return 2 * p + 1 < a.vocab ? 2 * p + 1 : 2 * p; };
  auto pair = [&](int p) __attribute__((always_inline)) {
    return g.rows(a.wcls.w, 2 * p, a.wcls.w, r1_of(p), a.wcls.scales, a.wcls.scales, a.dim);
  };
  float rs = 1.f;  // RMS scale of x: set by the staging, applied in the epilogue
  auto epi = [&](int p, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (lane != 0) return;
    const int r0 = 2 * p, r1 = r1_of(p);
    s0 *= rs;
    s1 *= rs;
    a.logits[r0] = s0;
    amax_merge(bv, bi, s0, r0);
    if (r1 != r0) {
      a.logits[r1] = s1;
      amax_merge(bv, bi, s1, r1);
    }
  };
  gemv_pairs<1, /*ROLL=*/false>(g, xs, (a.vocab + 1) >> 1, lane, nullptr, pair,
                          [](int) __attribute__((always_inline)) { return NoAux{}; },
                       [&]() __attribute__((always_inline)) { st.issue(); },
                       [&]() __attribute__((always_inline)) { rs = st.finish(xs, a.eps, red); }, epi);
  // stage-1 argmax: one partial per workgroup (ties -> lowest index)
  int* redi = (int*)(red + 3 * KH_WAVES_MAX);
  __syncthreads();
  if (lane == 0) {
    red[wave] = bv;
    redi[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = red[0];
    int i = redi[0];
    for (int w = 1, nw = kh_nwaves(); w < nw; ++w) amax_merge(v, i, red[w], redi[w]);
    a.part_val[blockIdx.x] = v;
    a.part_idx[blockIdx.x] = i;
  }
  KH_STAMP_FLUSH();
}
static inline size_t cls_lds_bytes(bool quant, int M) {
  return fused_lds_bytes(quant, M) + KH_WAVES_MAX * sizeof(int);
}

// ---------------------------------------------------------------------------------------------
struct KhSampleArgs {
  const float* part_val;
  const int32_t* part_idx;
  int nparts;
  const int32_t* forced;  // [n_forced] token to feed at position i, or -1 => sampled
  int n_forced;
  int32_t* words;         // [>= steps] the reference's `words` vector (main.cpp:14,36-41)
  int words_cap;
  int32_t* d_next;        // argmax result (-1 while in the prompt, like post_processing)
  int32_t* d_token;       // token fed at the NEXT position
  int32_t* d_pos;
  const float* tok_emb;   // [vocab, dim]
  float* x;               // residual stream: receives the next token's embedding row
  int dim, vocab;
  int advance;            // 1: generate loop (feed next token, ++pos); 0: predict() only
};
static __global__ __launch_bounds__(KH_WG) void k_sample(const KhSampleArgs a) {
  __shared__ float sv[KH_WAVES_PER_WG];
  __shared__ int si[KH_WAVES_PER_WG];
  __shared__ int s_next;
  float v = -INFINITY;
  int idx = 0x7fffffff;
  for (int i = threadIdx.x; i < a.nparts; i += KH_WG) amax_merge(v, idx, a.part_val[i], a.part_idx[i]);
  wave_amax(v, idx);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sv[wave] = v;
    si[wave] = idx;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    v = sv[0];
    idx = si[0];
#pragma unroll
    for (int w = 1; w < KH_WAVES_PER_WG; ++w) amax_merge(v, idx, sv[w], si[w]);
    const int pos = *a.d_pos;
    int feed = idx;
    int reported = idx;
    if (a.forced && pos + 1 < a.n_forced && a.forced[pos + 1] >= 0) {
      feed = a.forced[pos + 1];  // prompt phase: next = tokens[pos+1] (main.cpp:36-38)
      reported = -1;             // post_processing returns -1 while is_prompt (llama3.cpp:738)
    }
    *a.d_next = reported;
    if (a.advance) {
      if (a.words && pos < a.words_cap) a.words[pos] = feed;
      *a.d_token = feed;
      *a.d_pos = pos + 1;
    }
    s_next = a.advance ? feed : -1;
  }
  __syncthreads();
  const int nxt = s_next;
  if (nxt >= 0 && nxt < a.vocab) {
    const f32x4* src = (const f32x4*)(a.tok_emb + (size_t)nxt * a.dim);
    f32x4* dst = (f32x4*)a.x;
    for (int i = threadIdx.x; i < (a.dim >> 2); i += KH_WG) dst[i] = src[i];
  }
}

// set (token, pos) from the host and gather the embedding row: start of generate / predict
static __global__ __launch_bounds__(KH_WG) void k_set_state(int token, int pos, int32_t* d_token,
                                                     int32_t* d_pos, const float* tok_emb,
                                                     float* x, int dim) {
  if (threadIdx.x == 0) {
    *d_token = token;
    *d_pos = pos;
  }
  const f32x4* src = (const f32x4*)(tok_emb + (size_t)token * dim);
  f32x4* dst = (f32x4*)x;
  for (int i = threadIdx.x; i < (dim >> 2); i += KH_WG) dst[i] = src[i];
}
