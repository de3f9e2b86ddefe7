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
// All GEMV kernels share kh_gemv.h: a wave streams one ROW PAIR, and the pair is chosen so the
// epilogue has both operands in registers (RoPE partner rows, (w1,w3) rows, adjacent rows).
#pragma once
#include "kh_attn.h"
#include "kh_common.h"
#include "kh_gemv.h"

// weights of one projection, fp32 or int8+scales
struct KhLin {
  const void* w;        // fp32 [K,M] or int8 [K,M]
  const float* scales;  // int8 only: [K*M/group]
  const float* bias;    // Qwen2 q/k/v only
};

template <bool QUANT, int U>
__device__ __forceinline__ void row_pair_dot(const KhLin& L, int r0, int r1, int M, int gshift,
                                             const f32x4* xs, int lane, float& s0, float& s1) {
  if (QUANT) {
    const int8_t* w = (const int8_t*)L.w;
    const int gpr = M >> gshift;
    dot2_q8<U>((const i32x4*)(w + (size_t)r0 * M), (const i32x4*)(w + (size_t)r1 * M),
               L.scales + (size_t)r0 * gpr, L.scales + (size_t)r1 * gpr, gshift, xs, M >> 4,
               lane, s0, s1);
  } else {
    const float* w = (const float*)L.w;
    dot2_f32<U>((const f32x4*)(w + (size_t)r0 * M), (const f32x4*)(w + (size_t)r1 * M), xs,
                M >> 2, lane, s0, s1);
  }
}

template <bool QUANT>
__device__ __forceinline__ float* lds_red_ptr(f32x4* xs, int M) {
  return (float*)(xs + (QUANT ? 4 * ((M >> 4) + 1) : (M >> 2)));
}
static inline size_t fused_lds_bytes(bool quant, int M) {
  return (quant ? kh_q8_lds_bytes(M) : (size_t)M * 4) + 16;
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

template <bool QUANT, int U>
__global__ __launch_bounds__(KH_WG) void k_qkv(const KhQkvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<QUANT>(xs, a.dim);
  stage_vec<true, QUANT>(a.x, a.att_norm, xs, a.dim, a.eps, red);
  const int pos = *a.d_pos;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * KH_WAVES_PER_WG + (threadIdx.x >> 6);
  const int nw = gridDim.x * KH_WAVES_PER_WG;
  const int hs = a.head_size, half = hs >> 1;
  const int npq = a.dim >> 1, npk = a.kv_dim >> 1;
  const int total = npq + 2 * npk;
  const float* srow = a.sin_cache + (size_t)pos * hs;
  const float* crow = a.cos_cache + (size_t)pos * hs;
  for (int p = gw; p < total; p += nw) {
    int which, pp;
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
    int r0, r1, cidx = 0;
    if (which < 2 && a.rope_mode == KH_ROPE_HALF) {
      // cpu/rope_kernel.cpp:18-42: pair (head*hs + j, + hs/2), cache column 2j
      const int head = pp / half, j = pp - head * half;
      r0 = head * hs + j;
      r1 = r0 + half;
      cidx = 2 * j;
    } else {
      // cpu/rope_kernel.cpp:98-121: pair (2i, 2i+1), cache column (2i % hs); v: plain pair
      r0 = 2 * pp;
      r1 = r0 + 1;
      cidx = r0 % hs;
    }
    const KhLin& L = which == 0 ? a.wq : (which == 1 ? a.wk : a.wv);
    float s0, s1;
    row_pair_dot<QUANT, U>(L, r0, r1, a.dim, a.gshift, xs, lane, s0, s1);
    if (lane == 0) {
      if (L.bias) {  // matmul.cpp:74-77: bias added after the matmul, before RoPE
        s0 = s0 + L.bias[r0];
        s1 = s1 + L.bias[r1];
      }
      float* dst = which == 0 ? a.q_out
                              : (which == 1 ? a.kcache_layer : a.vcache_layer) +
                                    (size_t)pos * a.kv_dim;
      if (which < 2) {
        const float fci = srow[cidx], fcr = crow[cidx];
        const float v0 = s0, v1 = s1;
        s0 = v0 * fcr - v1 * fci;
        s1 = v0 * fci + v1 * fcr;
      }
      dst[r0] = s0;
      dst[r1] = s1;
    }
  }
}

// ---------------------------------------------------------------------------------------------
struct KhAttnArgs {
  const float* q;          // [dim]
  const float* kcache_layer;
  const float* vcache_layer;
  float* out;              // [dim]
  const int32_t* d_pos;
  int kv_dim, kv_mul, head_size;
};
__global__ __launch_bounds__(KH_WG) void k_attn(const KhAttnArgs a) {
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
template <bool QUANT, int U>
__global__ __launch_bounds__(KH_WG) void k_gemv_res(const KhGemvResArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<QUANT>(xs, a.M);
  stage_vec<false, QUANT>(a.vec, nullptr, xs, a.M, 0.f, red);
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * KH_WAVES_PER_WG + (threadIdx.x >> 6);
  const int nw = gridDim.x * KH_WAVES_PER_WG;
  const int npairs = a.K >> 1;  // K even (checked at model build)
  for (int p = gw; p < npairs; p += nw) {
    const int r0 = 2 * p, r1 = r0 + 1;
    float s0, s1;
    row_pair_dot<QUANT, U>(a.w, r0, r1, a.M, a.gshift, xs, lane, s0, s1);
    if (lane == 0) {
      a.x[r0] = a.x[r0] + s0;
      a.x[r1] = a.x[r1] + s1;
    }
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
template <bool QUANT, int U>
__global__ __launch_bounds__(KH_WG) void k_ffn13(const KhFfn13Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<QUANT>(xs, a.dim);
  stage_vec<true, QUANT>(a.x, a.ffn_norm, xs, a.dim, a.eps, red);
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * KH_WAVES_PER_WG + (threadIdx.x >> 6);
  const int nw = gridDim.x * KH_WAVES_PER_WG;
  for (int r = gw; r < a.hidden; r += nw) {
    float s0, s1;
    if (QUANT) {
      const int gpr = a.dim >> a.gshift;
      dot2_q8<U>((const i32x4*)((const int8_t*)a.w1.w + (size_t)r * a.dim),
                 (const i32x4*)((const int8_t*)a.w3.w + (size_t)r * a.dim),
                 a.w1.scales + (size_t)r * gpr, a.w3.scales + (size_t)r * gpr, a.gshift, xs,
                 a.dim >> 4, lane, s0, s1);
    } else {
      dot2_f32<U>((const f32x4*)((const float*)a.w1.w + (size_t)r * a.dim),
                  (const f32x4*)((const float*)a.w3.w + (size_t)r * a.dim), xs, a.dim >> 2,
                  lane, s0, s1);
    }
    if (lane == 0) a.h[r] = swiglu1(s0, s1);
  }
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
template <bool QUANT, int U>
__global__ __launch_bounds__(KH_WG) void k_cls(const KhClsArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<QUANT>(xs, a.dim);
  stage_vec<true, QUANT>(a.x, a.final_norm, xs, a.dim, a.eps, red);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int gw = blockIdx.x * KH_WAVES_PER_WG + wave;
  const int nw = gridDim.x * KH_WAVES_PER_WG;
  const int npairs = (a.vocab + 1) >> 1;
  float bv = -INFINITY;
  int bi = 0x7fffffff;
  for (int p = gw; p < npairs; p += nw) {
    const int r0 = 2 * p;
    const int r1 = r0 + 1 < a.vocab ? r0 + 1 : r0;
    float s0, s1;
    row_pair_dot<QUANT, U>(a.wcls, r0, r1, a.dim, a.gshift, xs, lane, s0, s1);
    if (lane == 0) {
      a.logits[r0] = s0;
      amax_merge(bv, bi, s0, r0);
      if (r1 != r0) {
        a.logits[r1] = s1;
        amax_merge(bv, bi, s1, r1);
      }
    }
  }
  // stage-1 argmax: one partial per workgroup (ties -> lowest index)
  int* redi = (int*)(red + KH_WAVES_PER_WG);
  __syncthreads();
  if (lane == 0) {
    red[wave] = bv;
    redi[wave] = bi;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float v = red[0];
    int i = redi[0];
#pragma unroll
    for (int w = 1; w < KH_WAVES_PER_WG; ++w) amax_merge(v, i, red[w], redi[w]);
    a.part_val[blockIdx.x] = v;
    a.part_idx[blockIdx.x] = i;
  }
}
static inline size_t cls_lds_bytes(bool quant, int M) {
  return fused_lds_bytes(quant, M) + 2 * KH_WAVES_PER_WG * sizeof(float);
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
__global__ __launch_bounds__(KH_WG) void k_sample(const KhSampleArgs a) {
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
__global__ __launch_bounds__(KH_WG) void k_set_state(int token, int pos, int32_t* d_token,
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
