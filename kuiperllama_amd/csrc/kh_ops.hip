// kh_ops.hip — operator-level kernels + C-ABI, one per reference kernel
// (kuiper/source/op/kernels/kernels_interface.h:6-68).  gfx950 only.
//
// These are the literal drop-ins for kernel::get_*_kernel(kDeviceHIP): same argument
// meaning as the reference's CUDA kernels, raw device pointers instead of tensor::Tensor.
// The fused decode path (kh_model_step.hip) reuses the same device cores (kh_gemv.h, kh_attn.h).
#include "kh_attn.h"
#include <deque>
#include <mutex>
#include <vector>

#include "kh_common.h"
#include "kh_gemv.h"

// =============================================================================================
// add / swiglu / scale : elementwise, HBM/L2-bound, float4 body + scalar tail
// reference: cuda/add_kernel.cu:4-12 (512 thr), cuda/swiglu_kernel.cu:4-22 (pointless smem
// staging dropped)
enum { EW_ADD = 0, EW_SWIGLU = 1 };
template <int OP>
__global__ __launch_bounds__(KH_WG) void k_elementwise(const float* a, const float* b,
                                                       float* out, int n, int vec_ok) {
  // no __restrict__: out aliases an input in the model (llama3.cpp:684,708,718)
  const int tid = blockIdx.x * KH_WG + threadIdx.x;
  const int nthr = gridDim.x * KH_WG;
  int done = 0;
  if (vec_ok) {
    const int n4 = n >> 2;
    const f32x4* a4 = (const f32x4*)a;
    const f32x4* b4 = (const f32x4*)b;
    f32x4* o4 = (f32x4*)out;
    for (int i = tid; i < n4; i += nthr) {
      const f32x4 x = a4[i], y = b4[i];
      f32x4 r;
      if (OP == EW_ADD) {
        r = x + y;
      } else {
        r.x = swiglu1(x.x, y.x);
        r.y = swiglu1(x.y, y.y);
        r.z = swiglu1(x.z, y.z);
        r.w = swiglu1(x.w, y.w);
      }
      o4[i] = r;
    }
    done = n4 << 2;
  }
  for (int i = done + tid; i < n; i += nthr)
    out[i] = OP == EW_ADD ? a[i] + b[i] : swiglu1(a[i], b[i]);
}

static inline int ew_grid(int n) {
  int g = (n / 4 + KH_WG - 1) / KH_WG;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;
  return g;
}

extern "C" int kh_add_f32(const float* in1, const float* in2, float* out, int32_t n,
                          void* stream) {
  if (!in1 || !in2 || !out || n <= 0) return KH_ERR_INVALID_ARG;
  const int vec = kh_aligned16(in1) && kh_aligned16(in2) && kh_aligned16(out);
  hipLaunchKernelGGL(k_elementwise<EW_ADD>, dim3(ew_grid(n)), dim3(KH_WG), 0,
                     (hipStream_t)stream, in1, in2, out, n, vec);
  return kh_launch_status();
}

extern "C" int kh_swiglu_f32(const float* a, const float* b, float* out, int32_t n,
                             void* stream) {
  if (!a || !b || !out || n <= 0) return KH_ERR_INVALID_ARG;
  const int vec = kh_aligned16(a) && kh_aligned16(b) && kh_aligned16(out);
  hipLaunchKernelGGL(k_elementwise<EW_SWIGLU>, dim3(ew_grid(n)), dim3(KH_WG), 0,
                     (hipStream_t)stream, a, b, out, n, vec);
  return kh_launch_status();
}

__global__ __launch_bounds__(KH_WG) void k_scale(float scale, float* __restrict__ x, int n) {
  for (int i = blockIdx.x * KH_WG + threadIdx.x; i < n; i += gridDim.x * KH_WG)
    x[i] = x[i] * scale;
}
extern "C" int kh_scale_f32(float scale, float* x, int32_t n, void* stream) {
  if (!x || n <= 0) return KH_ERR_INVALID_ARG;
  int g = (n + KH_WG - 1) / KH_WG;
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(k_scale, dim3(g), dim3(KH_WG), 0, (hipStream_t)stream, scale, x, n);
  return kh_launch_status();
}

// =============================================================================================
// matmul fp32 (GEMV).  reference: cuda/matmul_kernel.cu:7-54 = one 128-thread block per row,
// x re-read from global by every block.  Here: wave per row pair, x staged once per WG in LDS.
template <int U, int MAXV>
__global__ __launch_bounds__(KH_WG) void k_matmul_f32(const float* __restrict__ x,
                                                      const float* __restrict__ w,
                                                      float* __restrict__ y, int M, int K,
                                                      float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  const int M4 = M >> 2;
  float* red = (float*)(xs + M4);
  const int lane = threadIdx.x & 63;
  const Gemv<false, U> g(M, 0);
  Stager<false, false, MAXV> st(x, nullptr, M);
  auto r1_of = [&](int p) __attribute__((always_inline)) { return 2 * p + 1 < K ? 2 * p + 1 : 2 * p; };
  auto pair = [&](int p) __attribute__((always_inline)) { return g.rows(w, 2 * p, w, r1_of(p), nullptr, nullptr, M); };
  auto epi = [&](int p, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (lane != 0) return;
    const int r0 = 2 * p, r1 = r1_of(p);
    y[r0] = s0 * scale;
    if (r1 != r0) y[r1] = s1 * scale;
  };
  gemv_pairs<1, /*ROLL=*/false>(g, xs, (K + 1) >> 1, lane, nullptr, pair, [](int) __attribute__((always_inline)) { return NoAux{}; },
                       [&]() __attribute__((always_inline)) { st.issue(); }, [&]() __attribute__((always_inline)) { (void)st.finish(xs, 0.f, red); }, epi);
}

// any M / any alignment (also M too large for LDS): wave per row, scalar lane-strided loads
__global__ __launch_bounds__(KH_WG) void k_matmul_f32_generic(const float* __restrict__ x,
                                                              const float* __restrict__ w,
                                                              float* __restrict__ y, int M,
                                                              int K, float scale) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * KH_WAVES_PER_WG + (threadIdx.x >> 6);
  const int nw = gridDim.x * KH_WAVES_PER_WG;
  for (int r = gw; r < K; r += nw) {
    const float* row = w + (size_t)r * M;
    float a = 0.f;
    for (int i = lane; i < M; i += KH_WAVE) a = __builtin_fmaf(row[i], x[i], a);
    a = wave_sum(a);
    if (lane == 0) y[r] = a * scale;
  }
}

static inline int gemv_grid(int nitems_per_wave_unit) {
  int g = (nitems_per_wave_unit + KH_WAVES_PER_WG - 1) / KH_WAVES_PER_WG;
  if (g < 1) g = 1;
  if (g > 1024) g = 1024;  // 4 WGs per CU x 256 CUs, grid-stride beyond that
  return g;
}

extern "C" int kh_matmul_f32(const float* x, const float* w, float* y, int32_t M, int32_t K,
                             float scale, void* stream) {
  if (!x || !w || !y || M <= 0 || K <= 0) return KH_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const bool vec = (M % 4 == 0) && kh_aligned16(x) && kh_aligned16(w) &&
                   ((size_t)M * 4 + 16 <= 64 * 1024);
  if (!vec) {
    hipLaunchKernelGGL(k_matmul_f32_generic, dim3(gemv_grid(K)), dim3(KH_WG), 0, s, x, w, y, M,
                       K, scale);
    return kh_launch_status();
  }
  const size_t lds = (size_t)M * 4 + 16;
  const int grid = gemv_grid((K + 1) / 2);
  const int per_lane = (M / 4 + KH_WAVE - 1) / KH_WAVE;
#define KH_MM(UU, MV) \
  hipLaunchKernelGGL((k_matmul_f32<UU, MV>), dim3(grid), dim3(KH_WG), lds, s, x, w, y, M, K, scale)
  const bool inreg = kh_stage_fits4(M);
  if (per_lane >= 8) {
    if (inreg) KH_MM(8, 4); else KH_MM(8, 0);
  } else if (per_lane >= 3) {
    if (inreg) KH_MM(4, 4); else KH_MM(4, 0);
  } else {
    KH_MM(2, 4);  // per_lane < 3 implies M <= 512
  }
#undef KH_MM
  return kh_launch_status();
}

// =============================================================================================
// matmul int8 group-dequant.  reference: cuda/matmul_kernel.cu:56-87 (scalar byte loads, an
// integer divide and a scale load per element).
template <int U, int MAXV>
__global__ __launch_bounds__(KH_WG) void k_matmul_q8(const float* __restrict__ x,
                                                     const int8_t* __restrict__ w,
                                                     const float* __restrict__ scales,
                                                     int gshift, float* __restrict__ y, int M,
                                                     int K) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  const int M16 = M >> 4;
  float* red = (float*)(xs + 4 * (M16 + 1));
  const int lane = threadIdx.x & 63;
  const Gemv<true, U> g(M, gshift);
  Stager<false, true, MAXV> st(x, nullptr, M);
  auto r1_of = [&](int p) __attribute__((always_inline)) { return 2 * p + 1 < K ? 2 * p + 1 : 2 * p; };
  auto pair = [&](int p) __attribute__((always_inline)) { return g.rows(w, 2 * p, w, r1_of(p), scales, scales, M); };
  auto epi = [&](int p, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (lane != 0) return;
    const int r0 = 2 * p, r1 = r1_of(p);
    y[r0] = s0;
    if (r1 != r0) y[r1] = s1;
  };
  gemv_pairs<1, /*ROLL=*/false>(g, xs, (K + 1) >> 1, lane, nullptr, pair, [](int) __attribute__((always_inline)) { return NoAux{}; },
                      [&]() __attribute__((always_inline)) { st.issue(); }, [&]() __attribute__((always_inline)) { (void)st.finish(xs, 0.f, red); }, epi);
}

// literal restatement of the reference formula for any M/group/alignment
__global__ __launch_bounds__(KH_WG) void k_matmul_q8_generic(const float* __restrict__ x,
                                                             const int8_t* __restrict__ w,
                                                             const float* __restrict__ scales,
                                                             int group, float* __restrict__ y,
                                                             int M, int K) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * KH_WAVES_PER_WG + (threadIdx.x >> 6);
  const int nw = gridDim.x * KH_WAVES_PER_WG;
  for (int r = gw; r < K; r += nw) {
    const size_t base = (size_t)r * M;
    float a = 0.f;
    for (int i = lane; i < M; i += KH_WAVE) {
      const size_t widx = base + i;
      a += x[i] * scales[widx / (size_t)group] * (float)w[widx];
    }
    a = wave_sum(a);
    if (lane == 0) y[r] = a;
  }
}

static inline int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

extern "C" int kh_matmul_q8(const float* x, const int8_t* w8, const float* scales,
                            int32_t group_size, float* y, int32_t M, int32_t K, void* stream) {
  if (!x || !w8 || !scales || !y || M <= 0 || K <= 0 || group_size <= 0)
    return KH_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  const int gshift = ilog2_exact(group_size);
  const bool vec = gshift >= 4 && (M % 16 == 0) && (M % group_size == 0) && kh_aligned16(x) &&
                   kh_aligned16(w8) && ((((uintptr_t)scales) & 3u) == 0) &&
                   (kh_q8_lds_bytes(M) + 16 <= 64 * 1024);
  if (!vec) {
    hipLaunchKernelGGL(k_matmul_q8_generic, dim3(gemv_grid(K)), dim3(KH_WG), 0, s, x, w8,
                       scales, group_size, y, M, K);
    return kh_launch_status();
  }
  const size_t lds = kh_q8_lds_bytes(M) + 16;
  const int grid = gemv_grid((K + 1) / 2);
  const int per_lane = (M / 16 + KH_WAVE - 1) / KH_WAVE;
#define KH_MMQ(UU, MV)                                                                        \
  hipLaunchKernelGGL((k_matmul_q8<UU, MV>), dim3(grid), dim3(KH_WG), lds, s, x, w8, scales, gshift, \
                     y, M, K)
  const bool inreg = kh_stage_fits4(M);
  if (per_lane >= 3) {
    if (inreg) KH_MMQ(4, 4); else KH_MMQ(4, 0);
  } else {
    KH_MMQ(2, 4);  // per_lane < 3 implies M <= 2048
  }
#undef KH_MMQ
  return kh_launch_status();
}

// =============================================================================================
// embedding gather.  reference: cuda/emb_kernel.cu:3-21 (fixed 512 blocks => prompts > 512
// tokens silently truncated; here the grid covers every token).
__global__ __launch_bounds__(KH_WG) void k_embedding(const int32_t* __restrict__ tokens,
                                                     const float* __restrict__ w,
                                                     float* __restrict__ out, int dim,
                                                     int vocab, int vec_ok) {
  const int t = blockIdx.x;
  const int token = tokens[t];
  if (token < 0 || token >= vocab) return;
  const float* src = w + (size_t)token * dim;
  float* dst = out + (size_t)t * dim;
  if (vec_ok) {
    const f32x4* s4 = (const f32x4*)src;
    f32x4* d4 = (f32x4*)dst;
    for (int i = threadIdx.x; i < (dim >> 2); i += KH_WG) d4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < dim; i += KH_WG) dst[i] = src[i];
  }
}
extern "C" int kh_embedding_f32(const int32_t* tokens, int32_t n_tokens, const float* w,
                                float* out, int32_t dim, int32_t vocab, void* stream) {
  if (!tokens || !w || !out || n_tokens <= 0 || dim <= 0 || vocab <= 0)
    return KH_ERR_INVALID_ARG;
  const int vec = (dim % 4 == 0) && kh_aligned16(w) && kh_aligned16(out);
  hipLaunchKernelGGL(k_embedding, dim3(n_tokens), dim3(KH_WG), 0, (hipStream_t)stream, tokens,
                     w, out, dim, vocab, vec);
  return kh_launch_status();
}

// The reference hands the embedding kernel a HOST tensor of token ids and uploads it per call
// (cuda/emb_kernel.cu:25-29: clone + cudaMalloc + cudaMemcpy).  Here the ids of up to
// KH_EMB_BATCH tokens travel BY VALUE in the kernel arguments: no staging buffer, no H2D copy,
// nothing to own, and the launch stays graph-capturable; longer token lists take several launches.
#define KH_EMB_BATCH 64
struct KhEmbTokens {
  int32_t t[KH_EMB_BATCH];
};
__global__ __launch_bounds__(KH_WG) void k_embedding_args(const KhEmbTokens tk,
                                                          const float* __restrict__ w,
                                                          float* __restrict__ out, int dim,
                                                          int vocab, int vec_ok) {
  const int t = blockIdx.x;
  const int token = tk.t[t];
  if (token < 0 || token >= vocab) return;
  const float* src = w + (size_t)token * dim;
  float* dst = out + (size_t)t * dim;
  if (vec_ok) {
    const f32x4* s4 = (const f32x4*)src;
    f32x4* d4 = (f32x4*)dst;
    for (int i = threadIdx.x; i < (dim >> 2); i += KH_WG) d4[i] = s4[i];
  } else {
    for (int i = threadIdx.x; i < dim; i += KH_WG) dst[i] = src[i];
  }
}
extern "C" int kh_embedding_f32_host(const int32_t* h_tokens, int32_t n_tokens, const float* w,
                                     float* out, int32_t dim, int32_t vocab, void* stream) {
  if (!h_tokens || !w || !out || n_tokens <= 0 || dim <= 0 || vocab <= 0)
    return KH_ERR_INVALID_ARG;
  const int vec = (dim % 4 == 0) && kh_aligned16(w) && kh_aligned16(out);
  for (int t0 = 0; t0 < n_tokens; t0 += KH_EMB_BATCH) {
    const int n = n_tokens - t0 < KH_EMB_BATCH ? n_tokens - t0 : KH_EMB_BATCH;
    KhEmbTokens tk;
    for (int i = 0; i < KH_EMB_BATCH; ++i) tk.t[i] = i < n ? h_tokens[t0 + i] : -1;
    hipLaunchKernelGGL(k_embedding_args, dim3(n), dim3(KH_WG), 0, (hipStream_t)stream, tk, w,
                       out + (size_t)t0 * dim, dim, vocab, vec);
  }
  return kh_launch_status();
}

// =============================================================================================
// rmsnorm.  reference: cuda/rmsnorm_kernel.cu:5-50 (1 block x 128 thr, rsqrtf); arithmetic here
// follows the CPU backend (cpu/rmsnorm_kernel.cpp:24-32): 1/sqrt, then w * (r * x).
__global__ __launch_bounds__(KH_WG) void k_rmsnorm(const float* x, const float* __restrict__ w,
                                                   float* out, int n, float eps, int vec_ok) {
  __shared__ float red[KH_WAVES_PER_WG];
  float ss = 0.f;
  if (vec_ok) {
    const f32x4* x4 = (const f32x4*)x;
    for (int i = threadIdx.x; i < (n >> 2); i += KH_WG) {
      const f32x4 v = x4[i];
      ss = fma4(v, v, ss);
    }
  } else {
    for (int i = threadIdx.x; i < n; i += KH_WG) ss = __builtin_fmaf(x[i], x[i], ss);
  }
  ss = block_sum(ss, red);
  const float rs = 1.0f / sqrtf(ss / (float)n + eps);
  // out may alias x: each element is read and written by the same thread only
  if (vec_ok) {
    const f32x4* x4 = (const f32x4*)x;
    const f32x4* w4 = (const f32x4*)w;
    f32x4* o4 = (f32x4*)out;
    for (int i = threadIdx.x; i < (n >> 2); i += KH_WG) {
      const f32x4 v = x4[i], g = w4[i];
      f32x4 r;
      r.x = g.x * (rs * v.x);
      r.y = g.y * (rs * v.y);
      r.z = g.z * (rs * v.z);
      r.w = g.w * (rs * v.w);
      o4[i] = r;
    }
  } else {
    for (int i = threadIdx.x; i < n; i += KH_WG) out[i] = w[i] * (rs * x[i]);
  }
}
extern "C" int kh_rmsnorm_f32(const float* x, const float* w, float* out, int32_t n, float eps,
                              void* stream) {
  if (!x || !w || !out || n <= 0) return KH_ERR_INVALID_ARG;
  const int vec = (n % 4 == 0) && kh_aligned16(x) && kh_aligned16(w) && kh_aligned16(out);
  hipLaunchKernelGGL(k_rmsnorm, dim3(1), dim3(KH_WG), 0, (hipStream_t)stream, x, w, out, n, eps,
                     vec);
  return kh_launch_status();
}

// =============================================================================================
// RoPE.  reference: cuda/rope_kernel.cu:5-36 / 51-82 (half; note its `idx > total_pairs`
// off-by-one guard at :13,:59 is NOT reproduced) and :104-122 (interleaved).
__global__ __launch_bounds__(KH_WG) void k_rope(int dim, int kv_dim, int head_size,
                                                float* __restrict__ q, float* __restrict__ k,
                                                const int32_t* __restrict__ d_pos, int pos_val,
                                                const float* __restrict__ sin_cache,
                                                const float* __restrict__ cos_cache, int mode) {
  const int pos = d_pos ? *d_pos : pos_val;
  const int pair = blockIdx.x * KH_WG + threadIdx.x;
  if (pair >= dim / 2) return;
  int i0, i1, cidx;
  if (mode == KH_ROPE_HALF) {
    const int half = head_size >> 1;
    const int head = pair / half, j = pair - head * half;
    i0 = head * head_size + j;
    i1 = i0 + half;
    cidx = 2 * j;
  } else {
    i0 = 2 * pair;
    i1 = i0 + 1;
    cidx = i0 % head_size;
  }
  const float fci = sin_cache[(size_t)pos * head_size + cidx];
  const float fcr = cos_cache[(size_t)pos * head_size + cidx];
  {
    const float v0 = q[i0], v1 = q[i1];
    q[i0] = v0 * fcr - v1 * fci;
    q[i1] = v0 * fci + v1 * fcr;
  }
  // the reference rotates k while the (first) index is < kv_dim
  if ((mode == KH_ROPE_HALF ? (i0 / head_size) * head_size : i0) < kv_dim) {
    const float v0 = k[i0], v1 = k[i1];
    k[i0] = v0 * fcr - v1 * fci;
    k[i1] = v0 * fci + v1 * fcr;
  }
}
extern "C" int kh_rope_f32(int32_t dim, int32_t kv_dim, int32_t head_size, float* q, float* k,
                           const int32_t* d_pos, int32_t pos, const float* sin_cache,
                           const float* cos_cache, int32_t mode, void* stream) {
  if (!q || !k || !sin_cache || !cos_cache || dim <= 0 || kv_dim <= 0 || head_size <= 0 ||
      (head_size & 1) || dim % head_size || kv_dim % head_size || kv_dim > dim ||
      (mode != KH_ROPE_HALF && mode != KH_ROPE_INTERLEAVED) || (!d_pos && pos < 0))
    return KH_ERR_INVALID_ARG;
  const int pairs = dim / 2;
  hipLaunchKernelGGL(k_rope, dim3((pairs + KH_WG - 1) / KH_WG), dim3(KH_WG), 0,
                     (hipStream_t)stream, dim, kv_dim, head_size, q, k, d_pos, pos, sin_cache,
                     cos_cache, mode);
  return kh_launch_status();
}

// sin/cos cache.  reference: cuda/rope_kernel.cu:38-49 (1 block x head_size threads looping
// over all positions).  freq/angle are formed in fp32 exactly like the CPU backend
// (cpu/rope_kernel.cpp:7-13): freq = 1/powf(theta, d/hs); val = float(pos)*freq.  pow and
// sin/cos are evaluated in fp64 and rounded once, which reproduces glibc's (correctly
// rounded in practice) powf/sinf/cosf — the fp32 product pos*freq is what loses bits at long
// context and that rounding is kept.
__global__ __launch_bounds__(KH_WG) void k_sincos(int head_size, int max_seq_len, float theta,
                                                  float* __restrict__ sin_cache,
                                                  float* __restrict__ cos_cache) {
  const size_t total = (size_t)max_seq_len * head_size;
  for (size_t i = (size_t)blockIdx.x * KH_WG + threadIdx.x; i < total;
       i += (size_t)gridDim.x * KH_WG) {
    const int pos = (int)(i / head_size);
    const int d = (int)(i - (size_t)pos * head_size);
    const float e = (float)d / (float)head_size;
    const float p = (float)pow((double)theta, (double)e);
    const float freq = 1.0f / p;
    const float val = (float)pos * freq;
    sin_cache[i] = (float)sin((double)val);
    cos_cache[i] = (float)cos((double)val);
  }
}
extern "C" int kh_sincos_cache_f32(int32_t head_size, int32_t max_seq_len, float theta,
                                   float* sin_cache, float* cos_cache, void* stream) {
  if (!sin_cache || !cos_cache || head_size <= 0 || max_seq_len <= 0 || !(theta > 0.f))
    return KH_ERR_INVALID_ARG;
  const size_t total = (size_t)max_seq_len * head_size;
  size_t g = (total + KH_WG - 1) / KH_WG;
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(k_sincos, dim3((unsigned)g), dim3(KH_WG), 0, (hipStream_t)stream,
                     head_size, max_seq_len, theta, sin_cache, cos_cache);
  return kh_launch_status();
}

// =============================================================================================
// multi-head attention (decode).  Device core in kh_attn.h.
__global__ __launch_bounds__(KH_WG) void k_mha(const int32_t* __restrict__ d_pos, int pos_val,
                                               int layer_index, int seq_len, int kv_dim,
                                               int kv_mul, int head_size,
                                               float* __restrict__ mha_out,
                                               const float* __restrict__ q,
                                               float* __restrict__ score,
                                               const float* __restrict__ kcache,
                                               const float* __restrict__ vcache) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int pos = d_pos ? *d_pos : pos_val;
  const int h = blockIdx.x;
  const size_t layer_off = (size_t)layer_index * (size_t)seq_len * (size_t)kv_dim;
  const size_t head_off = (size_t)(h / kv_mul) * head_size;
  attn_head_decode(q + (size_t)h * head_size, kcache + layer_off + head_off,
                   vcache + layer_off + head_off, kv_dim, head_size, pos,
                   mha_out + (size_t)h * head_size, score ? score + (size_t)h * seq_len : nullptr,
                   (float*)smem_raw);
}

// decode attention through the launch the fused step uses (kh_attn.h); ns_g == 0: per-head only
static int launch_mha_fast(const int32_t* d_pos, int32_t pos, int32_t head_num,
                           int32_t layer_index, int32_t seq_len, int32_t kv_dim, int32_t kv_mul,
                           int32_t head_size, float* mha_out, const float* q, const float* kcache,
                           const float* vcache, int nsplit, int nsplit_g, int ws_stride,
                           int t_long, void* ws, hipStream_t s) {
  const size_t layer_off = (size_t)layer_index * (size_t)seq_len * (size_t)kv_dim;
  KhAttnArgs a;
  a.q = q;
  a.kcache_layer = kcache + layer_off;
  a.vcache_layer = vcache + layer_off;
  a.out = mha_out;
  a.d_pos = d_pos;
  a.kv_dim = kv_dim;
  a.kv_mul = kv_mul;
  a.head_size = head_size;
  a.kv_heads = head_num / kv_mul;
  a.nsplit = nsplit;
  a.ws = ws;
  a.ws_stride = ws_stride;
  a.nsplit_g = nsplit_g;
  a.t_long = t_long;
  a.ts_shift = attn_ts_shift_for(head_size);
  a.defer = 0;  // operator level: the launch leaves the final output
  {
    const char* e = khm::dbg("KH_ATTN_FENCED");
    a.fenced = (e && e[0] == '1') ? 1 : 0;
  }
  a.tok_stride = 0;
  a.ws_tok_bytes = 0;
  launch_attn_decode(a, pos, KH_WG_MAX, s);
  return kh_launch_status();
}

extern "C" int kh_mha_f32(const int32_t* d_pos, int32_t pos, int32_t head_num,
                          int32_t layer_index, int32_t seq_len, int32_t kv_dim, int32_t kv_mul,
                          int32_t head_size, float* mha_out, const float* q, float* score,
                          const float* kcache, const float* vcache, void* stream) {
  if (!mha_out || !q || !kcache || !vcache || head_num <= 0 || layer_index < 0 ||
      seq_len <= 0 || kv_dim <= 0 || kv_mul <= 0 || head_size <= 0 || head_size % 4 ||
      head_size > 256 || kv_dim % 4 || (!d_pos && (pos < 0 || pos >= seq_len)) ||
      !kh_aligned16(q) || !kh_aligned16(kcache) || !kh_aligned16(vcache) ||
      !kh_aligned16(mha_out))
    return KH_ERR_INVALID_ARG;
  hipStream_t s = (hipStream_t)stream;
  if (!score && head_size > 32 && head_num % kv_mul == 0)
    // no score tensor requested: the fused decode path's one-round-trip kernel (kh_attn.h),
    // one workgroup per head (kh_mha_decode_f32 adds the long-context time split)
    return launch_mha_fast(d_pos, pos, head_num, layer_index, seq_len, kv_dim, kv_mul, head_size,
                           mha_out, q, kcache, vcache, 1, 0, 1, 1 << 30, nullptr, s);
  hipLaunchKernelGGL(k_mha, dim3(head_num), dim3(KH_WG), attn_lds_bytes(head_size), s, d_pos,
                     pos, layer_index, seq_len, kv_dim, kv_mul, head_size, mha_out, q, score,
                     kcache, vcache);
  return kh_launch_status();
}

// Decode attention with the long-context time split (the kernel the fused step launches).
static void mha_decode_geometry(int head_num, int kv_mul, int head_size, int seq_len, int* ns,
                                int* ns_g, int* stride, int* t_long) {
  const AttnPlan p = attn_plan(head_num, kv_mul, head_size, seq_len, KH_WG_MAX, attn_tlong_hook());
  *ns = p.ns;
  *ns_g = p.ns_g;
  *stride = p.stride;
  *t_long = p.t_long;
}
static int64_t mha_decode_workspace_bytes_for(int32_t head_num, int32_t kv_mul, int32_t head_size,
                                              int32_t seq_len) {
  if (head_num <= 0 || head_size <= 0 || seq_len <= 0 || kv_mul <= 0) return KH_ERR_INVALID_ARG;
  int ns, ns_g, stride, tl;
  mha_decode_geometry(head_num, kv_mul, head_size, seq_len, &ns, &ns_g, &stride, &tl);
  return (int64_t)attn_ws_bytes(head_num, head_size, stride);
}
extern "C" int64_t kh_mha_decode_workspace_bytes(int32_t head_num, int32_t head_size,
                                                 int32_t seq_len) {
  // kv_mul unknown: size for the widest slot stride any geometry can ask for
  if (head_num <= 0 || head_size <= 0 || seq_len <= 0) return KH_ERR_INVALID_ARG;
  int64_t best = 0;
  for (int kvm : {1, 2, 4, 7, 8})
    if (head_num % kvm == 0) {
      const int64_t b = mha_decode_workspace_bytes_for(head_num, kvm, head_size, seq_len);
      if (b > best) best = b;
    }
  return best;
}
// Host-only view of the decode-attention geometry (tools, CPU test-suite): out8 = {ns, ns_g, slot stride, t_long,
// path at `pos` (0 = per-head workgroups, 1 = GQA group path), active splits at `pos`, timesteps per split at `pos`,
// workgroups that own timesteps at `pos`}.
extern "C" int kh_plan_attention(int32_t head_num, int32_t kv_mul, int32_t head_size, int32_t seq_len, int32_t pos,
                                 int32_t* out8) {
  if (!out8 || head_num <= 0 || kv_mul <= 0 || head_num % kv_mul || head_size <= 0 || seq_len <= 0 || pos < 0 ||
      pos >= seq_len)
    return KH_ERR_INVALID_ARG;
  int ns, ns_g, stride, tl;
  mha_decode_geometry(head_num, kv_mul, head_size, seq_len, &ns, &ns_g, &stride, &tl);
  const bool grp = ns_g > 0 && pos + 1 >= tl;
  const int NS = grp ? ns_g : ns;
  int nact, TS;
  attn_split_geometry(pos, NS, grp ? KH_ATTN_TSG_SHIFT : attn_ts_shift_for(head_size), TS, nact);
  out8[0] = ns;
  out8[1] = ns_g;
  out8[2] = stride;
  out8[3] = tl;
  out8[4] = grp ? 1 : 0;
  out8[5] = nact;
  out8[6] = TS;
  out8[7] = (grp ? head_num / kv_mul : head_num) * nact;
  return KH_OK;
}
extern "C" int kh_mha_decode_f32(const int32_t* d_pos, int32_t pos, int32_t head_num,
                                 int32_t layer_index, int32_t seq_len, int32_t kv_dim,
                                 int32_t kv_mul, int32_t head_size, float* mha_out, const float* q,
                                 const float* kcache, const float* vcache, void* workspace,
                                 int64_t workspace_bytes, void* stream) {
  if (head_size <= 32 || kv_mul <= 0 || head_num % kv_mul)
    return kh_mha_f32(d_pos, pos, head_num, layer_index, seq_len, kv_dim, kv_mul, head_size,
                      mha_out, q, nullptr, kcache, vcache, stream);
  int ns, ns_g, stride, tl;
  mha_decode_geometry(head_num, kv_mul, head_size, seq_len, &ns, &ns_g, &stride, &tl);
  if (!mha_out || !q || !kcache || !vcache || head_num <= 0 || layer_index < 0 || seq_len <= 0 ||
      kv_dim <= 0 || head_size % 4 || head_size > 256 || kv_dim % 4 ||
      (!d_pos && (pos < 0 || pos >= seq_len)) || !kh_aligned16(q) || !kh_aligned16(kcache) ||
      !kh_aligned16(vcache) || !kh_aligned16(mha_out) ||
      (stride > 1 &&
       (!workspace || workspace_bytes < (int64_t)attn_ws_bytes(head_num, head_size, stride) ||
        !kh_aligned16(workspace))))
    return KH_ERR_INVALID_ARG;
  return launch_mha_fast(d_pos, pos, head_num, layer_index, seq_len, kv_dim, kv_mul, head_size,
                         mha_out, q, kcache, vcache, ns, ns_g, stride, tl, workspace,
                         (hipStream_t)stream);
}

// =============================================================================================
// softmax in place (single vector).  reference: cpu/softmax_kernel.cpp:4-15 (CPU only).
__global__ __launch_bounds__(KH_WG) void k_softmax(float* __restrict__ x, int n) {
  __shared__ float red[KH_WAVES_PER_WG];
  float mx = -INFINITY;
  for (int i = threadIdx.x; i < n; i += KH_WG) mx = fmaxf(mx, x[i]);
  mx = block_max(mx, red);
  float s = 0.f;
  for (int i = threadIdx.x; i < n; i += KH_WG) {
    const float e = expf(x[i] - mx);
    x[i] = e;
    s += e;
  }
  s = block_sum(s, red);
  for (int i = threadIdx.x; i < n; i += KH_WG) x[i] = x[i] / s;
}
extern "C" int kh_softmax_f32(float* x, int32_t n, void* stream) {
  if (!x || n <= 0) return KH_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_softmax, dim3(1), dim3(KH_WG), 0, (hipStream_t)stream, x, n);
  return kh_launch_status();
}

// out[d] += sum_{t<=pos} scale[t]*value[t*stride+d], ascending t per output element
// reference: cpu/scale_sum_kernel.cpp:5-22
__global__ __launch_bounds__(KH_WG) void k_scale_sum(const float* __restrict__ value,
                                                     const float* __restrict__ scale,
                                                     float* __restrict__ out, int pos, int size,
                                                     int stride) {
  for (int d = blockIdx.x * KH_WG + threadIdx.x; d < size; d += gridDim.x * KH_WG) {
    float a = out[d];
    for (int t = 0; t <= pos; ++t) a += scale[t] * value[(size_t)t * stride + d];
    out[d] = a;
  }
}
extern "C" int kh_scale_sum_f32(const float* value, const float* scale, float* out, int32_t pos,
                                int32_t size, int32_t stride, void* stream) {
  if (!value || !scale || !out || pos < 0 || size <= 0 || stride < size)
    return KH_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_scale_sum, dim3((size + KH_WG - 1) / KH_WG), dim3(KH_WG), 0,
                     (hipStream_t)stream, value, scale, out, pos, size, stride);
  return kh_launch_status();
}

// =============================================================================================
// argmax.  reference: cuda/argmax_kernel.cu:5-71 uses 32-lane __shfl_down_sync/ballot masks
// and shared[32]; rebuilt for wave64: one 1024-thread WG = 16 waves.
#define KH_ARGMAX_THREADS 1024
__global__ __launch_bounds__(KH_ARGMAX_THREADS) void k_argmax(const float* __restrict__ logits,
                                                              long long n,
                                                              int32_t* __restrict__ out) {
  __shared__ float sv[KH_ARGMAX_THREADS / KH_WAVE];
  __shared__ int si[KH_ARGMAX_THREADS / KH_WAVE];
  float v = -INFINITY;
  int idx = 0x7fffffff;
  for (long long i = threadIdx.x; i < n; i += KH_ARGMAX_THREADS) {
    const float x = logits[i];
    if (x > v || idx == 0x7fffffff) {  // strictly greater: keeps the first occurrence
      v = x;
      idx = (int)i;
    }
  }
  wave_amax(v, idx);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    sv[wave] = v;
    si[wave] = idx;
  }
  __syncthreads();
  if (wave == 0) {
    v = lane < KH_ARGMAX_THREADS / KH_WAVE ? sv[lane] : -INFINITY;
    idx = lane < KH_ARGMAX_THREADS / KH_WAVE ? si[lane] : 0x7fffffff;
    wave_amax(v, idx);
    if (lane == 0) *out = idx;
  }
}
extern "C" int kh_argmax_f32(const float* logits, int64_t n, int32_t* d_out_index,
                             void* stream) {
  if (!logits || !d_out_index || n <= 0 || n > 0x7fffffffLL) return KH_ERR_INVALID_ARG;
  hipLaunchKernelGGL(k_argmax, dim3(1), dim3(KH_ARGMAX_THREADS), 0, (hipStream_t)stream, logits,
                     (long long)n, d_out_index);
  return kh_launch_status();
}
// The reference's sampler path (ArgmaxSampler::sample -> argmax_kernel_cu, argmax_kernel.cu:53-77) allocates 8 bytes
// per token and never frees them; rounds 1-4 of this library allocated and freed 4 bytes per call, i.e. a
// device-wide synchronisation per token.  One 4-byte device word per (device, stream), created on first use and
// kept for the life of the process (a few bytes per stream ever used).  Calls on different streams use different
// words; two host threads calling on the SAME stream (the null stream, say) are serialised by the word's own mutex,
// held from the kernel launch to the stream synchronisation - otherwise kernel A, kernel B, copy A, copy B would hand
// caller A the index of B's logits.
namespace {
struct ArgmaxSlot {
  int device;
  void* stream;
  int32_t* d;
  std::mutex busy;
  ArgmaxSlot(int dev, void* s, int32_t* p) : device(dev), stream(s), d(p) {}
};
std::mutex g_argmax_mu;
std::deque<ArgmaxSlot> g_argmax_slots;  // deque: slots never move once handed out
int argmax_slot(void* stream, ArgmaxSlot** out) {
  int dev = 0;
  KH_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> g(g_argmax_mu);
  for (auto& s : g_argmax_slots)
    if (s.device == dev && s.stream == stream) {
      *out = &s;
      return KH_OK;
    }
  int32_t* d = nullptr;
  KH_CHECK_HIP(hipMalloc((void**)&d, sizeof(int32_t)));
  g_argmax_slots.emplace_back(dev, stream, d);
  *out = &g_argmax_slots.back();
  return KH_OK;
}
}  // namespace
extern "C" int kh_argmax_f32_host(const float* logits, int64_t n, int64_t* h_out_index,
                                  void* stream) {
  if (!h_out_index) return KH_ERR_INVALID_ARG;
  ArgmaxSlot* slot = nullptr;
  int rc = argmax_slot(stream, &slot);
  if (rc != KH_OK) return rc;
  std::lock_guard<std::mutex> busy(slot->busy);
  rc = kh_argmax_f32(logits, n, slot->d, stream);
  int32_t h = -1;
  if (rc == KH_OK) {
    hipError_t e = hipMemcpyAsync(&h, slot->d, sizeof(int32_t), hipMemcpyDeviceToHost,
                                  (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    rc = e == hipSuccess ? KH_OK : (int)e;
  }
  *h_out_index = h;
  return rc;
}

// =============================================================================================
extern "C" const char* kh_error_string(int code) {
  switch (code) {
    case KH_OK: return "success";
    case KH_ERR_INVALID_ARG: return "invalid argument";
    case KH_ERR_UNSUPPORTED: return "unsupported configuration";
    case KH_ERR_IO: return "file i/o error";
    case KH_ERR_FORMAT: return "malformed model image";
    case KH_ERR_NO_DEVICE: return "no HIP device";
    case KH_ERR_RANGE: return "token or position out of range";
    case KH_ERR_INTERNAL: return "C++ exception inside the library (caught at the boundary)";
    default: break;
  }
  if (code > 0) return hipGetErrorString((hipError_t)code);
  return "unknown error";
}
extern "C" int kh_version(void) { return KH_VERSION; }
extern "C" int kh_device_count(void) {
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    return e == hipErrorNoDevice ? 0 : -(int)e;
  }
  return n;
}
