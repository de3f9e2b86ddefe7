/*
 * kuiper_oracle.c — CPU restatement of KuiperLLama's CPU decode path (see kuiper_oracle.h).
 * TEST INFRASTRUCTURE ONLY — never linked into the product library.
 *
 * Build: oracle/Makefile  (gcc -O3 -march=x86-64-v3 -ffp-contract=off -fopenmp).
 * -ffp-contract=off + fixed-order blocked sums make results identical on any AVX2 host,
 * so the same .so gives the same numbers in the build container and on the GPU box.
 *
 * Summation order note: the reference computes every dot product through Armadillo ->
 * BLAS sgemv/sdot (cpu/matmul_kernel.cpp:37-40), whose internal order is unspecified and
 * whose version is unpinned.  dot_f32() below uses a fixed 16-way blocked
 * order (what a vectorised BLAS kernel does); dot_f64() is the fp64-accumulated "gold"
 * used to judge which of {HIP, this fp32 restatement} is closer to exact.
 */
#include "kuiper_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static int g_threads = 0;
void ko_set_threads(int n) {
  g_threads = n;
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#endif
}
int ko_get_threads(void) {
#ifdef _OPENMP
  return g_threads > 0 ? g_threads : omp_get_max_threads();
#else
  return 1;
#endif
}

/* ---- fixed-order reductions ------------------------------------------------------- */
#define KO_LANES 16
static inline float tree16_f32(const float* a) {
  float b0 = a[0] + a[8], b1 = a[1] + a[9], b2 = a[2] + a[10], b3 = a[3] + a[11];
  float b4 = a[4] + a[12], b5 = a[5] + a[13], b6 = a[6] + a[14], b7 = a[7] + a[15];
  float c0 = b0 + b4, c1 = b1 + b5, c2 = b2 + b6, c3 = b3 + b7;
  return (c0 + c2) + (c1 + c3);
}
static inline double tree16_f64(const double* a) {
  double b0 = a[0] + a[8], b1 = a[1] + a[9], b2 = a[2] + a[10], b3 = a[3] + a[11];
  double b4 = a[4] + a[12], b5 = a[5] + a[13], b6 = a[6] + a[14], b7 = a[7] + a[15];
  double c0 = b0 + b4, c1 = b1 + b5, c2 = b2 + b6, c3 = b3 + b7;
  return (c0 + c2) + (c1 + c3);
}
static float dot_f32(const float* a, const float* b, int n) {
  float acc[KO_LANES] = {0};
  int i = 0;
  for (; i + KO_LANES <= n; i += KO_LANES)
    for (int j = 0; j < KO_LANES; ++j) acc[j] += a[i + j] * b[i + j];
  for (int j = 0; i < n; ++i, ++j) acc[j] += a[i] * b[i];
  return tree16_f32(acc);
}
static float dot_f64(const float* a, const float* b, int n) {
  double acc[KO_LANES] = {0};
  int i = 0;
  for (; i + KO_LANES <= n; i += KO_LANES)
    for (int j = 0; j < KO_LANES; ++j) acc[j] += (double)a[i + j] * (double)b[i + j];
  for (int j = 0; i < n; ++i, ++j) acc[j] += (double)a[i] * (double)b[i];
  return (float)tree16_f64(acc);
}
/* arma::accu / arma::sum on a contiguous fvec: two interleaved accumulators
 * (armadillo arrayops::accumulate), used by mean() in rmsnorm and sum() in softmax. */
static float arma_accumulate(const float* x, int n) {
  float acc1 = 0.f, acc2 = 0.f;
  int i, j;
  for (i = 0, j = 1; j < n; i += 2, j += 2) {
    acc1 += x[i];
    acc2 += x[j];
  }
  if (i < n) acc1 += x[i];
  return acc1 + acc2;
}

/* ---- cpu/matmul_kernel.cpp:5-41 ---------------------------------------------------
 * weight is [K rows, M cols] row-major (wei_dim0=K, wei_dim1=M); arma views it as M x K
 * column-major, so output[p] = sum_i input[i]*weight[p*M+i]; then "* scale" (line 40). */
/* Timing-only variant (bench.py cpu_baseline): the reference's CPU matmul is
 * Armadillo -> BLAS sgemv (cpu/matmul_kernel.cpp:37-40), so a caller may plug in a
 * cblas_sgemv-compatible ILP64 entry point (numpy's bundled OpenBLAS).  Never set by tests:
 * BLAS summation order is not the pinned one. */
typedef void (*ko_sgemv_fn)(int order, int trans, int64_t m, int64_t n, float alpha,
                            const float* a, int64_t lda, const float* x, int64_t incx, float beta,
                            float* y, int64_t incy);
static ko_sgemv_fn g_sgemv = NULL;
void ko_set_sgemv(void* fn) { g_sgemv = (ko_sgemv_fn)fn; }

void ko_matmul_f32(const float* x, const float* w, float* y, int M, int K, float scale,
                   int acc) {
  if (g_sgemv && acc == KO_ACC_F32) {
    g_sgemv(101 /*RowMajor*/, 111 /*NoTrans*/, K, M, scale, w, M, x, 1, 0.f, y, 1);
    return;
  }
#pragma omp parallel for schedule(static) if (K >= 64)
  for (int p = 0; p < K; ++p) {
    const float* row = w + (size_t)p * (size_t)M;
    float d = acc == KO_ACC_F64 ? dot_f64(x, row, M) : dot_f32(x, row, M);
    y[p] = d * scale;
  }
}

/* ---- cuda/matmul_kernel.cu:56-87 (the reference has no CPU int8 kernel) -------------
 * term_i = input[i] * scales[(p*M+i)/group_size] * float(weight[p*M+i]), left to right
 * (line 73); the group index runs over the FLATTENED weight (lines 71-72). */
void ko_matmul_q8(const float* x, const int8_t* w, const float* scales, int group, float* y,
                  int M, int K, int acc) {
#pragma omp parallel for schedule(static) if (K >= 64)
  for (int p = 0; p < K; ++p) {
    const size_t base = (size_t)p * (size_t)M;
    if (acc == KO_ACC_F64) {
      double a[KO_LANES] = {0};
      for (int i = 0; i < M; ++i) {
        const size_t widx = base + (size_t)i;
        a[i & (KO_LANES - 1)] +=
            (double)x[i] * (double)scales[widx / (size_t)group] * (double)w[widx];
      }
      y[p] = (float)tree16_f64(a);
    } else {
      float a[KO_LANES] = {0};
      for (int i = 0; i < M; ++i) {
        const size_t widx = base + (size_t)i;
        float t = x[i] * scales[widx / (size_t)group];
        t = t * (float)w[widx];
        a[i & (KO_LANES - 1)] += t;
      }
      y[p] = tree16_f32(a);
    }
  }
}

/* ---- cpu/rmsnorm_kernel.cpp:4-33 ---------------------------------------------------
 * mean = as_scalar(mean(pow(x,2))) + eps; rsqrt = 1/std::sqrt(mean);
 * out = w % (rsqrt * x)   (r*x first, then elementwise * w; lines 30-32).
 * eps is an #ifdef in the reference (1e-6 QWEN2 / 1e-5), a runtime parameter here. */
void ko_rmsnorm_f32(const float* x, const float* w, float* out, int n, float eps) {
  float* sq = (float*)malloc(sizeof(float) * (size_t)n);
  for (int i = 0; i < n; ++i) sq[i] = x[i] * x[i];
  const float mean = arma_accumulate(sq, n) / (float)n + eps;
  free(sq);
  const float rs = 1.f / sqrtf(mean);
  for (int i = 0; i < n; ++i) {
    float t = rs * x[i];
    out[i] = w[i] * t;
  }
}

/* ---- cpu/rope_kernel.cpp:4-16 / 44-56 / 84-96 ---------------------------------------
 * freq = 1.0f / std::pow(theta, float(d)/float(hs))  (float pow)
 * val = float(pos) * freq ; cache[pos*hs + d] = sinf/cosf(val), for EVERY d < head_size.
 * theta is 500000 (LLAMA3), 1000000 (QWEN2) or 10000 in the reference's #ifdef. */
void ko_sincos_cache(int head_size, int max_seq_len, float theta, float* sin_cache,
                     float* cos_cache) {
  float* freq = (float*)malloc(sizeof(float) * (size_t)head_size);
  for (int d = 0; d < head_size; ++d)
    freq[d] = 1.0f / powf(theta, (float)d / (float)head_size);
#pragma omp parallel for schedule(static) if (max_seq_len >= 1024)
  for (int pos = 0; pos < max_seq_len; ++pos) {
    for (int d = 0; d < head_size; ++d) {
      float val = (float)pos * freq[d];
      sin_cache[(size_t)pos * head_size + d] = sinf(val);
      cos_cache[(size_t)pos * head_size + d] = cosf(val);
    }
  }
  free(freq);
}

/* ---- cpu/rope_kernel.cpp:18-42 / 58-82 (half) , 98-121 (interleaved) ----------------
 * half:        pair (i+j, i+j+hs/2) for j<hs/2 in every head i; cache index pos*hs + 2j
 * interleaved: pair (i, i+1) for even i; cache index pos*hs + (i % hs)
 * v0' = v0*c - v1*s ; v1' = v0*s + v1*c ; q always, k only while i < kv_dim. */
void ko_rope_f32(int dim, int kv_dim, int head_size, float* q, float* k, int pos,
                 const float* sin_cache, const float* cos_cache, int mode) {
  const float* srow = sin_cache + (size_t)pos * head_size;
  const float* crow = cos_cache + (size_t)pos * head_size;
  if (mode == KO_ROPE_HALF) {
    const int half = head_size / 2;
    for (int i = 0; i < dim; i += head_size) {
      for (int j = 0; j < half; ++j) {
        const float fci = srow[j * 2], fcr = crow[j * 2];
        const int rotn = i < kv_dim ? 2 : 1;
        for (int v = 0; v < rotn; ++v) {
          float* vec = v == 0 ? q : k;
          const float v0 = vec[i + j], v1 = vec[i + j + half];
          const float a = v0 * fcr, b = v1 * fci, c = v0 * fci, d = v1 * fcr;
          vec[i + j] = a - b;
          vec[i + j + half] = c + d;
        }
      }
    }
  } else {
    for (int i = 0; i < dim; i += 2) {
      const int hd = i % head_size;
      const float fci = srow[hd], fcr = crow[hd];
      const int rotn = i < kv_dim ? 2 : 1;
      for (int v = 0; v < rotn; ++v) {
        float* vec = v == 0 ? q : k;
        const float v0 = vec[i], v1 = vec[i + 1];
        const float a = v0 * fcr, b = v1 * fci, c = v0 * fci, d = v1 * fcr;
        vec[i] = a - b;
        vec[i + 1] = c + d;
      }
    }
  }
}

/* ---- cpu/softmax_kernel.cpp:4-15 ---------------------------------------------------- */
void ko_softmax_f32(float* x, int n) {
  float mx = x[0];
  for (int i = 1; i < n; ++i)
    if (x[i] > mx) mx = x[i];
  for (int i = 0; i < n; ++i) x[i] = expf(x[i] - mx);
  const float s = arma_accumulate(x, n);
  for (int i = 0; i < n; ++i) x[i] = x[i] / s;
}

/* ---- cpu/scale_sum_kernel.cpp:5-22 : ascending t, out += scale[t]*value_t ---------- */
void ko_scale_sum_f32(const float* value, const float* scale, float* out, int pos, int size,
                      int stride) {
  for (int t = 0; t <= pos; ++t) {
    const float s = scale[t];
    const float* v = value + (size_t)t * (size_t)stride;
    for (int d = 0; d < size; ++d) {
      float p = s * v[d];
      out[d] += p;
    }
  }
}

/* ---- cpu/scale_kernel.cpp:3-9 ------------------------------------------------------- */
void ko_scale_f32(float scale, float* x, int n) {
  for (int i = 0; i < n; ++i) x[i] = x[i] * scale;
}

/* ---- cpu/mha_kernel.cpp:5-61 ---------------------------------------------------------
 * per head h: score[t] = (q_h . K[layer, t, (h/kv_mul)*hs : +hs]) * (1/sqrt(hs)) for
 * t<=pos (matmul kernel with scale, lines 28-39); softmax in place (44); zero out_h (47);
 * out_h += sum_t score[t]*V[layer, t, ...] ascending t (58). */
void ko_mha_f32(int pos, int head_num, int layer_index, int seq_len, int kv_dim, int kv_mul,
                int head_size, float* mha_out, const float* q, float* score,
                const float* kcache, const float* vcache, int acc) {
  const size_t layer_offset = (size_t)layer_index * (size_t)seq_len * (size_t)kv_dim;
  const float scale = 1.f / sqrtf((float)head_size);
#pragma omp parallel for schedule(static) if (head_num >= 4 && pos >= 32)
  for (int h = 0; h < head_num; ++h) {
    float* sh = score + (size_t)h * (size_t)seq_len;
    const float* qh = q + (size_t)h * head_size;
    const size_t head_off = (size_t)(h / kv_mul) * head_size;
    for (int t = 0; t <= pos; ++t) {
      const float* kt = kcache + layer_offset + (size_t)t * kv_dim + head_off;
      float d = acc == KO_ACC_F64 ? dot_f64(qh, kt, head_size) : dot_f32(qh, kt, head_size);
      sh[t] = d * scale;
    }
    ko_softmax_f32(sh, pos + 1);
    float* oh = mha_out + (size_t)h * head_size;
    memset(oh, 0, sizeof(float) * (size_t)head_size);
    ko_scale_sum_f32(vcache + layer_offset + head_off, sh, oh, pos, head_size, kv_dim);
  }
}

/* ---- cpu/swiglu_kernel.cpp:3-23 : in1 %= 1/(1+exp(-in1)); out = in1 % in2 ---------- */
void ko_swiglu_f32(const float* a, const float* b, float* out, int n) {
  for (int i = 0; i < n; ++i) {
    const float av = a[i];
    const float sg = 1.0f / (1.0f + expf(-av));
    const float g = av * sg;
    out[i] = g * b[i];
  }
}

/* ---- cpu/add_kernel.cpp:5-19 --------------------------------------------------------- */
void ko_add_f32(const float* a, const float* b, float* out, int n) {
  for (int i = 0; i < n; ++i) out[i] = a[i] + b[i];
}

/* ---- cpu/emb_kernel.cpp:4-29 : rejects token > vocab_size (sic, line 16) ------------ */
int ko_embedding_f32(const int32_t* tokens, int n_tokens, const float* w, float* out, int dim,
                     int vocab) {
  for (int i = 0; i < n_tokens; ++i) {
    const int32_t t = tokens[i];
    if (t > vocab || t < 0) return -1;
    memcpy(out + (size_t)i * dim, w + (size_t)t * dim, sizeof(float) * (size_t)dim);
  }
  return 0;
}

/* ---- argmax_sampler.cpp:7 ------------------------------------------------------------ */
size_t ko_argmax_f32(const float* logits, size_t n) {
  size_t best = 0;
  for (size_t i = 1; i < n; ++i)
    if (logits[i] > logits[best]) best = i;
  return best;
}

/* ---- tools/export.py:49-73 quantize_q80 ---------------------------------------------- */
void ko_quantize_q80(const float* w, size_t n, int group, int8_t* q, float* scales) {
  const size_t ng = n / (size_t)group;
#pragma omp parallel for schedule(static) if (ng >= 1024)
  for (size_t g = 0; g < ng; ++g) {
    const float* wg = w + g * (size_t)group;
    float wmax = 0.f;
    for (int i = 0; i < group; ++i) {
      float a = fabsf(wg[i]);
      if (a > wmax) wmax = a;
    }
    const float scale = wmax / 127.0f;
    scales[g] = scale;
    for (int i = 0; i < group; ++i) {
      float v = wg[i] / scale;
      q[g * (size_t)group + i] = (int8_t)nearbyintf(v); /* torch.round = half-to-even */
    }
  }
}

/* ======================================================================================
 * Model level: weight offsets + per-token forward
 * ==================================================================================== */
typedef struct {
  const float* w;
  const int8_t* w8;
  const float* scales;
  const float* bias; /* Qwen2 q/k/v only */
  int K, M;
} ko_linear;

struct ko_model {
  ko_config c;
  size_t expected_bytes;
  const float* tok_emb;
  const float** att_norm;
  const float** ffn_norm;
  const float* final_norm;
  ko_linear *wq, *wk, *wv, *wo, *w1, *w2, *w3;
  ko_linear cls;
  /* buffers (llama3.cpp:425-500) */
  float *x, *rms, *q, *attn, *mha_out, *w1o, *w3o, *w2o, *score, *kcache, *vcache, *logits;
  float *sin_cache, *cos_cache;
};

static void linear_fwd(const ko_model* m, const ko_linear* l, const float* x, float* y,
                       int acc) {
  if (m->c.is_quant && l->w8)
    ko_matmul_q8(x, l->w8, l->scales, m->c.group_size, y, l->M, l->K, acc);
  else
    ko_matmul_f32(x, l->w, y, l->M, l->K, 1.f, acc);
  /* matmul.cpp:74-77: bias added with the add kernel after the matmul */
  if (l->bias) ko_add_f32(y, l->bias, y, l->K);
}

const ko_config* ko_model_config(const ko_model* m) { return &m->c; }
size_t ko_model_expected_bytes(const ko_model* m) { return m->expected_bytes; }
const float* ko_model_logits(const ko_model* m) { return m->logits; }
float* ko_model_kcache(ko_model* m) { return m->kcache; }
float* ko_model_vcache(ko_model* m) { return m->vcache; }

void ko_model_destroy(ko_model* m) {
  if (!m) return;
  free(m->att_norm);
  free(m->ffn_norm);
  free(m->wq);
  free(m->wk);
  free(m->wv);
  free(m->wo);
  free(m->w1);
  free(m->w2);
  free(m->w3);
  free(m->x);
  free(m->rms);
  free(m->q);
  free(m->mha_out);
  free(m->w1o);
  free(m->w3o);
  free(m->w2o);
  free(m->score);
  free(m->kcache);
  free(m->vcache);
  free(m->logits);
  free(m->sin_cache);
  free(m->cos_cache);
  free(m);
}

ko_model* ko_model_create(const void* image, size_t nbytes, int family, int is_quant,
                          int rope_mode, float rope_theta, float rms_eps, int cache_len) {
  if (!image || nbytes < 28) return NULL;
  const int32_t* h = (const int32_t*)image;
  ko_model* m = (ko_model*)calloc(1, sizeof(ko_model));
  ko_config* c = &m->c;
  /* model.cpp:57-71 + generate_model_infos :125-151 */
  c->dim = h[0];
  c->hidden_dim = h[1];
  c->layer_num = h[2];
  c->head_num = h[3];
  c->kv_head_num = h[4];
  c->is_shared_weight = h[5] > 0;
  c->vocab_size = h[5] < 0 ? -h[5] : h[5];
  c->seq_len = h[6];
  c->is_quant = is_quant;
  c->group_size = is_quant ? h[7] : 0;
  if (c->dim <= 0 || c->head_num <= 0 || c->kv_head_num <= 0 || c->layer_num <= 0 ||
      c->hidden_dim <= 0 || c->vocab_size <= 0 || c->seq_len <= 0 ||
      (is_quant && (nbytes < 32 || c->group_size <= 0))) {
    free(m);
    return NULL;
  }
  c->kv_dim = (c->dim * c->kv_head_num) / c->head_num;
  c->kv_mul = c->head_num / c->kv_head_num;
  c->head_size = c->dim / c->head_num;
  c->family = family;
  c->rope_mode = rope_mode;
  c->rope_theta = rope_theta;
  c->rms_eps = rms_eps;
  c->cache_len = (cache_len > 0 && cache_len < c->seq_len) ? cache_len : c->seq_len;

  const int L = c->layer_num, dim = c->dim, kvd = c->kv_dim, hid = c->hidden_dim,
            V = c->vocab_size;
  m->att_norm = (const float**)calloc((size_t)L, sizeof(float*));
  m->ffn_norm = (const float**)calloc((size_t)L, sizeof(float*));
  m->wq = (ko_linear*)calloc((size_t)L, sizeof(ko_linear));
  m->wk = (ko_linear*)calloc((size_t)L, sizeof(ko_linear));
  m->wv = (ko_linear*)calloc((size_t)L, sizeof(ko_linear));
  m->wo = (ko_linear*)calloc((size_t)L, sizeof(ko_linear));
  m->w1 = (ko_linear*)calloc((size_t)L, sizeof(ko_linear));
  m->w2 = (ko_linear*)calloc((size_t)L, sizeof(ko_linear));
  m->w3 = (ko_linear*)calloc((size_t)L, sizeof(ko_linear));
  const int bias = (family == KO_FAMILY_QWEN2) && !is_quant;

  if (!is_quant) {
    /* llama3.cpp:290-423 / qwen2.cpp:290-426 ; writer tools/export.py:79-131 */
    const float* base = (const float*)((const char*)image + 28);
    size_t pos = 0;
    m->tok_emb = base;
    pos += (size_t)V * dim;
    for (int l = 0; l < L; ++l) m->att_norm[l] = base + pos + (size_t)l * dim;
    pos += (size_t)L * dim;
#define KO_TAKE(arr, KK, MM, HASB)                  \
  for (int l = 0; l < L; ++l) {                     \
    arr[l].w = base + pos;                          \
    arr[l].K = (KK);                                \
    arr[l].M = (MM);                                \
    pos += (size_t)(KK) * (size_t)(MM);             \
    if (HASB) {                                     \
      arr[l].bias = base + pos;                     \
      pos += (size_t)(KK);                          \
    }                                               \
  }
    KO_TAKE(m->wq, dim, dim, bias)
    KO_TAKE(m->wk, kvd, dim, bias)
    KO_TAKE(m->wv, kvd, dim, bias)
    KO_TAKE(m->wo, dim, dim, 0)
    for (int l = 0; l < L; ++l) m->ffn_norm[l] = base + pos + (size_t)l * dim;
    pos += (size_t)L * dim;
    KO_TAKE(m->w1, hid, dim, 0)
    KO_TAKE(m->w2, dim, hid, 0)
    KO_TAKE(m->w3, hid, dim, 0)
#undef KO_TAKE
    m->final_norm = base + pos;
    pos += (size_t)dim;
    pos += (size_t)c->seq_len * c->head_size; /* freqs_cos + freqs_sin, skipped (:367-368) */
    m->cls.K = V;
    m->cls.M = dim;
    if (c->is_shared_weight) {
      m->cls.w = m->tok_emb;
    } else {
      m->cls.w = base + pos;
      pos += (size_t)V * dim;
    }
    m->expected_bytes = 28 + pos * sizeof(float);
  } else {
    /* llama3.cpp:184-288 ; writer tools/export.py:134-210 (version 3) */
    if (c->is_shared_weight || family == KO_FAMILY_QWEN2) {
      /* reference points the int8 classifier at fp32 embedding bytes (llama3.cpp:259-262):
       * undefined behaviour there; refuse here. */
      ko_model_destroy(m);
      return NULL;
    }
    const int8_t* base = (const int8_t*)image + 32;
    const int gs = c->group_size;
    size_t pos = 0;
#define KO_TAKEQ(arr, KK, MM)                                        \
  for (int l = 0; l < L; ++l) {                                      \
    const size_t n = (size_t)(KK) * (size_t)(MM);                    \
    arr[l].w8 = base + pos;                                          \
    arr[l].scales = (const float*)(base + pos + n);                  \
    arr[l].K = (KK);                                                 \
    arr[l].M = (MM);                                                 \
    pos += n + (n / (size_t)gs) * sizeof(float);                     \
  }
    KO_TAKEQ(m->wq, dim, dim)
    KO_TAKEQ(m->wk, kvd, dim)
    KO_TAKEQ(m->wv, kvd, dim)
    KO_TAKEQ(m->wo, dim, dim)
    KO_TAKEQ(m->w1, hid, dim)
    KO_TAKEQ(m->w2, dim, hid)
    KO_TAKEQ(m->w3, hid, dim)
#undef KO_TAKEQ
    {
      const size_t n = (size_t)V * dim;
      m->cls.w8 = base + pos;
      m->cls.scales = (const float*)(base + pos + n);
      m->cls.K = V;
      m->cls.M = dim;
      pos += n + (n / (size_t)gs) * sizeof(float);
    }
    const float* fp = (const float*)(base + pos);
    m->tok_emb = fp;
    fp += (size_t)V * dim;
    for (int l = 0; l < L; ++l) m->att_norm[l] = fp + (size_t)l * dim;
    fp += (size_t)L * dim;
    for (int l = 0; l < L; ++l) m->ffn_norm[l] = fp + (size_t)l * dim;
    fp += (size_t)L * dim;
    m->final_norm = fp;
    fp += dim;
    m->expected_bytes = (size_t)((const char*)fp - (const char*)image);
  }
  if (m->expected_bytes > nbytes) {
    ko_model_destroy(m);
    return NULL;
  }

  const size_t CL = (size_t)c->cache_len;
  m->x = (float*)calloc((size_t)dim, sizeof(float));
  m->rms = (float*)calloc((size_t)dim, sizeof(float));
  m->q = (float*)calloc((size_t)dim, sizeof(float));
  m->mha_out = (float*)calloc((size_t)dim, sizeof(float));
  m->w2o = (float*)calloc((size_t)dim, sizeof(float));
  m->w1o = (float*)calloc((size_t)hid, sizeof(float));
  m->w3o = (float*)calloc((size_t)hid, sizeof(float));
  m->score = (float*)calloc((size_t)c->head_num * CL, sizeof(float));
  m->kcache = (float*)calloc((size_t)L * CL * kvd, sizeof(float));
  m->vcache = (float*)calloc((size_t)L * CL * kvd, sizeof(float));
  m->logits = (float*)calloc((size_t)V, sizeof(float));
  m->sin_cache = (float*)calloc(CL * c->head_size, sizeof(float));
  m->cos_cache = (float*)calloc(CL * c->head_size, sizeof(float));
  ko_sincos_cache(c->head_size, c->cache_len, rope_theta, m->sin_cache, m->cos_cache);
  return m;
}

/* LLama2Model::forward (llama3.cpp:147-167) with its helpers :600-731.
 * Buffer aliasing of the reference (rms/mha_out/w2o share storage; attn aliases q) does not
 * change values, so separate buffers are used for clarity. The KV "cache row" views of
 * slice_kv_cache (model.cpp:226-243) become direct pointers into kcache/vcache. */
int ko_model_forward(ko_model* m, int32_t token, int32_t pos, int acc) {
  const ko_config* c = &m->c;
  if (pos < 0 || pos >= c->cache_len || token < 0 || token >= c->vocab_size) return -1;
  const int dim = c->dim, kvd = c->kv_dim, CL = c->cache_len;
  memcpy(m->x, m->tok_emb + (size_t)token * dim, sizeof(float) * (size_t)dim);
  for (int l = 0; l < c->layer_num; ++l) {
    /* attention_rms (:600-609) */
    ko_rmsnorm_f32(m->x, m->att_norm[l], m->rms, dim, c->rms_eps);
    /* attention_qkv (:611-640) */
    float* krow = m->kcache + ((size_t)l * CL + (size_t)pos) * kvd;
    float* vrow = m->vcache + ((size_t)l * CL + (size_t)pos) * kvd;
    linear_fwd(m, &m->wq[l], m->rms, m->q, acc);
    linear_fwd(m, &m->wk[l], m->rms, krow, acc);
    linear_fwd(m, &m->wv[l], m->rms, vrow, acc);
    ko_rope_f32(dim, kvd, c->head_size, m->q, krow, pos, m->sin_cache, m->cos_cache,
                c->rope_mode);
    /* attention_mha (:652-676) */
    ko_mha_f32(pos, c->head_num, l, CL, kvd, c->kv_mul, c->head_size, m->mha_out, m->q,
               m->score, m->kcache, m->vcache, acc);
    linear_fwd(m, &m->wo[l], m->mha_out, m->q /* kAttnOutput aliases kQuery */, acc);
    /* feed_forward (:678-720) */
    ko_add_f32(m->x, m->q, m->x, dim);
    ko_rmsnorm_f32(m->x, m->ffn_norm[l], m->rms, dim, c->rms_eps);
    linear_fwd(m, &m->w1[l], m->rms, m->w1o, acc);
    linear_fwd(m, &m->w3[l], m->rms, m->w3o, acc);
    ko_swiglu_f32(m->w1o, m->w3o, m->w1o, c->hidden_dim);
    linear_fwd(m, &m->w2[l], m->w1o, m->w2o, acc);
    ko_add_f32(m->x, m->w2o, m->x, dim);
  }
  /* cls_logits (:722-731): final norm in place, then classifier */
  ko_rmsnorm_f32(m->x, m->final_norm, m->x, dim, c->rms_eps);
  linear_fwd(m, &m->cls, m->x, m->logits, acc);
  return 0;
}

/* demo/main.cpp:5-47 */
int ko_model_generate(ko_model* m, const int32_t* prompt, int n_prompt, int total_steps,
                      int32_t* out_words, int acc) {
  return ko_model_generate_until(m, prompt, n_prompt, total_steps, NULL, 0, out_words, acc);
}

int ko_model_generate_until(ko_model* m, const int32_t* prompt, int n_prompt, int total_steps,
                            const int32_t* stop, int n_stop, int32_t* out_words, int acc) {
  if (n_prompt <= 0) return -1;
  int pos = 0, nw = 0;
  int32_t next = -1;
  while (pos < total_steps) {
    const int is_prompt = pos < n_prompt - 1;
    if (is_prompt) {
      if (ko_model_forward(m, prompt[pos], pos, acc)) return -1;
    } else {
      /* at pos == n_prompt-1 the reference's `next` holds tokens[pos] (set by the previous
       * prompt iteration, or by main_qwen.cpp:12 for a 1-token prompt) */
      const int32_t tok = (pos == n_prompt - 1) ? prompt[n_prompt - 1] : next;
      if (ko_model_forward(m, tok, pos, acc)) return -1;
      next = (int32_t)ko_argmax_f32(m->logits, (size_t)m->c.vocab_size);
      /* demo/main.cpp:30-32 is_sentence_ending(next): in the prompt phase post_processing
       * returned -1 (llama3.cpp:736-737), so only a sampled token can end the loop */
      int hit = 0;
      for (int i = 0; i < n_stop; ++i) hit |= (stop[i] == next);
      if (hit) break;
    }
    if (is_prompt) next = prompt[pos + 1]; /* sampling skipped, next forced (:33-35) */
    out_words[nw++] = next;
    pos += 1;
  }
  return nw;
}
