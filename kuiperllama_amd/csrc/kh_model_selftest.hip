// kh_model_selftest.hip — run-time self-checks of the two fast paths that are correct by test rather than by
// construction, executed once at the end of kh_model_create_* on the model's own device, weights and geometry:
//
//  (a) the int8 LDS-DMA ring kernels (kh_q8ring.h): their input-vector staging uses asm loads whose destination
//      registers the compiler believes written at the asm statement.  One launch of every adopted
//      ring kernel (ffn13 on the first layer's w1 / w3, the classifier) against the register-tile kernel of the same
//      launch on the same input, outputs compared word for word.  A mismatch turns the ring off for this model
//      (kh_config.ring_selftest = -1; the register-tile kernels are bit-identical by design, so nothing else changes).
//  (b) the fence-free in-launch merge of the decode-attention time splits (kh_attn.h::attn_publish_barrier, default
//      form): it leans on gfx950 behaviour the HIP memory model does not spell out.  One fenced launch gives the
//      reference, KH_SELFTEST_ATTN_LAUNCHES back-to-back fence-free launches of the same merge (per-head path and,
//      where the geometry has one, the GQA group path) must reproduce it word for word.  A mismatch makes this
//      model use the fenced form (kh_config.attn_merge_selftest = -1).
//
// Both run on scratch state the model owns at that moment (activation buffers, rows of layer 0 of the still
// empty KV cache, which are zeroed again) in about a millisecond.  This is a tripwire for a part, a compiler or a
// partition mode on which the suite never ran - a race that shows once in a million launches will not trip it.
// Hooks (kh_debug_set / KH_* environment at load): KH_SELFTEST=0 skips both; KH_SELFTEST_FAIL="ring", "attn" or
// "ring,attn" reports the named comparison as failed (fault injection: tests/test_model_gpu.py checks that the
// fallbacks engage and that decode still matches the oracle).
// Replaces nothing in the reference (it has neither path); cf. kuiper/source/op/kernels/cuda/matmul_kernel.cu:56-87,
// mha_kernel.cu:47-130 for the kernels these paths stand in for.
#include <chrono>
#include <string.h>

#include "kh_model_internal.h"

#define KH_SELFTEST_ATTN_LAUNCHES 24

namespace {

__global__ __launch_bounds__(KH_WG) void k_st_fill(float* p, size_t n, uint32_t seed, float scale) {
  for (size_t i = (size_t)blockIdx.x * KH_WG + threadIdx.x; i < n; i += (size_t)gridDim.x * KH_WG) {
    uint32_t h = (uint32_t)i * 2654435761u ^ seed;
    h ^= h >> 15;
    h *= 2246822519u;
    h ^= h >> 13;
    p[i] = scale * ((float)(h & 0xffffu) / 32768.0f - 1.0f);
  }
}
// *flag |= (a[i] != b[i] for some i), compared as words (NaN == NaN when the bits agree)
__global__ __launch_bounds__(KH_WG) void k_st_diff(const uint32_t* a, const uint32_t* b, size_t n, int32_t* flag) {
  bool d = false;
  for (size_t i = (size_t)blockIdx.x * KH_WG + threadIdx.x; i < n; i += (size_t)gridDim.x * KH_WG) d |= a[i] != b[i];
  if (d) *flag = 1;
}
inline int grid_for(size_t n) {
  const size_t g = (n + KH_WG - 1) / KH_WG;
  return (int)(g < 1 ? 1 : (g > 1024 ? 1024 : g));
}
inline bool hook_has(const char* v, const char* word) { return v && strstr(v, word) != nullptr; }

}  // namespace

namespace khm {

// *result: 0 not applicable, 1 passed, -1 failed (fallback engaged); the return value is a KH / HIP status
static int ring_selftest(kh_model* m, int32_t* d_flag, bool inject, int* result) {
  const kh_config& c = m->cfg;
  *result = 0;
  if (!m->ring.ffn_r && !m->ring.cls_r) return KH_OK;  // no ring kernel planned for this model
  const kh_model::RingPlan plan = m->ring;
  hipStream_t s = m->stream;
  int rc = KH_OK;
  set_state(m, 1 % c.vocab_size, 0);  // x = a real embedding row
  if (plan.ffn_r) {
    launch_ffn13(m, 0);  // ring -> h1
    KH_CHECK_HIP(hipMemcpyAsync(m->h3, m->h1, sizeof(float) * (size_t)c.hidden_dim, hipMemcpyDeviceToDevice, s));
    m->ring.ffn_r = 0;
    launch_ffn13(m, 0);  // register tiles -> h1
    m->ring = plan;
    hipLaunchKernelGGL(k_st_diff, dim3(grid_for((size_t)c.hidden_dim)), dim3(KH_WG), 0, s, (const uint32_t*)m->h1,
                       (const uint32_t*)m->h3, (size_t)c.hidden_dim, d_flag);
  }
  float* tmp_logits = nullptr;
  float* tmp_pv = nullptr;
  int32_t* tmp_pi = nullptr;
  if (plan.cls_r) {
    const size_t np = (size_t)(m->sh_cls.grid > plan.cls_grid ? m->sh_cls.grid : plan.cls_grid);
    if ((rc = dalloc(&tmp_logits, (size_t)c.vocab_size)) == KH_OK && (rc = dalloc(&tmp_pv, np)) == KH_OK &&
        (rc = dalloc(&tmp_pi, np)) == KH_OK) {
      launch_cls(m);  // ring -> logits
      float* const lg = m->logits;
      float* const pv = m->part_val;
      int32_t* const pi = m->part_idx;
      m->logits = tmp_logits;
      m->part_val = tmp_pv;
      m->part_idx = tmp_pi;
      m->ring.cls_r = 0;
      launch_cls(m);  // register tiles -> tmp
      m->ring = plan;
      m->logits = lg;
      m->part_val = pv;
      m->part_idx = pi;
      hipLaunchKernelGGL(k_st_diff, dim3(grid_for((size_t)c.vocab_size)), dim3(KH_WG), 0, s, (const uint32_t*)lg,
                         (const uint32_t*)tmp_logits, (size_t)c.vocab_size, d_flag);
    }
  }
  int32_t flag = 0;
  hipError_t e = hipMemcpyAsync(&flag, d_flag, sizeof(flag), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  for (void* q : {(void*)tmp_logits, (void*)tmp_pv, (void*)tmp_pi})
    if (q) (void)hipFree(q);
  if (rc != KH_OK) return rc;
  if (e != hipSuccess) return (int)e;
  if ((rc = kh_launch_status()) != KH_OK) return rc;
  *result = 1;
  if (!flag && !inject) return KH_OK;
  // fall back: register-tile kernels for this model; the argmax partials follow the register-tile grid
  *result = -1;
  m->ring = kh_model::RingPlan();
  if (m->sh_cls.grid != m->nparts) {
    (void)hipFree(m->part_val);
    (void)hipFree(m->part_idx);
    m->part_val = nullptr;
    m->part_idx = nullptr;
    m->nparts = m->sh_cls.grid;
    if ((rc = dalloc(&m->part_val, (size_t)m->nparts)) != KH_OK) return rc;
    if ((rc = dalloc(&m->part_idx, (size_t)m->nparts)) != KH_OK) return rc;
  }
  return KH_OK;
}

// one in-launch merge configuration: reference by the fenced form, then back-to-back fence-free launches
static int attn_merge_case(kh_model* m, int pos, int variant, int32_t* d_flag) {
  const kh_config& c = m->cfg;
  hipStream_t s = m->stream;
  set_state(m, 1 % c.vocab_size, pos);
  m->step_var = variant;
  m->attn_fenced = true;
  launch_attn(m, 0);
  KH_CHECK_HIP(hipMemcpyAsync(m->rms, m->att, sizeof(float) * (size_t)c.dim, hipMemcpyDeviceToDevice, s));
  m->attn_fenced = false;
  for (int i = 0; i < KH_SELFTEST_ATTN_LAUNCHES; ++i) {
    launch_attn(m, 0);
    hipLaunchKernelGGL(k_st_diff, dim3(grid_for((size_t)c.dim)), dim3(KH_WG), 0, s, (const uint32_t*)m->att,
                       (const uint32_t*)m->rms, (size_t)c.dim, d_flag);
  }
  return kh_launch_status();
}

// *result: 0 not applicable, 1 passed, -1 failed (fenced form engaged), 2 the fenced form was requested
static int attn_selftest(kh_model* m, int32_t* d_flag, bool inject, int* result) {
  const kh_config& c = m->cfg;
  *result = 2;
  if (m->attn_fenced) return KH_OK;  // the caller asked for the fenced form (flag / hook): nothing to decide
  *result = 0;
  if (c.head_size <= 32 || m->attn_ns <= 1) return KH_OK;  // generic kernel / no time splits in this cache
  // per-head path, merge in the launch (what a step takes when it cannot defer the merge to wo)
  int pos1 = (c.cache_len < 1024 ? c.cache_len : 1024) - 1;
  if (pos1 + 1 >= m->attn_t_long) pos1 = m->attn_t_long - 2;
  const bool head_case = pos1 >= 1 && attn_active_splits(pos1, m->attn_ns, m->attn_ts_shift) >= 2;
  // GQA group path (always merged by its last arriver)
  int pos2 = -1;
  if (m->attn_ns_g > 0 && m->attn_t_long < c.cache_len) {
    pos2 = m->attn_t_long + 255 < c.cache_len ? m->attn_t_long + 255 : c.cache_len - 1;
    if (attn_active_splits(pos2, m->attn_ns_g, KH_ATTN_TSG_SHIFT) < 2) pos2 = -1;
  }
  if (!head_case && pos2 < 0) return KH_OK;
  const size_t rows = (size_t)((head_case ? pos1 : 0) > pos2 ? (head_case ? pos1 : 0) : pos2) + 1;
  const size_t n = rows * (size_t)c.kv_dim;
  hipStream_t s = m->stream;
  int rc = kv_ensure(m, (int)rows, /*layer=*/0);  // layer 0 only: the rows the check borrows
  if (rc != KH_OK) return rc;
  hipLaunchKernelGGL(k_st_fill, dim3(grid_for(n)), dim3(KH_WG), 0, s, m->kcache, n, 0x9e3779b9u, 1.0f);
  hipLaunchKernelGGL(k_st_fill, dim3(grid_for(n)), dim3(KH_WG), 0, s, m->vcache, n, 0x85ebca6bu, 1.0f);
  hipLaunchKernelGGL(k_st_fill, dim3(grid_for((size_t)c.dim)), dim3(KH_WG), 0, s, m->q, (size_t)c.dim, 0xc2b2ae35u, 1.0f);
  if (head_case) rc = attn_merge_case(m, pos1, m->attn_ns_g > 0 ? 2 : 0, d_flag);
  if (rc == KH_OK && pos2 >= 0) rc = attn_merge_case(m, pos2, 0, d_flag);
  m->step_var = 0;
  m->attn_fenced = false;
  int32_t flag = 0;
  hipError_t e = hipMemsetAsync(m->kcache, 0, n * sizeof(float), s);  // the cache is empty again
  if (e == hipSuccess) e = hipMemsetAsync(m->vcache, 0, n * sizeof(float), s);
  if (e == hipSuccess) e = hipMemcpyAsync(&flag, d_flag, sizeof(flag), hipMemcpyDeviceToHost, s);
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (rc != KH_OK) return rc;
  if (e != hipSuccess) return (int)e;
  *result = 1;
  if (!flag && !inject) return KH_OK;
  *result = -1;
  m->attn_fenced = true;
  return KH_OK;
}

// -> KH_OK, or a HIP / KH error when a launch itself failed (the model is then not usable).  Results in
// m->cfg.ring_selftest / attn_merge_selftest: 0 not applicable (or skipped), 1 passed, -1 failed -> fallback
// engaged, 2 (attention only) the fenced form was requested.
int run_selftests(kh_model* m) {
  m->cfg.ring_selftest = 0;
  m->cfg.attn_merge_selftest = m->attn_fenced ? 2 : 0;
  if (dbg_off("KH_SELFTEST")) return KH_OK;
  const auto t0 = std::chrono::steady_clock::now();
  const char* inj = dbg("KH_SELFTEST_FAIL");
  int32_t* d_flag = nullptr;
  KH_CHECK_HIP(hipMalloc((void**)&d_flag, 2 * sizeof(int32_t)));
  int r = 0, a = 0;
  const hipError_t e = hipMemsetAsync(d_flag, 0, 2 * sizeof(int32_t), m->stream);
  int rc = e == hipSuccess ? ring_selftest(m, d_flag, hook_has(inj, "ring"), &r) : (int)e;
  if (rc == KH_OK) rc = attn_selftest(m, d_flag + 1, hook_has(inj, "attn"), &a);
  (void)hipFree(d_flag);
  if (rc != KH_OK) return rc;
  m->cfg.ring_selftest = r;
  m->cfg.attn_merge_selftest = a;
  if (dbg("KH_LOAD_DEBUG") || dbg("KH_SHAPE_DEBUG"))
    fprintf(stderr, "[kh] self-tests: ring %d, attention merge %d (%.2f ms)\n", r, a,
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
  return KH_OK;
}

}  // namespace khm
