// kh_model_profile.hip — timing entry points of the model level (bench.py, tools/): back-to-back
// per-kernel launches, single-step latency, per-kernel events inside a step.
#include <vector>

#include "kh_model_internal.h"

using namespace khm;

extern "C" int kh_model_profile_kernel(kh_model* m, int32_t kclass, int32_t pos, int32_t reps,
                                       float* h_avg_us) {
  if (!m || !h_avg_us || reps <= 0 || kclass < 0 || kclass >= KH_NUM_KCLASS)
    return KH_ERR_INVALID_ARG;
  const kh_config& c = m->cfg;
  if (pos < 0 || pos >= c.cache_len) return KH_ERR_RANGE;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  {
    const int rc = kv_ensure(m, pos + 1);
    if (rc != KH_OK) return rc;
  }
  set_state(m, 1 % c.vocab_size, pos);
  m->step_var = step_variant(m, pos, pos);  // the attention / wo pair a step at `pos` launches
  const bool per_layer = kclass < KH_K_CLS;
  const int n_inner = per_layer ? c.layer_num : 1;
  auto sweep = [&]() {
    for (int l = 0; l < n_inner; ++l) switch (kclass) {
        case KH_K_QKV: launch_qkv(m, l); break;
        case KH_K_ATTN: launch_attn(m, l); break;
        case KH_K_WO: launch_wo(m, l); break;
        case KH_K_FFN13: launch_ffn13(m, l); break;
        case KH_K_W2: launch_w2(m, l); break;
        case KH_K_CLS: launch_cls(m); break;
        default: launch_sample(m, /*advance=*/0, /*n_forced=*/0); break;
      }
  };
  // The sweeps are captured into a graph and replayed: a 3-4 us kernel finishes faster than the
  // host can enqueue the next one, so eager back-to-back launches would time the host.
  hipGraph_t g = nullptr;
  hipGraphExec_t ge = nullptr;
  KH_CHECK_HIP(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
  for (int r = 0; r < reps; ++r) sweep();
  hipError_t e = hipStreamEndCapture(m->stream, &g);
  if (e != hipSuccess) return (int)e;
  e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(g);
    return (int)e;
  }
  int rc = KH_OK;
  e = hipGraphLaunch(ge, m->stream);  // untimed: first-touch effects
  if (e == hipSuccess) e = hipEventRecord(m->ev0, m->stream);
  if (e == hipSuccess) e = hipGraphLaunch(ge, m->stream);
  if (e == hipSuccess) e = hipEventRecord(m->ev1, m->stream);
  if (e == hipSuccess) e = hipEventSynchronize(m->ev1);
  (void)hipGraphExecDestroy(ge);
  (void)hipGraphDestroy(g);
  if (e != hipSuccess) return (int)e;
  rc = kh_launch_status();
  if (rc != KH_OK) return rc;
  float ms = 0.f;
  KH_CHECK_HIP(hipEventElapsedTime(&ms, m->ev0, m->ev1));
  *h_avg_us = ms * 1e3f / (float)(reps * n_inner);
  return KH_OK;
}

extern "C" int kh_model_time_step(kh_model* m, int32_t pos, int32_t reps, float* h_us) {
  if (!m || !h_us || reps <= 0) return KH_ERR_INVALID_ARG;
  if (pos < 0 || pos >= m->cfg.cache_len) return KH_ERR_RANGE;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  // deep positions (long-context probes): grow the forced/words buffers to cover `pos` and
  // (re)capture the step graph if that replaced them
  int rc;
  if ((rc = kv_ensure(m, pos + 1)) != KH_OK) return rc;
  if ((rc = ensure_seq_cap(m, pos + 1)) != KH_OK) return rc;
  hipGraphExec_t ge = nullptr;
  if ((rc = step_graph(m, m->seq_cap + 1, step_variant(m, pos, pos), false, &ge)) != KH_OK) return rc;
  for (int r = 0; r < reps; ++r) {
    set_state(m, 1 % m->cfg.vocab_size, pos);
    KH_CHECK_HIP(hipEventRecord(m->ev0, m->stream));
    KH_CHECK_HIP(hipGraphLaunch(ge, m->stream));
    KH_CHECK_HIP(hipEventRecord(m->ev1, m->stream));
    KH_CHECK_HIP(hipEventSynchronize(m->ev1));
    float ms = 0.f;
    KH_CHECK_HIP(hipEventElapsedTime(&ms, m->ev0, m->ev1));
    h_us[r] = ms * 1e3f;
  }
  return KH_OK;
}

static const char* const kKClassNames[KH_NUM_KCLASS] = {"qkv", "attn", "wo", "ffn13", "w2", "cls",
                                                       "sample"};
extern "C" const char* kh_kclass_name(int k) {
  return (k >= 0 && k < KH_NUM_KCLASS) ? kKClassNames[k] : "?";
}

extern "C" int kh_model_profile_step(kh_model* m, int32_t start_pos, int32_t n_steps,
                                     float* h_avg_us, int32_t* h_count) {
  if (!m || !h_avg_us || !h_count || n_steps <= 0 || start_pos < 0) return KH_ERR_INVALID_ARG;
  const kh_config& c = m->cfg;
  if (start_pos + n_steps > c.cache_len) return KH_ERR_RANGE;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  {
    const int rc = kv_ensure(m, start_pos + n_steps);
    if (rc != KH_OK) return rc;
  }
  const int L = c.layer_num;
  const int nk = 5 * L + 2;
  std::vector<hipEvent_t> ev((size_t)nk + 1);
  for (auto& e : ev) KH_CHECK_HIP(hipEventCreate(&e));
  double acc[KH_NUM_KCLASS] = {0};
  int cnt[KH_NUM_KCLASS] = {0};
  set_state(m, 1 % c.vocab_size, start_pos);
  int rc = KH_OK;
  for (int s = 0; s < n_steps && rc == KH_OK; ++s) {
    launch_step_fused(m, 1, 0, ev.data(), step_variant(m, start_pos + s, start_pos + s));
    hipError_t e = hipStreamSynchronize(m->stream);
    if (e != hipSuccess) {
      rc = (int)e;
      break;
    }
    for (int k = 0; k < nk; ++k) {
      float ms = 0.f;
      (void)hipEventElapsedTime(&ms, ev[k], ev[k + 1]);
      int cls;
      if (k < 5 * L)
        cls = k % 5;  // qkv, attn, wo, ffn13, w2
      else
        cls = k == 5 * L ? KH_K_CLS : KH_K_SAMPLE;
      acc[cls] += (double)ms * 1000.0;
      cnt[cls] += 1;
    }
  }
  for (auto& e : ev) (void)hipEventDestroy(e);
  for (int k = 0; k < KH_NUM_KCLASS; ++k) {
    h_avg_us[k] = cnt[k] ? (float)(acc[k] / cnt[k]) : 0.f;
    h_count[k] = cnt[k] / n_steps;
  }
  return rc;
}
