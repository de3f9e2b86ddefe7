// kh_model.hip — model level of the C-ABI: .bin image -> HBM arena, per-token forward,
// greedy generate loop replayed as a hipGraph.  Replaces, for the decode path,
//   model::Model::read_model_file / generate_model_infos   kuiper/source/model/model.cpp:41-151
//   LLama2Model::create_param_layers / _quant_layers       kuiper/source/model/llama3.cpp:184-423
//   Qwen2Model::create_param_layers (q/k/v bias)           kuiper/source/model/qwen2.cpp:290-426
//   LLama2Model::init_mem / forward / predict              llama3.cpp:425-500, 147-167, 642-650
//   generate()                                             demo/main.cpp:5-47
// gfx950 only.  No CPU fallback: every path below launches HIP kernels.
#include <fcntl.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <new>
#include <thread>
#include <vector>

#include "kh_gemm.h"
#include "kh_pattn.h"
#include "kh_prefill.h"

namespace {

struct LayerW {
  KhLin wq, wk, wv, wo, w1, w2, w3;
  const float* att_norm;
  const float* ffn_norm;
};

}  // namespace

struct kh_model {
  kh_config cfg{};
  kh_model_opts opts{};
  hipStream_t stream = nullptr;
  // weights: one arena holding the .bin bytes after the header, in file order
  char* arena = nullptr;
  bool owns_arena = false;
  size_t arena_bytes = 0;
  std::vector<LayerW> layers;
  const float* tok_emb = nullptr;
  const float* final_norm = nullptr;
  KhLin cls{};
  int gshift = 0;
  // activations / caches (llama3.cpp:425-500)
  float *x = nullptr, *rms = nullptr, *q = nullptr, *att = nullptr, *h1 = nullptr,
        *h3 = nullptr, *w2o = nullptr, *logits = nullptr, *score = nullptr, *kcache = nullptr,
        *vcache = nullptr, *sin_cache = nullptr, *cos_cache = nullptr;
  float* part_val = nullptr;
  int32_t* part_idx = nullptr;
  int nparts = 0;
  float load_ms = 0.f;      // host image -> HBM upload time (kh_model_get_load_ms)
  void* attn_ws = nullptr;  // split-T attention partials + tickets (kh_attn.h)
  int attn_ns = 1;
  int attn_ns_g = 0;        // GQA long-context path: splits per KV group (0 = path off)
  int attn_ws_stride = 1;   // split slots per head in attn_ws
  int attn_t_long = 1 << 30;
  int attn_wg = KH_WG;
  int32_t *d_pos = nullptr, *d_token = nullptr, *d_next = nullptr, *d_forced = nullptr,
          *d_words = nullptr;
  int seq_cap = 0;  // capacity of d_forced / d_words
  // prefill (kh_prefill.h): residual / q / attention / hidden rows of up to KH_PF_BMAX prompt tokens
  float *pf_x = nullptr, *pf_q = nullptr, *pf_att = nullptr, *pf_h = nullptr;
  void* pf_ws = nullptr;        // KH_PF_BMAX attention split workspaces
  size_t pf_ws_tok_bytes = 0;
  // GEMM prefill (kh_gemm.h): slabs of KH_PG_TMAX token rows
  float *pg_x = nullptr, *pg_xn = nullptr, *pg_q = nullptr, *pg_att = nullptr, *pg_h = nullptr;
  void* pg_ws = nullptr;        // KH_PG_TMAX attention split workspaces
  size_t pg_ws_tok_bytes = 0;
  int32_t* h_words_pin = nullptr;  // pinned mirror of d_words (stop-token check)
  int pin_cap = 0;
  hipEvent_t ev_chunk[2] = {nullptr, nullptr};
  // launch geometry
  struct Shape {
    int u = 2, split = 1, grid = 1, wg = KH_WG;
  };
  Shape sh_qkv, sh_wo, sh_ffn, sh_w2, sh_cls;
  // graph
  // the decode step captured once as a 1-step graph and once as a KH_GRAPH_STEPS-step graph:
  // consecutive hipGraphLaunch calls leave the GPU idle for ~8 us (measured), so the long
  // graph amortises that gap over several tokens
  hipGraph_t graph = nullptr, graphN = nullptr;
  hipGraphExec_t gexec = nullptr, gexecN = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

namespace {

// Launch shape of one GEMV: rows are processed as `pairs` work items of two M-long rows.
//  split: waves sharing a pair (1/2/4) — raised while the launch has < 4096 waves and each wave
//         would still stream >= 8 KiB (fp32) / 4 KiB (int8);
//  u    : 16-byte loads per row per lane in flight (covers the wave's column range when it can);
//  grid : workgroups, <= 1024 (4 per CU), chosen so every wave gets the same number of items.
kh_model::Shape pick_shape(bool quant, int pairs, int M, int max_split, const char* env,
                           int wg = KH_WG, int wg_max = KH_WG, bool many_waves = false) {
  kh_model::Shape sh;
  sh.wg = wg;
  // tuning hook (tools/sweep_shapes.py): KH_SHAPE_<K>="split,u,grid[,wg]" overrides the heuristic
  if (const char* ov = env ? getenv(env) : nullptr) {
    int sp = 0, u = 0, g = 0, w = wg;
    const int nf = sscanf(ov, "%d,%d,%d,%d", &sp, &u, &g, &w);
    if (nf >= 3 && (sp == 1 || sp == 2 || sp == 4) && sp <= max_split &&
        (u == 2 || u == 4 || u == 8) && !(quant && u == 8) && g >= 1 && g <= 4096 &&
        (w == 256 || (w == 512 && wg_max >= 512))) {
      sh.split = sp;
      sh.u = u;
      sh.grid = g;
      sh.wg = w;
      return sh;
    }
  }
  const int elem = quant ? 1 : 4;
  const int min_bytes = quant ? 4096 : 8192;  // bytes one wave must still stream per pair
  const long pair_bytes = 2L * M * elem;
  // waves to aim for before rows are split.  With the rolling tile refill a wave that walks several
  // chunks of a long row keeps its loads in flight, so fewer, longer-lived waves beat many short
  // ones (r3 sweeps, profiles/r3_shape_sweep.md: Llama-2-7B w2 int8 split 4 -> 2: 12.6 -> 10.8 us,
  // fp32 33.6 -> 31.4; wo int8 split 2 -> 1: 5.6 -> 5.25); rounds 1-2 aimed at twice as many.
  const int target_waves = pair_bytes >= 16384 ? 4096 : 2048;
  while (sh.split < max_split && pairs * sh.split < target_waves &&
         pair_bytes / (sh.split * 2) >= min_bytes)
    sh.split *= 2;
  const int Mc = quant ? M / 16 : M / 4;
  const int per_lane = ((Mc + sh.split - 1) / sh.split + KH_WAVE - 1) / KH_WAVE;
  if (quant)
    // one chunk when it covers the column range; ranges that need several chunks anyway take the
    // small one (less padding in the last chunk, finer refill: w2 int8 u4 -> u2 11.9 -> 10.8 us)
    sh.u = per_lane > 4 ? 2 : (per_lane >= 3 ? 4 : 2);
  else
    sh.u = per_lane >= 8 ? 8 : (per_lane >= 3 ? 4 : 2);
  const int ppw = (wg / KH_WAVE) / sh.split;  // pairs per workgroup per iteration
  const int need = (pairs + ppw - 1) / ppw;
  // every workgroup re-stages the M-float input vector from L2: keep that below ~75 % of the
  // weight bytes (matters for w2, whose input is the hidden-sized vector; sweep in
  // profiles/r1_shape_sweep.md), and never more than 4 workgroups per CU
  long cap = (long)(0.75 * (double)pairs * (double)pair_bytes / ((double)M * 4.0));
  const long cap_hi = 1024L * KH_WG / wg, cap_lo = 256L * KH_WG / wg;  // 4 .. 1 x 256 threads / CU
  if (cap > cap_hi) cap = cap_hi;
  if (cap < cap_lo) cap = cap_lo;
  if (need <= cap) {
    sh.grid = need;
  } else {
    // several iterations per workgroup: keep the grid a whole number of workgroups per CU (256
    // CUs; 384- or 688-wide grids measured 5-10 % slower than their balanced neighbours) and
    // minimise the per-CU critical path (g/256)*ceil(need/g).  Ties: the many-row matrices
    // (qkv, ffn13, cls) prefer 8 resident waves per CU, then 12, 16 -- fewer, longer-lived
    // workgroups re-stage x less often; the few-long-row matrices (wo, w2: one or two chunks
    // per wave) prefer 16 -- everything is in flight at once (profiles/r1_shape_sweep.md).
    const int wpw = wg / KH_WAVE;           // waves per workgroup
    // resident waves per CU, in order of preference
    const int pref_lo[4] = {8, 12, 16, 4}, pref_hi[4] = {16, 12, 8, 4};
    long best_cost = -1;
    for (int wv : (many_waves ? pref_hi : pref_lo)) {
      if (wv == 4 && best_cost >= 0) break;  // 4 waves per CU only when nothing else fits
      if (wv % wpw) continue;
      const long g = 256L * (wv / wpw);
      if (g > cap) continue;
      const long cost = g * ((need + g - 1) / g);
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        sh.grid = (int)g;
      }
    }
    if (best_cost < 0) sh.grid = (int)cap;
  }
  return sh;
}

int ilog2_exact(int v) {
  if (v <= 0 || (v & (v - 1))) return -1;
  int s = 0;
  while ((1 << s) < v) ++s;
  return s;
}

// ---- fused launches -------------------------------------------------------------------------
// Template dispatch.  U: 16-byte loads per row in flight per lane; MV: in-register staging depth
// (kh_stage_maxv of the input length); SP: waves sharing one row pair.
// The workgroup size comes from a variable `kh_launch_wg` in scope at the dispatch site.
#define KH_L3(KERNEL, Q, UU, MV, GRID, LDS, STREAM, ARGS) \
  hipLaunchKernelGGL((KERNEL<Q, UU, MV>), dim3(GRID), dim3(kh_launch_wg), LDS, STREAM, ARGS)
#define KH_L4(KERNEL, Q, UU, MV, SP, GRID, LDS, STREAM, ARGS) \
  hipLaunchKernelGGL((KERNEL<Q, UU, MV, SP>), dim3(GRID), dim3(kh_launch_wg), LDS, STREAM, ARGS)
#define KH_SEL_MV3(KERNEL, Q, UU, MV, ...)                  \
  do {                                                      \
    if ((MV) == 4)                                          \
      KH_L3(KERNEL, Q, UU, 4, __VA_ARGS__);                 \
    else                                                    \
      KH_L3(KERNEL, Q, UU, 0, __VA_ARGS__);                 \
  } while (0)
#define KH_SEL_SP4(KERNEL, Q, UU, MV, SP, ...)              \
  do {                                                      \
    if ((SP) == 4)                                          \
      KH_L4(KERNEL, Q, UU, MV, 4, __VA_ARGS__);             \
    else if ((SP) == 2)                                     \
      KH_L4(KERNEL, Q, UU, MV, 2, __VA_ARGS__);             \
    else                                                    \
      KH_L4(KERNEL, Q, UU, MV, 1, __VA_ARGS__);             \
  } while (0)
#define KH_SEL_MV4(KERNEL, Q, UU, MV, SP, ...)              \
  do {                                                      \
    if ((MV) == 4)                                          \
      KH_SEL_SP4(KERNEL, Q, UU, 4, SP, __VA_ARGS__);        \
    else                                                    \
      KH_SEL_SP4(KERNEL, Q, UU, 0, SP, __VA_ARGS__);        \
  } while (0)
// k_gemv_res only: the 6-deep in-register staging for hidden-sized inputs (kh_stage_maxv)
#define KH_SEL_MV4X(KERNEL, Q, UU, MV, SP, ...)             \
  do {                                                      \
    if ((MV) == 6)                                          \
      KH_SEL_SP4(KERNEL, Q, UU, 6, SP, __VA_ARGS__);        \
    else                                                    \
      KH_SEL_MV4(KERNEL, Q, UU, MV, SP, __VA_ARGS__);       \
  } while (0)
#define KH_SEL_U(SEL, KERNEL, QUANT, U, ...)                \
  do {                                                      \
    if (QUANT) {                                            \
      if ((U) >= 4)                                         \
        SEL(KERNEL, true, 4, __VA_ARGS__);                  \
      else                                                  \
        SEL(KERNEL, true, 2, __VA_ARGS__);                  \
    } else {                                                \
      if ((U) >= 8)                                         \
        SEL(KERNEL, false, 8, __VA_ARGS__);                 \
      else if ((U) >= 4)                                    \
        SEL(KERNEL, false, 4, __VA_ARGS__);                 \
      else                                                  \
        SEL(KERNEL, false, 2, __VA_ARGS__);                 \
    }                                                       \
  } while (0)
// kernels without / with the SPLIT parameter
#define KH_DISPATCH3(KERNEL, QUANT, U, MV, GRID, LDS, STREAM, ARGS) \
  KH_SEL_U(KH_SEL_MV3, KERNEL, QUANT, U, MV, GRID, LDS, STREAM, ARGS)
#define KH_DISPATCH4(KERNEL, QUANT, U, MV, SP, GRID, LDS, STREAM, ARGS) \
  KH_SEL_U(KH_SEL_MV4, KERNEL, QUANT, U, MV, SP, GRID, LDS, STREAM, ARGS)
#define KH_DISPATCH4X(KERNEL, QUANT, U, MV, SP, GRID, LDS, STREAM, ARGS) \
  KH_SEL_U(KH_SEL_MV4X, KERNEL, QUANT, U, MV, SP, GRID, LDS, STREAM, ARGS)

KhQkvArgs fill_qkv(kh_model* m, int l) {
  const kh_config& c = m->cfg;
  const LayerW& W = m->layers[l];
  KhQkvArgs a;
  a.x = m->x;
  a.att_norm = W.att_norm;
  a.wq = W.wq;
  a.wk = W.wk;
  a.wv = W.wv;
  a.q_out = m->q;
  a.kcache_layer = m->kcache + (size_t)l * c.cache_len * c.kv_dim;
  a.vcache_layer = m->vcache + (size_t)l * c.cache_len * c.kv_dim;
  a.d_pos = m->d_pos;
  a.sin_cache = m->sin_cache;
  a.cos_cache = m->cos_cache;
  a.dim = c.dim;
  a.kv_dim = c.kv_dim;
  a.head_size = c.head_size;
  a.rope_mode = c.rope_mode;
  a.gshift = m->gshift;
  a.eps = c.rms_eps;
  return a;
}
void launch_qkv(kh_model* m, int l) {
  const kh_config& c = m->cfg;
  const KhQkvArgs a = fill_qkv(m, l);
  const bool qn = c.is_quant;
  const int kh_launch_wg = m->sh_qkv.wg;
  KH_DISPATCH4(k_qkv, qn, m->sh_qkv.u, kh_stage_maxv(c.dim, kh_launch_wg), m->sh_qkv.split, m->sh_qkv.grid,
               fused_lds_bytes(qn, c.dim), m->stream, a);
}
KhAttnArgs fill_attn(kh_model* m, int l) {
  const kh_config& c = m->cfg;
  KhAttnArgs a;
  a.q = m->q;
  a.kcache_layer = m->kcache + (size_t)l * c.cache_len * c.kv_dim;
  a.vcache_layer = m->vcache + (size_t)l * c.cache_len * c.kv_dim;
  a.out = m->att;
  a.d_pos = m->d_pos;
  a.kv_dim = c.kv_dim;
  a.kv_mul = c.kv_mul;
  a.head_size = c.head_size;
  a.kv_heads = c.kv_head_num;
  a.nsplit = m->attn_ns;
  a.ws = m->attn_ws;
  a.ws_stride = m->attn_ws_stride;
  a.nsplit_g = m->attn_ns_g;
  a.t_long = m->attn_t_long;
  a.tok_stride = 0;
  a.ws_tok_bytes = 0;
  return a;
}
int attn_group_lanes(const kh_config& c) {
  int G = 1;
  while (G < c.head_size / 4) G <<= 1;
  return G < 16 ? 16 : G;
}
void launch_attn(kh_model* m, int l) {
  const kh_config& c = m->cfg;
  const KhAttnArgs a = fill_attn(m, l);
  const int wg = m->attn_wg;
  if (c.head_size > 32)
    launch_attn_decode(a, 0, wg, m->stream);
  else  // head_size <= 32: generic LDS-score kernel (tiny test models)
    hipLaunchKernelGGL(k_attn_generic, dim3(c.head_num), dim3(wg),
                       attn_lds_bytes(c.head_size, wg), m->stream, a);
}
KhGemvResArgs fill_wo(kh_model* m, int l) {
  const kh_config& c = m->cfg;
  KhGemvResArgs a;
  a.vec = m->att;
  a.w = m->layers[l].wo;
  a.x = m->x;
  a.M = c.dim;
  a.K = c.dim;
  a.gshift = m->gshift;
  return a;
}
void launch_wo(kh_model* m, int l) {
  const kh_config& c = m->cfg;
  const KhGemvResArgs a = fill_wo(m, l);
  const bool qn = c.is_quant;
  const int kh_launch_wg = m->sh_wo.wg;
  KH_DISPATCH4(k_gemv_res, qn, m->sh_wo.u, kh_stage_maxv(c.dim, kh_launch_wg), m->sh_wo.split, m->sh_wo.grid,
               fused_lds_bytes(qn, c.dim), m->stream, a);
}
void launch_ffn13(kh_model* m, int l) {
  const kh_config& c = m->cfg;
  const LayerW& W = m->layers[l];
  KhFfn13Args a;
  a.x = m->x;
  a.ffn_norm = W.ffn_norm;
  a.w1 = W.w1;
  a.w3 = W.w3;
  a.h = m->h1;
  a.dim = c.dim;
  a.hidden = c.hidden_dim;
  a.gshift = m->gshift;
  a.eps = c.rms_eps;
  const bool qn = c.is_quant;
  const int kh_launch_wg = m->sh_ffn.wg;
  KH_DISPATCH3(k_ffn13, qn, m->sh_ffn.u, kh_stage_maxv(c.dim, kh_launch_wg), m->sh_ffn.grid,
               fused_lds_bytes(qn, c.dim), m->stream, a);
}
void launch_w2(kh_model* m, int l) {
  const kh_config& c = m->cfg;
  KhGemvResArgs a;
  a.vec = m->h1;
  a.w = m->layers[l].w2;
  a.x = m->x;
  a.M = c.hidden_dim;
  a.K = c.dim;
  a.gshift = m->gshift;
  const bool qn = c.is_quant;
  const int kh_launch_wg = m->sh_w2.wg;
  KH_DISPATCH4X(k_gemv_res, qn, m->sh_w2.u, kh_stage_maxv(c.hidden_dim, kh_launch_wg), m->sh_w2.split, m->sh_w2.grid,
                fused_lds_bytes(qn, c.hidden_dim), m->stream, a);
}
void launch_cls(kh_model* m) {
  const kh_config& c = m->cfg;
  KhClsArgs a;
  a.x = m->x;
  a.final_norm = m->final_norm;
  a.wcls = m->cls;
  a.logits = m->logits;
  a.part_val = m->part_val;
  a.part_idx = m->part_idx;
  a.dim = c.dim;
  a.vocab = c.vocab_size;
  a.gshift = m->gshift;
  a.eps = c.rms_eps;
  // the classifier is int8 only when the model is quantised (untied; llama3.cpp:255-268)
  const bool qn = c.is_quant;
  const int kh_launch_wg = m->sh_cls.wg;
  KH_DISPATCH3(k_cls, qn, m->sh_cls.u, kh_stage_maxv(c.dim, kh_launch_wg), m->sh_cls.grid, cls_lds_bytes(qn, c.dim),
               m->stream, a);
}
void launch_sample(kh_model* m, int advance, int n_forced) {
  const kh_config& c = m->cfg;
  KhSampleArgs a;
  a.part_val = m->part_val;
  a.part_idx = m->part_idx;
  a.nparts = m->nparts;
  a.forced = n_forced > 0 ? m->d_forced : nullptr;
  a.n_forced = n_forced;
  a.words = m->d_words;
  a.words_cap = m->seq_cap;
  a.d_next = m->d_next;
  a.d_token = m->d_token;
  a.d_pos = m->d_pos;
  a.tok_emb = m->tok_emb;
  a.x = m->x;
  a.dim = c.dim;
  a.vocab = c.vocab_size;
  a.advance = advance;
  hipLaunchKernelGGL(k_sample, dim3(1), dim3(KH_WG), 0, m->stream, a);
}

// one fused decode step = 5L + 2 launches.  ev (optional) receives an event after each launch.
void launch_step_fused(kh_model* m, int advance, int n_forced, hipEvent_t* ev) {
  int e = 0;
  auto mark = [&]() {
    if (ev) (void)hipEventRecord(ev[e++], m->stream);
  };
  mark();
  for (int l = 0; l < m->cfg.layer_num; ++l) {
    launch_qkv(m, l);
    mark();
    launch_attn(m, l);
    mark();
    launch_wo(m, l);
    mark();
    launch_ffn13(m, l);
    mark();
    launch_w2(m, l);
    mark();
  }
  launch_cls(m);
  mark();
  launch_sample(m, advance, n_forced);
  mark();
}

// the reference's own launch sequence, one C-ABI op per reference kernel (llama3.cpp:147-167)
int launch_step_unfused(kh_model* m, int pos) {
  const kh_config& c = m->cfg;
  void* s = (void*)m->stream;
  int rc;
#define KH_TRY(x)          \
  if ((rc = (x)) != KH_OK) \
  return rc
  auto lin = [&](const KhLin& L, const float* in, float* out, int M, int K) -> int {
    int r = c.is_quant ? kh_matmul_q8(in, (const int8_t*)L.w, L.scales, c.group_size, out, M, K, s)
                       : kh_matmul_f32(in, (const float*)L.w, out, M, K, 1.f, s);
    if (r == KH_OK && L.bias) r = kh_add_f32(out, L.bias, out, K, s);  // matmul.cpp:74-77
    return r;
  };
  for (int l = 0; l < c.layer_num; ++l) {
    const LayerW& W = m->layers[l];
    float* krow = m->kcache + ((size_t)l * c.cache_len + pos) * c.kv_dim;
    float* vrow = m->vcache + ((size_t)l * c.cache_len + pos) * c.kv_dim;
    KH_TRY(kh_rmsnorm_f32(m->x, W.att_norm, m->rms, c.dim, c.rms_eps, s));
    KH_TRY(lin(W.wq, m->rms, m->q, c.dim, c.dim));
    KH_TRY(lin(W.wk, m->rms, krow, c.dim, c.kv_dim));
    KH_TRY(lin(W.wv, m->rms, vrow, c.dim, c.kv_dim));
    KH_TRY(kh_rope_f32(c.dim, c.kv_dim, c.head_size, m->q, krow, nullptr, pos, m->sin_cache,
                       m->cos_cache, c.rope_mode, s));
    KH_TRY(kh_mha_f32(nullptr, pos, c.head_num, l, c.cache_len, c.kv_dim, c.kv_mul, c.head_size,
                      m->att, m->q, m->score, m->kcache, m->vcache, s));
    KH_TRY(lin(W.wo, m->att, m->q /* kAttnOutput aliases kQuery, llama3.cpp:478-489 */, c.dim,
               c.dim));
    KH_TRY(kh_add_f32(m->x, m->q, m->x, c.dim, s));
    KH_TRY(kh_rmsnorm_f32(m->x, W.ffn_norm, m->rms, c.dim, c.rms_eps, s));
    KH_TRY(lin(W.w1, m->rms, m->h1, c.dim, c.hidden_dim));
    KH_TRY(lin(W.w3, m->rms, m->h3, c.dim, c.hidden_dim));
    KH_TRY(kh_swiglu_f32(m->h1, m->h3, m->h1, c.hidden_dim, s));
    KH_TRY(lin(W.w2, m->h1, m->w2o, c.hidden_dim, c.dim));
    KH_TRY(kh_add_f32(m->x, m->w2o, m->x, c.dim, s));
  }
  KH_TRY(kh_rmsnorm_f32(m->x, m->final_norm, m->x, c.dim, c.rms_eps, s));
  KH_TRY(lin(m->cls, m->x, m->logits, c.dim, c.vocab_size));
  KH_TRY(kh_argmax_f32(m->logits, c.vocab_size, m->d_next, s));
#undef KH_TRY
  return KH_OK;
}

void set_state(kh_model* m, int token, int pos) {
  hipLaunchKernelGGL(k_set_state, dim3(1), dim3(KH_WG), 0, m->stream, token, pos, m->d_token,
                     m->d_pos, m->tok_emb, m->x, m->cfg.dim);
}

// Host image (typically the mmap of a .bin file: pageable, possibly not yet paged in) -> HBM.
// A plain hipMemcpy from pageable memory is staged by the driver in small pieces; here two
// pinned 64 MiB buffers are filled by a helper thread (page-in + memcpy) while the previous
// buffer is in flight on the copy engine, so disk/page-cache reads, the host memcpy and the
// PCIe transfer overlap.  Matters for the 27 GB fp32 7B image of the 8-replica config.
hipError_t upload_chunked(char* d_dst, const char* h_src, size_t n, hipStream_t stream,
                          float* ms_out) {
  const size_t CH = (size_t)64 << 20;
  const auto t0 = std::chrono::steady_clock::now();
  if (n <= CH) {
    hipError_t e = hipMemcpy(d_dst, h_src, n, hipMemcpyHostToDevice);
    if (ms_out)
      *ms_out = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return e;
  }
  char* pin[2] = {nullptr, nullptr};
  hipEvent_t done[2] = {nullptr, nullptr};
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; ++i) {
    e = hipHostMalloc((void**)&pin[i], CH, hipHostMallocDefault);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&done[i], hipEventDisableTiming);
  }
  if (e == hipSuccess) {
    const size_t nchunks = (n + CH - 1) / CH;
    auto fill = [&](size_t c) {  // helper-thread body: page-in + copy chunk c into its pinned buffer
      const size_t off = c * CH, len = off + CH <= n ? CH : n - off;
      memcpy(pin[c & 1], h_src + off, len);
    };
    std::thread filler(fill, (size_t)0);
    for (size_t c = 0; c < nchunks && e == hipSuccess; ++c) {
      filler.join();  // chunk c is staged
      const size_t off = c * CH, len = off + CH <= n ? CH : n - off;
      e = hipMemcpyAsync(d_dst + off, pin[c & 1], len, hipMemcpyHostToDevice, stream);
      if (e == hipSuccess) e = hipEventRecord(done[c & 1], stream);
      if (c + 1 < nchunks) {
        // the other buffer was last used by chunk c-1: wait for that transfer, then refill it
        if (c >= 1 && e == hipSuccess) e = hipEventSynchronize(done[(c + 1) & 1]);
        filler = std::thread(fill, c + 1);
      }
    }
    if (filler.joinable()) filler.join();
    if (e == hipSuccess) e = hipStreamSynchronize(stream);
  }
  for (int i = 0; i < 2; ++i) {
    if (done[i]) (void)hipEventDestroy(done[i]);
    if (pin[i]) (void)hipHostFree(pin[i]);
  }
  if (ms_out)
    *ms_out = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return e;
}

template <typename T>
int dalloc(T** p, size_t n) {
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  return e == hipSuccess ? KH_OK : (int)e;
}

int ensure_pinned_words(kh_model* m, int n) {
  for (auto& e : m->ev_chunk)
    if (!e && hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return (int)hipErrorUnknown;
  if (n <= m->pin_cap) return KH_OK;
  if (m->h_words_pin) (void)hipHostFree(m->h_words_pin);
  m->h_words_pin = nullptr;
  m->pin_cap = 0;
  if (hipHostMalloc((void**)&m->h_words_pin, sizeof(int32_t) * (size_t)n, hipHostMallocDefault) !=
      hipSuccess)
    return (int)hipErrorUnknown;
  m->pin_cap = n;
  return KH_OK;
}

int ensure_seq_cap(kh_model* m, int n) {
  if (n <= m->seq_cap) return KH_OK;
  if (m->d_forced) (void)hipFree(m->d_forced);
  if (m->d_words) (void)hipFree(m->d_words);
  m->d_forced = m->d_words = nullptr;
  m->seq_cap = 0;
  int rc;
  if ((rc = dalloc(&m->d_forced, (size_t)n + 1)) != KH_OK) return rc;
  if ((rc = dalloc(&m->d_words, (size_t)n + 1)) != KH_OK) return rc;
  // forced[i] = -1 (0xFFFFFFFF): every position sampled, until a generate uploads its prompt
  KH_CHECK_HIP(hipMemsetAsync(m->d_forced, 0xFF, sizeof(int32_t) * ((size_t)n + 1), m->stream));
  m->seq_cap = n;
  // the graph captured pointers/capacity: rebuild
  if (m->gexec) (void)hipGraphExecDestroy(m->gexec);
  if (m->gexecN) (void)hipGraphExecDestroy(m->gexecN);
  if (m->graph) (void)hipGraphDestroy(m->graph);
  if (m->graphN) (void)hipGraphDestroy(m->graphN);
  m->gexec = m->gexecN = nullptr;
  m->graph = m->graphN = nullptr;
  return KH_OK;
}

#define KH_GRAPH_STEPS 8
#define KH_PG_MIN_TOKENS 16  // prompts with fewer fed-only tokens stay on the bit-identical path
int capture_steps(kh_model* m, int n_forced, int steps, hipGraph_t* g, hipGraphExec_t* ge) {
  KH_CHECK_HIP(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < steps; ++i) launch_step_fused(m, /*advance=*/1, n_forced, nullptr);
  hipError_t e = hipStreamEndCapture(m->stream, g);
  if (e != hipSuccess) return (int)e;
  KH_CHECK_HIP(hipGraphInstantiate(ge, *g, nullptr, nullptr, 0));
  return KH_OK;
}
int ensure_graph(kh_model* m, int n_forced) {
  int rc;
  if (!m->gexec && (rc = capture_steps(m, n_forced, 1, &m->graph, &m->gexec)) != KH_OK) return rc;
  if (!m->gexecN &&
      (rc = capture_steps(m, n_forced, KH_GRAPH_STEPS, &m->graphN, &m->gexecN)) != KH_OK)
    return rc;
  return KH_OK;
}

// ---- weight table ------------------------------------------------------------------------
// Byte offsets are relative to the weight data (= file bytes after the header), mirroring
// kuiperllama_amd/binfmt.py::layout.
int build_weight_table(kh_model* m) {
  const kh_config& c = m->cfg;
  const int L = c.layer_num, dim = c.dim, kvd = c.kv_dim, hid = c.hidden_dim, V = c.vocab_size;
  m->layers.assign((size_t)L, LayerW{});
  char* base = m->arena;
  size_t off = 0;
  if (!c.is_quant) {
    const bool bias = c.family == KH_FAMILY_QWEN2;
    auto takef = [&](size_t n) {
      const float* p = (const float*)(base + off);
      off += n * sizeof(float);
      return p;
    };
    m->tok_emb = takef((size_t)V * dim);
    for (int l = 0; l < L; ++l) m->layers[l].att_norm = takef((size_t)dim);
    for (int l = 0; l < L; ++l) {
      m->layers[l].wq.w = takef((size_t)dim * dim);
      if (bias) m->layers[l].wq.bias = takef((size_t)dim);
    }
    for (int l = 0; l < L; ++l) {
      m->layers[l].wk.w = takef((size_t)kvd * dim);
      if (bias) m->layers[l].wk.bias = takef((size_t)kvd);
    }
    for (int l = 0; l < L; ++l) {
      m->layers[l].wv.w = takef((size_t)kvd * dim);
      if (bias) m->layers[l].wv.bias = takef((size_t)kvd);
    }
    for (int l = 0; l < L; ++l) m->layers[l].wo.w = takef((size_t)dim * dim);
    for (int l = 0; l < L; ++l) m->layers[l].ffn_norm = takef((size_t)dim);
    for (int l = 0; l < L; ++l) m->layers[l].w1.w = takef((size_t)hid * dim);
    for (int l = 0; l < L; ++l) m->layers[l].w2.w = takef((size_t)dim * hid);
    for (int l = 0; l < L; ++l) m->layers[l].w3.w = takef((size_t)hid * dim);
    m->final_norm = takef((size_t)dim);
    (void)takef((size_t)c.seq_len * c.head_size);  // freqs_cos + freqs_sin: skipped (:367-368)
    if (c.is_shared_weight) {
      m->cls.w = m->tok_emb;  // llama3.cpp:372-375
    } else {
      m->cls.w = takef((size_t)V * dim);
    }
  } else {
    const size_t gs = (size_t)c.group_size;
    auto takeq = [&](KhLin& Lw, size_t K, size_t M) {
      const size_t n = K * M;
      Lw.w = base + off;
      Lw.scales = (const float*)(base + off + n);  // layer.cpp:209-215
      off += n + (n / gs) * sizeof(float);
    };
    for (int l = 0; l < L; ++l) takeq(m->layers[l].wq, dim, dim);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].wk, kvd, dim);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].wv, kvd, dim);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].wo, dim, dim);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].w1, hid, dim);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].w2, dim, hid);
    for (int l = 0; l < L; ++l) takeq(m->layers[l].w3, hid, dim);
    takeq(m->cls, V, dim);
    const float* fp = (const float*)(base + off);
    m->tok_emb = fp;
    fp += (size_t)V * dim;
    for (int l = 0; l < L; ++l) m->layers[l].att_norm = fp + (size_t)l * dim;
    fp += (size_t)L * dim;
    for (int l = 0; l < L; ++l) m->layers[l].ffn_norm = fp + (size_t)l * dim;
    fp += (size_t)L * dim;
    m->final_norm = fp;
    fp += dim;
    off = (size_t)((const char*)fp - base);
  }
  if (off > m->arena_bytes) return KH_ERR_FORMAT;
  return KH_OK;
}

int parse_header(const int32_t* h, const kh_model_opts* o, kh_config* c) {
  // model.cpp:57-71, 125-151
  memset(c, 0, sizeof(*c));
  c->dim = h[0];
  c->hidden_dim = h[1];
  c->layer_num = h[2];
  c->head_num = h[3];
  c->kv_head_num = h[4];
  c->is_shared_weight = h[5] > 0;
  c->vocab_size = h[5] < 0 ? -h[5] : h[5];
  c->seq_len = h[6];
  c->is_quant = o->is_quant ? 1 : 0;
  c->group_size = o->is_quant ? h[7] : 0;
  if (c->dim <= 0 || c->hidden_dim <= 0 || c->layer_num <= 0 || c->head_num <= 0 ||
      c->kv_head_num <= 0 || c->vocab_size <= 0 || c->seq_len <= 0)
    return KH_ERR_FORMAT;
  if (c->dim % c->head_num || c->head_num % c->kv_head_num) return KH_ERR_FORMAT;
  c->kv_dim = (c->dim * c->kv_head_num) / c->head_num;
  c->kv_mul = c->head_num / c->kv_head_num;
  c->head_size = c->dim / c->head_num;
  c->family = o->family;
  c->rope_mode = o->rope_mode;
  c->rope_theta = o->rope_theta;
  c->rms_eps = o->rms_eps;
  c->cache_len = (o->max_seq_len > 0 && o->max_seq_len < c->seq_len) ? o->max_seq_len : c->seq_len;
  if (c->is_quant) {
    if (c->group_size <= 0) return KH_ERR_FORMAT;
    // the reference wires an int8 classifier onto fp32 embedding bytes when the classifier
    // is tied (llama3.cpp:259-262) and has no Qwen2 int8 bias layout: refuse both
    if (c->is_shared_weight || c->family == KH_FAMILY_QWEN2) return KH_ERR_UNSUPPORTED;
  }
  if (o->rope_mode != KH_ROPE_HALF && o->rope_mode != KH_ROPE_INTERLEAVED) return KH_ERR_INVALID_ARG;
  if (o->family != KH_FAMILY_LLAMA && o->family != KH_FAMILY_QWEN2) return KH_ERR_INVALID_ARG;
  if (!(o->rope_theta > 0.f) || !(o->rms_eps > 0.f)) return KH_ERR_INVALID_ARG;
  // vector-path preconditions of the fused kernels
  const int a = c->is_quant ? 16 : 4;
  if (c->dim % a || c->hidden_dim % a || c->head_size % 4 || (c->head_size & 1) ||
      c->head_size > 256 || c->kv_dim % 4 || (c->dim & 1) || (c->kv_dim & 1))
    return KH_ERR_UNSUPPORTED;
  if (c->is_quant) {
    const int gs = ilog2_exact(c->group_size);
    if (gs < 4 || c->dim % c->group_size || c->hidden_dim % c->group_size) return KH_ERR_UNSUPPORTED;
  }
  return KH_OK;
}

size_t expected_weight_bytes(const kh_config& c) {
  const size_t L = c.layer_num, dim = c.dim, kvd = c.kv_dim, hid = c.hidden_dim, V = c.vocab_size;
  const size_t lin = L * (2 * dim * dim + 2 * kvd * dim + 3 * hid * dim);
  if (!c.is_quant) {
    size_t n = V * dim + 2 * L * dim + lin + dim + (size_t)c.seq_len * c.head_size;
    if (c.family == KH_FAMILY_QWEN2) n += L * (dim + 2 * kvd);
    if (!c.is_shared_weight) n += V * dim;
    return n * sizeof(float);
  }
  const size_t q = lin + V * dim;
  return q + (q / (size_t)c.group_size) * sizeof(float) + (V * dim + 2 * L * dim + dim) * sizeof(float);
}

int finish_create(kh_model* m) {
  const kh_config& c = m->cfg;
  int rc;
  if ((rc = build_weight_table(m)) != KH_OK) return rc;
  m->gshift = c.is_quant ? ilog2_exact(c.group_size) : 0;
  const size_t CL = (size_t)c.cache_len;
#define KH_ALLOC(ptr, n) \
  if ((rc = dalloc(&(ptr), (n))) != KH_OK) return rc
  KH_ALLOC(m->x, (size_t)c.dim);
  KH_ALLOC(m->rms, (size_t)c.dim);
  KH_ALLOC(m->q, (size_t)c.dim);
  KH_ALLOC(m->att, (size_t)c.dim);
  KH_ALLOC(m->w2o, (size_t)c.dim);
  KH_ALLOC(m->h1, (size_t)c.hidden_dim);
  KH_ALLOC(m->h3, (size_t)c.hidden_dim);
  KH_ALLOC(m->logits, (size_t)c.vocab_size);
  KH_ALLOC(m->score, (size_t)c.head_num * CL);
  KH_ALLOC(m->kcache, (size_t)c.layer_num * CL * c.kv_dim);
  KH_ALLOC(m->vcache, (size_t)c.layer_num * CL * c.kv_dim);
  KH_ALLOC(m->sin_cache, CL * c.head_size);
  KH_ALLOC(m->cos_cache, CL * c.head_size);
  KH_ALLOC(m->d_pos, 1);
  KH_ALLOC(m->d_token, 1);
  KH_ALLOC(m->d_next, 1);
  // launch geometry
  m->sh_qkv = pick_shape(c.is_quant, (c.dim + 2 * c.kv_dim) / 2, c.dim, 2, "KH_SHAPE_QKV", KH_WG,
                         KH_WG_MAX);
  m->sh_wo = pick_shape(c.is_quant, c.dim / 2, c.dim, 4, "KH_SHAPE_WO", KH_WG, KH_WG_MAX, true);
  m->sh_ffn = pick_shape(c.is_quant, c.hidden_dim, c.dim, 1, "KH_SHAPE_FFN", KH_WG, KH_WG_MAX);
  // w2 re-stages the hidden-sized input in every workgroup: 512-thread workgroups halve that
  // L2 -> LDS traffic for the same number of waves (measured 14.1 -> 11.8 us on Llama-3.2-1B)
  m->sh_w2 = pick_shape(c.is_quant, c.dim / 2, c.hidden_dim, 4, "KH_SHAPE_W2", KH_WG_MAX, KH_WG_MAX,
                        true);
  m->sh_cls = pick_shape(c.is_quant, (c.vocab_size + 1) / 2, c.dim, 1, "KH_SHAPE_CLS",
                         c.is_quant ? KH_WG : KH_WG_MAX, KH_WG_MAX);
  m->nparts = m->sh_cls.grid;
  if (getenv("KH_SHAPE_DEBUG")) {
    const struct { const char* n; const kh_model::Shape* s; } all[] = {
        {"qkv", &m->sh_qkv}, {"wo", &m->sh_wo}, {"ffn13", &m->sh_ffn}, {"w2", &m->sh_w2}, {"cls", &m->sh_cls}};
    for (const auto& e : all)
      fprintf(stderr, "[kh] shape %-5s split %d u %d grid %d wg %d\n", e.n, e.s->split, e.s->u, e.s->grid, e.s->wg);
  }
  m->attn_ns = c.head_size > 32 ? attn_num_splits(c.cache_len) : 1;
  // attention: 8 waves per (head, split) shorten each lane's timestep loop
  m->attn_wg = KH_WG_MAX;
  if (const char* e = getenv("KH_ATTN_WG"))
    if (atoi(e) == 256 || atoi(e) == 512) m->attn_wg = atoi(e);
  // GQA long-context path (kh_attn.h): one workgroup per (kv group, split) from pos + 1 >=
  // t_long on; KH_ATTN_TLONG overrides the threshold (0 = never)
  m->attn_ws_stride = m->attn_ns;
  if (c.kv_mul > 1 && c.head_size > 32 &&
      attn_group_supported(c.head_size, c.kv_mul, m->attn_wg)) {
    // default policy: models with few KV heads (Qwen2.5-0.5B: 2) cannot fill the chip with
    // (group, split) workgroups and stay per-head; an explicit KH_ATTN_TLONG overrides
    int t_long = c.kv_head_num >= KH_ATTN_MIN_GROUPS ? KH_ATTN_TLONG_DEFAULT : 0;
    if (const char* e = getenv("KH_ATTN_TLONG")) t_long = atoi(e);
    if (t_long > 0 && t_long <= (int)c.cache_len) {
      m->attn_ns_g = attn_group_splits(c.cache_len, c.kv_head_num);
      m->attn_t_long = t_long;
      if (m->attn_ns_g > m->attn_ws_stride) m->attn_ws_stride = m->attn_ns_g;
    }
  }
  if (const size_t wsb = attn_ws_bytes(c.head_num, c.head_size, m->attn_ws_stride)) {
    KH_CHECK_HIP(hipMalloc(&m->attn_ws, wsb));
    KH_CHECK_HIP(hipMemsetAsync(m->attn_ws, 0, wsb, m->stream));
  }
  KH_ALLOC(m->part_val, (size_t)m->nparts);
  KH_ALLOC(m->part_idx, (size_t)m->nparts);
#undef KH_ALLOC
  KH_CHECK_HIP(hipMemsetAsync(m->kcache, 0, sizeof(float) * c.layer_num * CL * c.kv_dim, m->stream));
  KH_CHECK_HIP(hipMemsetAsync(m->vcache, 0, sizeof(float) * c.layer_num * CL * c.kv_dim, m->stream));
  KH_CHECK_HIP(hipMemsetAsync(m->d_pos, 0, sizeof(int32_t), m->stream));
  KH_CHECK_HIP(hipMemsetAsync(m->d_token, 0, sizeof(int32_t), m->stream));
  // sin/cos table: computed on the host with libm exactly as the CPU backend does
  // (cpu/rope_kernel.cpp:4-16) so the fp32 table is bit-identical to the CPU reference's,
  // then uploaded once.  (kh_sincos_cache_f32 is the on-device twin of sin_cos_cache_calc_cu.)
  {
    const size_t n = CL * c.head_size;
    std::vector<float> hs_(n), hc_(n);
    std::vector<float> freq((size_t)c.head_size);
    for (int d = 0; d < c.head_size; ++d)
      freq[d] = 1.0f / powf(c.rope_theta, (float)d / (float)c.head_size);
    for (size_t p = 0; p < CL; ++p)
      for (int d = 0; d < c.head_size; ++d) {
        const float val = (float)p * freq[d];
        hs_[p * c.head_size + d] = sinf(val);
        hc_[p * c.head_size + d] = cosf(val);
      }
    KH_CHECK_HIP(hipMemcpy(m->sin_cache, hs_.data(), n * sizeof(float), hipMemcpyHostToDevice));
    KH_CHECK_HIP(hipMemcpy(m->cos_cache, hc_.data(), n * sizeof(float), hipMemcpyHostToDevice));
  }
  // big activation vectors (hidden > 16 K floats) need the >64 KiB dynamic-LDS opt-in
  const size_t lds_need = fused_lds_bytes(c.is_quant, c.hidden_dim);
  if (lds_need > 160 * 1024) return KH_ERR_UNSUPPORTED;
  if (lds_need > 64 * 1024) {
    const int v = (int)lds_need;
#define KH_ATTR(Q, UU, SP)                                                                     \
  (void)hipFuncSetAttribute((const void*)k_gemv_res<Q, UU, 0, SP>,                             \
                            hipFuncAttributeMaxDynamicSharedMemorySize, v);                    \
  (void)hipFuncSetAttribute((const void*)k_gemv_res<Q, UU, 6, SP>,                             \
                            hipFuncAttributeMaxDynamicSharedMemorySize, v)
    KH_ATTR(false, 8, 1); KH_ATTR(false, 8, 2); KH_ATTR(false, 8, 4);
    KH_ATTR(false, 4, 1); KH_ATTR(false, 4, 2); KH_ATTR(false, 4, 4);
    KH_ATTR(false, 2, 1); KH_ATTR(false, 2, 2); KH_ATTR(false, 2, 4);
    KH_ATTR(true, 4, 1); KH_ATTR(true, 4, 2); KH_ATTR(true, 4, 4);
    KH_ATTR(true, 2, 1); KH_ATTR(true, 2, 2); KH_ATTR(true, 2, 4);
#undef KH_ATTR
  }
  KH_CHECK_HIP(hipEventCreate(&m->ev0));
  KH_CHECK_HIP(hipEventCreate(&m->ev1));
  if ((rc = ensure_seq_cap(m, 256)) != KH_OK) return rc;
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  return KH_OK;
}

int new_model(const int32_t* h_header, const kh_model_opts* opts, kh_model** out) {
  if (!h_header || !opts || !out) return KH_ERR_INVALID_ARG;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return KH_ERR_NO_DEVICE;
  }
  if (opts->device < 0 || opts->device >= ndev) return KH_ERR_INVALID_ARG;
  kh_config cfg;
  int rc = parse_header(h_header, opts, &cfg);
  if (rc != KH_OK) return rc;
  KH_CHECK_HIP(hipSetDevice(opts->device));
  kh_model* m = new (std::nothrow) kh_model();
  if (!m) return (int)hipErrorOutOfMemory;
  m->cfg = cfg;
  m->opts = *opts;
  hipError_t e = hipStreamCreateWithFlags(&m->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete m;
    return (int)e;
  }
  *out = m;
  return KH_OK;
}

}  // namespace

// =============================================================================================
extern "C" void kh_model_destroy(kh_model* m) {
  if (!m) return;
  if (m->stream) (void)hipStreamSynchronize(m->stream);
  if (m->gexec) (void)hipGraphExecDestroy(m->gexec);
  if (m->gexecN) (void)hipGraphExecDestroy(m->gexecN);
  if (m->graph) (void)hipGraphDestroy(m->graph);
  if (m->graphN) (void)hipGraphDestroy(m->graphN);
  if (m->ev0) (void)hipEventDestroy(m->ev0);
  if (m->ev1) (void)hipEventDestroy(m->ev1);
  for (void* q : {(void*)m->pf_x, (void*)m->pf_q, (void*)m->pf_att, (void*)m->pf_h, m->pf_ws,
                  (void*)m->pg_x, (void*)m->pg_xn, (void*)m->pg_q, (void*)m->pg_att, (void*)m->pg_h,
                  m->pg_ws})
    if (q) (void)hipFree(q);
  for (auto e : m->ev_chunk)
    if (e) (void)hipEventDestroy(e);
  if (m->h_words_pin) (void)hipHostFree(m->h_words_pin);
  void* bufs[] = {m->x,      m->rms,    m->q,         m->att,       m->h1,       m->h3,
                  m->w2o,    m->logits, m->score,     m->kcache,    m->vcache,   m->sin_cache,
                  m->cos_cache, m->part_val, m->part_idx, m->d_pos, m->d_token,  m->d_next,
                  m->d_forced, m->d_words, m->attn_ws};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  if (m->owns_arena && m->arena) (void)hipFree(m->arena);
  if (m->stream) (void)hipStreamDestroy(m->stream);
  delete m;
}

extern "C" int kh_model_create_from_device_weights(const int32_t* h_header,
                                                   const void* d_weight_data,
                                                   size_t weight_nbytes,
                                                   const kh_model_opts* opts, kh_model** out) {
  if (!d_weight_data || !kh_aligned16(d_weight_data)) return KH_ERR_INVALID_ARG;
  if (!out) return KH_ERR_INVALID_ARG;
  *out = nullptr;
  kh_model* m = nullptr;
  int rc = new_model(h_header, opts, &m);
  if (rc != KH_OK) return rc;
  if (weight_nbytes < expected_weight_bytes(m->cfg)) {
    kh_model_destroy(m);
    *out = nullptr;
    return KH_ERR_FORMAT;
  }
  m->arena = (char*)const_cast<void*>(d_weight_data);
  m->owns_arena = false;
  m->arena_bytes = weight_nbytes;
  m->cfg.weight_bytes = (int64_t)expected_weight_bytes(m->cfg);
  rc = finish_create(m);
  if (rc != KH_OK) {
    kh_model_destroy(m);
    m = nullptr;
  }
  *out = m;
  return rc;
}

extern "C" int kh_model_create_from_host_image(const void* h_image, size_t nbytes,
                                               const kh_model_opts* opts, kh_model** out) {
  if (!h_image || !opts || !out) return KH_ERR_INVALID_ARG;
  const size_t hdr = opts->is_quant ? 32 : 28;
  if (nbytes < hdr) return KH_ERR_FORMAT;
  int32_t header[8] = {0};
  memcpy(header, h_image, hdr);
  *out = nullptr;
  kh_model* m = nullptr;
  int rc = new_model(header, opts, &m);
  if (rc != KH_OK) return rc;
  const size_t need = expected_weight_bytes(m->cfg);
  if (nbytes - hdr < need) {
    kh_model_destroy(m);
    *out = nullptr;
    return KH_ERR_FORMAT;
  }
  hipError_t e = hipMalloc((void**)&m->arena, need);
  if (e == hipSuccess) {
    m->owns_arena = true;
    m->arena_bytes = need;
    // weights go up once, in file order, into one arena (the reference cudaMallocs and copies
    // every tensor separately: tensor.cpp:104-119)
    e = upload_chunked(m->arena, (const char*)h_image + hdr, need, m->stream, &m->load_ms);
  }
  if (e != hipSuccess) {
    kh_model_destroy(m);
    *out = nullptr;
    return (int)e;
  }
  m->cfg.weight_bytes = (int64_t)need;
  rc = finish_create(m);
  if (rc != KH_OK) {
    kh_model_destroy(m);
    m = nullptr;
  }
  *out = m;
  return rc;
}

extern "C" int kh_model_create_from_file(const char* path, const kh_model_opts* opts,
                                         kh_model** out) {
  if (!path || !opts || !out) return KH_ERR_INVALID_ARG;
  // model.cpp:41-123: open + fstat + mmap(PROT_READ, MAP_PRIVATE)
  const int fd = open(path, O_RDONLY);
  if (fd == -1) return KH_ERR_IO;
  struct stat st;
  if (fstat(fd, &st) == -1 || st.st_size <= 0) {
    close(fd);
    return KH_ERR_IO;
  }
  void* data = mmap(nullptr, (size_t)st.st_size, PROT_READ, MAP_PRIVATE, fd, 0);
  if (data == MAP_FAILED || data == nullptr) {
    close(fd);
    return KH_ERR_IO;
  }
  const int rc = kh_model_create_from_host_image(data, (size_t)st.st_size, opts, out);
  munmap(data, (size_t)st.st_size);
  close(fd);
  return rc;
}

extern "C" int kh_model_get_config(const kh_model* m, kh_config* out) {
  if (!m || !out) return KH_ERR_INVALID_ARG;
  *out = m->cfg;
  out->launches_per_token = 5 * m->cfg.layer_num + 2;
  return KH_OK;
}
extern "C" float kh_model_get_load_ms(const kh_model* m) { return m ? m->load_ms : -1.f; }
extern "C" void* kh_model_stream(kh_model* m) { return m ? (void*)m->stream : nullptr; }

extern "C" int kh_model_get_logits(kh_model* m, float* h_logits) {
  if (!m || !h_logits) return KH_ERR_INVALID_ARG;
  KH_CHECK_HIP(hipMemcpyAsync(h_logits, m->logits, sizeof(float) * m->cfg.vocab_size,
                              hipMemcpyDeviceToHost, m->stream));
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  return KH_OK;
}
extern "C" int kh_model_get_kv(kh_model* m, float** d_kcache, float** d_vcache) {
  if (!m || !d_kcache || !d_vcache) return KH_ERR_INVALID_ARG;
  *d_kcache = m->kcache;
  *d_vcache = m->vcache;
  return KH_OK;
}

extern "C" int kh_model_read_kv(kh_model* m, int32_t layer, int32_t row0, int32_t nrows,
                                float* h_k, float* h_v) {
  if (!m || !h_k || !h_v || layer < 0 || layer >= m->cfg.layer_num || row0 < 0 || nrows <= 0 ||
      (int64_t)row0 + nrows > m->cfg.cache_len)
    return KH_ERR_INVALID_ARG;
  const size_t off = ((size_t)layer * m->cfg.cache_len + row0) * m->cfg.kv_dim;
  const size_t nb = (size_t)nrows * m->cfg.kv_dim * sizeof(float);
  KH_CHECK_HIP(hipMemcpyAsync(h_k, m->kcache + off, nb, hipMemcpyDeviceToHost, m->stream));
  KH_CHECK_HIP(hipMemcpyAsync(h_v, m->vcache + off, nb, hipMemcpyDeviceToHost, m->stream));
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  return KH_OK;
}

extern "C" int kh_model_write_kv(kh_model* m, int32_t layer, int32_t row0, int32_t nrows,
                                 const float* h_k, const float* h_v) {
  if (!m || !h_k || !h_v || layer < 0 || layer >= m->cfg.layer_num || row0 < 0 || nrows <= 0 ||
      (int64_t)row0 + nrows > m->cfg.cache_len)
    return KH_ERR_INVALID_ARG;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  const size_t off = ((size_t)layer * m->cfg.cache_len + row0) * m->cfg.kv_dim;
  const size_t nb = (size_t)nrows * m->cfg.kv_dim * sizeof(float);
  // hipMemcpyDefault: the source may be host memory or memory of this device (unified addressing)
  KH_CHECK_HIP(hipMemcpyAsync(m->kcache + off, h_k, nb, hipMemcpyDefault, m->stream));
  KH_CHECK_HIP(hipMemcpyAsync(m->vcache + off, h_v, nb, hipMemcpyDefault, m->stream));
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  return KH_OK;
}

extern "C" int kh_model_predict(kh_model* m, int32_t token, int32_t pos, int32_t is_prompt,
                                int32_t exec, int32_t* h_next) {
  if (!m || !h_next) return KH_ERR_INVALID_ARG;
  const kh_config& c = m->cfg;
  if (token < 0 || token >= c.vocab_size || pos < 0 || pos >= c.cache_len) return KH_ERR_RANGE;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  set_state(m, token, pos);  // embedding() + fill_input (llama3.cpp:578-598, model.cpp:245-263)
  int rc = KH_OK;
  if (exec == KH_EXEC_UNFUSED) {
    rc = launch_step_unfused(m, pos);
  } else if (exec == KH_EXEC_FUSED || exec == KH_EXEC_GRAPH) {
    launch_step_fused(m, /*advance=*/0, /*n_forced=*/0, nullptr);
    rc = kh_launch_status();
  } else {
    return KH_ERR_INVALID_ARG;
  }
  if (rc != KH_OK) return rc;
  int32_t next = -1;
  KH_CHECK_HIP(hipMemcpyAsync(&next, m->d_next, sizeof(int32_t), hipMemcpyDeviceToHost, m->stream));
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  *h_next = is_prompt ? -1 : next;  // post_processing (llama3.cpp:733-745)
  return KH_OK;
}

// ---- prompt prefill (kh_prefill.h) ---------------------------------------------------------------
namespace {
// tokens per pass of the dim-input matrices (qkv, wo, ffn13)
int prefill_batch(const kh_model* m) {
  const kh_config& c = m->cfg;
  if (!c.is_quant && pf_lds_bytes(false, c.dim, 8) <= 80 * 1024) return 8;
  return 4;
}
// The B-token kernels mirror the decode kernels' arithmetic only for the staging variant the
// decode path uses at these sizes (in-register, MAXV = 4) and for the fast attention core.
bool prefill_supported(const kh_model* m) {
  const kh_config& c = m->cfg;
  if (c.head_size <= 32) return false;
  if (kh_stage_maxv(c.dim, m->sh_qkv.wg) != 4 || kh_stage_maxv(c.dim, m->sh_ffn.wg) != 4) return false;
  if (m->sh_qkv.split > 2 || m->sh_ffn.split != 1) return false;
  if (pf_lds_bytes(c.is_quant, c.dim, 4) > 160 * 1024) return false;
  if (pf_lds_bytes(c.is_quant, c.hidden_dim, 2) > 160 * 1024) return false;
  return true;
}
int ensure_prefill_buffers(kh_model* m) {
  if (m->pf_x) return KH_OK;
  const kh_config& c = m->cfg;
  int rc;
  if ((rc = dalloc(&m->pf_x, (size_t)KH_PF_BMAX * c.dim)) != KH_OK) return rc;
  if ((rc = dalloc(&m->pf_q, (size_t)KH_PF_BMAX * c.dim)) != KH_OK) return rc;
  if ((rc = dalloc(&m->pf_att, (size_t)KH_PF_BMAX * c.dim)) != KH_OK) return rc;
  if ((rc = dalloc(&m->pf_h, (size_t)KH_PF_BMAX * c.hidden_dim)) != KH_OK) return rc;
  m->pf_ws_tok_bytes = (attn_ws_bytes(c.head_num, c.head_size, m->attn_ws_stride) + 255) & ~(size_t)255;
  if (m->pf_ws_tok_bytes) {
    KH_CHECK_HIP(hipMalloc(&m->pf_ws, m->pf_ws_tok_bytes * KH_PF_BMAX));
    KH_CHECK_HIP(hipMemsetAsync(m->pf_ws, 0, m->pf_ws_tok_bytes * KH_PF_BMAX, m->stream));
  }
  return KH_OK;
}
// The B-token kernels are register- and LDS-heavy: a grid larger than what is resident at once
// runs in rounds and every round re-stages the B activation vectors, so the decode shape's grid
// is clipped to one resident round (the result does not depend on the grid).
template <class K, class A>
void pf_launch(K kernel, int grid, int wg, size_t lds, hipStream_t s, const A& args) {
  // resident-workgroup count and the >64 KiB LDS opt-in are per (device, kernel, shape): a thread
  // that drives models on several GPUs must not reuse device A's answer (or skip the attribute)
  // on device B
  struct Cached {
    int dev;
    const void* fn;
    int wg;
    size_t lds;
    int resident;
  };
  static thread_local std::vector<Cached> cache;
  int dev = 0;
  (void)hipGetDevice(&dev);
  int resident = 0;
  for (const auto& c : cache)
    if (c.dev == dev && c.fn == (const void*)kernel && c.wg == wg && c.lds == lds) resident = c.resident;
  if (!resident) {
    if (lds > 64 * 1024)
      (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds);
    int per_cu = 0, cus = 256;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, wg, lds) != hipSuccess ||
        per_cu < 1)
      per_cu = 1;
    int v = 0;
    if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0)
      cus = v;
    resident = per_cu * cus;
    cache.push_back({dev, (const void*)kernel, wg, lds, resident});
  }
  if (grid > resident) grid = resident;
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(wg), lds, s, args);
}
template <bool Q, int B>
void pf_launch_gemv_res(kh_model* m, const kh_model::Shape& sh, const KhPfGemvResArgs& a) {
  const size_t lds = pf_lds_bytes(Q, a.M, B);
  if (sh.split == 4)
    pf_launch(k_pf_gemv_res<Q, 4, B>, sh.grid, sh.wg, lds, m->stream, a);
  else if (sh.split == 2)
    pf_launch(k_pf_gemv_res<Q, 2, B>, sh.grid, sh.wg, lds, m->stream, a);
  else
    pf_launch(k_pf_gemv_res<Q, 1, B>, sh.grid, sh.wg, lds, m->stream, a);
}
// y = W.v ; X += y for the nvalid tokens of the chunk, in sub-batches of the largest of 8/4/2
// tokens (<= bmax) whose input vectors fit LDS
void pf_gemv_res(kh_model* m, const kh_model::Shape& sh, const KhLin& w, const float* V, float* X,
                 int M, int K, int nvalid, int bmax) {
  const bool q = m->cfg.is_quant;
  int bs = bmax;
  while (bs > 2 && pf_lds_bytes(q, M, bs) > 160 * 1024) bs >>= 1;
  KhPfGemvResArgs a;
  a.w = w;
  a.M = M;
  a.K = K;
  a.gshift = m->gshift;
  for (int t0 = 0; t0 < nvalid; t0 += bs) {
    a.V = V + (size_t)t0 * M;
    a.X = X + (size_t)t0 * K;
    a.nvalid = nvalid - t0 < bs ? nvalid - t0 : bs;
    if (q) {
      if (bs >= 4) pf_launch_gemv_res<true, 4>(m, sh, a); else pf_launch_gemv_res<true, 2>(m, sh, a);
    } else {
      if (bs == 8) pf_launch_gemv_res<false, 8>(m, sh, a);
      else if (bs == 4) pf_launch_gemv_res<false, 4>(m, sh, a);
      else pf_launch_gemv_res<false, 2>(m, sh, a);
    }
  }
}
template <bool Q, int B>
void pf_launch_qkv(kh_model* m, const KhPfQkvArgs& a) {
  const size_t lds = pf_lds_bytes(Q, a.dim, B);
  if (m->sh_qkv.split == 2)
    pf_launch(k_pf_qkv<Q, 2, B>, m->sh_qkv.grid, m->sh_qkv.wg, lds, m->stream, a);
  else
    pf_launch(k_pf_qkv<Q, 1, B>, m->sh_qkv.grid, m->sh_qkv.wg, lds, m->stream, a);
}
template <bool Q, int B>
void pf_launch_ffn13(kh_model* m, const KhPfFfn13Args& a) {
  pf_launch(k_pf_ffn13<Q, B>, m->sh_ffn.grid, m->sh_ffn.wg, pf_lds_bytes(Q, a.dim, B), m->stream, a);
}
// forward of nvalid (<= B) prompt tokens at positions pos0.. : fills their K/V cache rows
void launch_prefill_chunk(kh_model* m, const int32_t* toks, int nvalid, int pos0, int B) {
  const kh_config& c = m->cfg;
  const bool q = c.is_quant;
  KhPfTokens tk;
  for (int b = 0; b < KH_PF_BMAX; ++b) tk.t[b] = toks[b < nvalid ? b : nvalid - 1];
  hipLaunchKernelGGL(k_pf_embed, dim3(B), dim3(KH_WG), 0, m->stream, tk, m->tok_emb, m->pf_x, c.dim);
  for (int l = 0; l < c.layer_num; ++l) {
    const LayerW& W = m->layers[l];
    {
      KhPfQkvArgs a;
      a.X = m->pf_x;
      a.att_norm = W.att_norm;
      a.wq = W.wq;
      a.wk = W.wk;
      a.wv = W.wv;
      a.Q = m->pf_q;
      a.kcache_layer = m->kcache + (size_t)l * c.cache_len * c.kv_dim;
      a.vcache_layer = m->vcache + (size_t)l * c.cache_len * c.kv_dim;
      a.sin_cache = m->sin_cache;
      a.cos_cache = m->cos_cache;
      a.dim = c.dim;
      a.kv_dim = c.kv_dim;
      a.head_size = c.head_size;
      a.rope_mode = c.rope_mode;
      a.gshift = m->gshift;
      a.pos0 = pos0;
      a.nvalid = nvalid;
      a.eps = c.rms_eps;
      if (q) pf_launch_qkv<true, 4>(m, a);
      else if (B == 8) pf_launch_qkv<false, 8>(m, a);
      else pf_launch_qkv<false, 4>(m, a);
    }
    {
      KhAttnArgs a = fill_attn(m, l);
      a.q = m->pf_q;
      a.out = m->pf_att;
      a.d_pos = nullptr;
      a.ws = m->pf_ws;
      a.tok_stride = c.dim;
      a.ws_tok_bytes = m->pf_ws_tok_bytes;
      launch_attn_decode(a, pos0, m->attn_wg, m->stream, nvalid, pos0 + nvalid - 1);
    }
    pf_gemv_res(m, m->sh_wo, W.wo, m->pf_att, m->pf_x, c.dim, c.dim, nvalid, B);
    {
      KhPfFfn13Args a;
      a.X = m->pf_x;
      a.ffn_norm = W.ffn_norm;
      a.w1 = W.w1;
      a.w3 = W.w3;
      a.H = m->pf_h;
      a.dim = c.dim;
      a.hidden = c.hidden_dim;
      a.gshift = m->gshift;
      a.nvalid = nvalid;
      a.eps = c.rms_eps;
      if (q) pf_launch_ffn13<true, 4>(m, a);
      else if (B == 8) pf_launch_ffn13<false, 8>(m, a);
      else pf_launch_ffn13<false, 4>(m, a);
    }
    pf_gemv_res(m, m->sh_w2, W.w2, m->pf_h, m->pf_x, c.hidden_dim, c.dim, nvalid, B);
  }
}
}  // namespace

// ---- GEMM prefill (kh_gemm.h) ------------------------------------------------------------------
namespace {
bool pg_supported(const kh_model* m) {
  const kh_config& c = m->cfg;
  if (c.head_size <= 32) return false;  // attention: the fast multi-token decode kernel
  const int kq = c.is_quant ? 64 : 16;  // K granule of one MFMA operand load
  if (c.dim % kq || c.hidden_dim % kq || c.dim % 16 || c.kv_dim % 16 || c.hidden_dim % 16) return false;
  if (c.is_quant && m->gshift != 6) return false;
  return true;
}
int ensure_pg_buffers(kh_model* m) {
  if (m->pg_x) return KH_OK;
  const kh_config& c = m->cfg;
  const size_t T = KH_PG_TMAX;
  int rc;
  auto zalloc = [&](float** p, size_t n) -> int {
    if ((rc = dalloc(p, n)) != KH_OK) return rc;
    // rows beyond the valid tokens are read as MFMA operands (their columns are discarded):
    // they must hold finite numbers
    return (int)hipMemsetAsync(*p, 0, n * sizeof(float), m->stream);
  };
  if ((rc = zalloc(&m->pg_x, T * c.dim)) != KH_OK) return rc;
  if ((rc = zalloc(&m->pg_xn, T * c.dim)) != KH_OK) return rc;
  if ((rc = zalloc(&m->pg_q, T * c.dim)) != KH_OK) return rc;
  if ((rc = zalloc(&m->pg_att, T * c.dim)) != KH_OK) return rc;
  if ((rc = zalloc(&m->pg_h, T * c.hidden_dim)) != KH_OK) return rc;
  m->pg_ws_tok_bytes = (attn_ws_bytes(c.head_num, c.head_size, m->attn_ws_stride) + 255) & ~(size_t)255;
  if (m->pg_ws_tok_bytes) {
    KH_CHECK_HIP(hipMalloc(&m->pg_ws, m->pg_ws_tok_bytes * T));
    KH_CHECK_HIP(hipMemsetAsync(m->pg_ws, 0, m->pg_ws_tok_bytes * T, m->stream));
  }
  return KH_OK;
}
// Launch shape of one prefill GEMM (kh_gemm.h): R 16-row tiles and NT 16-token tiles per wave,
// ks waves splitting K per workgroup, grid.y token slices.  Picked by a small cost model of the
// busiest SIMD, fitted to a sweep on Llama-3.2-1B (profiles/r2_gemm_shape_sweep.txt):
//   * a workgroup lives on ONE CU: fewer than 256 workgroups leave CUs idle (the model prices the
//     busiest CU, so such shapes simply show their long per-wave work);
//   * fp32: ONE wave per SIMD is the sweet spot: two MFMA-bound waves on a SIMD cost ~1.6x the
//     time of the same work in one wave (the (w1,w3) GEMM went 140 -> 84 us per launch from 8 to 4
//     waves per CU), three or more ~1.8x;  int8: the opposite - its waves spend VALU time on the
//     dequant between MFMAs, so a second and third wave per SIMD fill the matrix pipe
//     (profiles/r2_gemm_shape_sweep_7b.txt: (w1,w3) 2 -> 4 waves per workgroup 30.4 -> 23.3 ms per
//     prefill), up to what the register file admits;
//   * bigger register tiles need fewer operand bytes per MFMA (small factor), more token slices
//     re-read the weights from L2 (small factor), padding tokens are wasted MFMAs.
struct PgShape {
  int R, NT, ks, slices;
};
PgShape pg_shape(int T, int rows_total, bool r2_ok, int nm, int kblocks, int min_blocks, bool quant) {
  const int nt_all = (T + 15) / 16;
  static const int cand[4][3] = {{2, 8, 2}, {2, 4, 3}, {2, 2, 4}, {1, 4, 4}};  // R, NT, waves/SIMD that fit
  PgShape best{1, 4, 1, (nt_all + 3) / 4};
  double best_cost = -1.0;
  for (const auto& c : cand) {
    const int R = c[0], NT = c[1], occ = c[2];
    if (R == 2 && !r2_ok) continue;
    if (R == 1 && quant && r2_ok) continue;  // int8: the 32-row tile measured better wherever it fits
    if (NT > 4 && nt_all <= 4) continue;     // no 128-token tile for <= 64 tokens
    const int slices = (nt_all + NT - 1) / NT;
    const long wgs = (long)(rows_total / (16 * R)) * slices;
    for (int ks = 1; ks * nm * 64 <= KH_PG_WG_MAX(quant); ks *= 2) {
      if (ks > 1 && kblocks / ks < min_blocks) break;  // keep a useful K range per wave
      const long cu_waves = ((wgs + 255) / 256) * (long)(nm * ks);  // on the busiest CU
      long wps = (cu_waves + 3) / 4;                                // waves per SIMD there
      const long rounds = (wps + occ - 1) / occ;                    // beyond the register file: queued
      if (wps > occ) wps = occ;
      const double pen = quant ? (wps <= 1 ? 1.0 : (wps == 2 ? 0.72 : 0.62))
                               : (wps <= 1 ? 1.0 : (wps == 2 ? 1.6 : 1.8));
      const double per_wave = (double)((kblocks + ks - 1) / ks) * R * NT;
      double cost = per_wave * (double)(wps * rounds) * pen;
      cost *= 1.0 + 0.15 * (double)(R + NT) / (double)(R * NT);
      // every token slice re-reads the weights from L2 - and, int8, dequantises them again
      cost *= 1.0 + (quant ? 0.15 : 0.05) * (double)(slices - 1);
      cost *= (double)(slices * NT) / (double)nt_all;
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        best = PgShape{R, NT, ks, slices};
      }
    }
  }
  return best;
}
template <bool Q, int EPI>
void pg_launch_cfg(const PgShape& sh, int tiles, int wg, hipStream_t s, const KhPgGemmArgs& a) {
  const size_t lds = pg_lds_bytes(wg / 64, sh.NT);
  auto go = [&](auto kern) {
    if (lds > 48 * 1024)
      (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(kern, dim3(tiles, sh.slices), dim3(wg), lds, s, a);
  };
  if (sh.R == 2 && sh.NT == 8) go(k_pg_gemm<Q, 2, 8, EPI>);
  else if (sh.R == 2 && sh.NT == 4) go(k_pg_gemm<Q, 2, 4, EPI>);
  else if (sh.R == 2) go(k_pg_gemm<Q, 2, 2, EPI>);
  else go(k_pg_gemm<Q, 1, 4, EPI>);
}
// returns whether the QKV epilogue rotates q / k itself (else k_pg_rope has to follow)
template <int EPI>
bool pg_launch(kh_model* m, int rows_total, bool r2_ok, KhPgGemmArgs a) {
  const bool q = m->cfg.is_quant;
  const int nm = EPI == KH_PG_SWIGLU ? 2 : 1;
  PgShape sh = pg_shape(a.T, rows_total, r2_ok, nm, a.K / (q ? 64 : 16), q ? 4 : 16, q);
  {  // tuning hook: KH_PG_SHAPE_<QKV|RESID|SWIGLU>="R,NT,ks" overrides the heuristic
    static const char* const names[3] = {"KH_PG_SHAPE_QKV", "KH_PG_SHAPE_RESID", "KH_PG_SHAPE_SWIGLU"};
    static const char* const ov = getenv(names[EPI]);  // read once per process (one static per EPI)
    if (ov) {
      int R = 0, NT = 0, ks = 0;
      if (sscanf(ov, "%d,%d,%d", &R, &NT, &ks) == 3 && ((R == 2 && (NT == 2 || NT == 4 || NT == 8) && r2_ok) || (R == 1 && NT == 4)) &&
          (ks == 1 || ks == 2 || ks == 4 || ks == 8) && ks * nm * 64 <= KH_PG_WG_MAX(q))
        sh = PgShape{R, NT, ks, ((a.T + 15) / 16 + NT - 1) / NT};
    }
  }
  static const bool debug = getenv("KH_PG_DEBUG") != nullptr;
  if (debug)
    fprintf(stderr, "[pg] epi %d rows %d K %d T %d -> R %d NT %d slices %d ks %d (%d wgs x %d waves)\n", EPI,
            rows_total, a.K, a.T, sh.R, sh.NT, sh.slices, sh.ks, rows_total / (16 * sh.R) * sh.slices,
            nm * sh.ks);
  if (EPI == KH_PG_QKV && a.rope == KH_PG_ROPE_TILES && (sh.R != 2 || sh.NT > 4))
    a.rope = KH_PG_ROPE_OFF;  // no partner tile in the wave / no registers to hold it: k_pg_rope follows
  const int tiles = rows_total / (16 * sh.R);
  if (q) pg_launch_cfg<true, EPI>(sh, tiles, nm * sh.ks * 64, m->stream, a);
  else pg_launch_cfg<false, EPI>(sh, tiles, nm * sh.ks * 64, m->stream, a);
  return EPI == KH_PG_QKV && a.rope != KH_PG_ROPE_OFF;
}
// forward of T (<= KH_PG_TMAX) prompt tokens at positions pos0..: fills their K/V cache rows
void launch_prefill_gemm_chunk(kh_model* m, const int32_t* toks, int T, int pos0) {
  const kh_config& c = m->cfg;
  const bool q = c.is_quant;
  (void)kh_embedding_f32_host(toks, T, m->tok_emb, m->pg_x, c.dim, c.vocab_size, (void*)m->stream);
  auto rmsnorm = [&](const float* w) {
    if (q) hipLaunchKernelGGL(k_pg_rmsnorm<true>, dim3(T), dim3(KH_WG), 0, m->stream, m->pg_x, w, m->pg_xn, c.dim, c.rms_eps);
    else hipLaunchKernelGGL(k_pg_rmsnorm<false>, dim3(T), dim3(KH_WG), 0, m->stream, m->pg_x, w, m->pg_xn, c.dim, c.rms_eps);
  };
  // attention of the slice: MFMA kernel (kh_pattn.h) unless KH_PG_ATTN=0 or an odd head size
  static const bool attn_env = [] { const char* e = getenv("KH_PG_ATTN"); return !(e && e[0] == '0'); }();
  const bool mfma_attn = attn_env && pg_attn_supported(c.head_size);
  static const bool rope_fuse_env = [] { const char* e = getenv("KH_PG_ROPE_FUSE"); return !(e && e[0] == '0'); }();
  for (int l = 0; l < c.layer_num; ++l) {
    const LayerW& W = m->layers[l];
    float* kc = m->kcache + (size_t)l * c.cache_len * c.kv_dim;
    float* vc = m->vcache + (size_t)l * c.cache_len * c.kv_dim;
    rmsnorm(W.att_norm);
    bool rope_fused = false;
    {
      KhPgGemmArgs a{};
      a.w[0] = W.wq; a.w[1] = W.wk; a.w[2] = W.wv;
      a.B = m->pg_xn; a.b_tiled = 1; a.out = m->pg_q; a.kc = kc; a.vc = vc;
      a.rows0 = c.dim; a.rows1 = c.kv_dim; a.ldo = c.dim; a.K = c.dim; a.T = T; a.pos0 = pos0;
      a.gshift = m->gshift;
      // RoPE in the epilogue: interleaved pairs sit in one lane's float4; half-mode partners need the
      // paired-tile mapping (R = 2, head size a multiple of 32).  KH_PG_ROPE_FUSE=0: separate kernel.
      a.head_size = c.head_size; a.sin_cache = m->sin_cache; a.cos_cache = m->cos_cache;
      a.rope = !rope_fuse_env ? KH_PG_ROPE_OFF
               : (c.rope_mode == KH_ROPE_HALF ? (c.head_size % 32 == 0 ? KH_PG_ROPE_TILES : KH_PG_ROPE_OFF)
                                              : KH_PG_ROPE_PAIRS);
      rope_fused = pg_launch<KH_PG_QKV>(m, c.dim + 2 * c.kv_dim, c.dim % 32 == 0 && c.kv_dim % 32 == 0, a);
    }
    if (!rope_fused)
      hipLaunchKernelGGL(k_pg_rope, dim3(T), dim3(KH_WG), 0, m->stream, m->pg_q, kc, m->sin_cache,
                         m->cos_cache, c.dim, c.kv_dim, c.head_size, pos0, c.rope_mode);
    if (mfma_attn) {
      KhPgAttnArgs a{};
      a.q = m->pg_q; a.kc = kc; a.vc = vc; a.out = m->pg_att;
      a.dim = c.dim; a.kv_dim = c.kv_dim; a.kv_heads = c.kv_head_num; a.kv_mul = c.kv_mul;
      a.T = T; a.pos0 = pos0; a.layout = q ? KH_PA_TILED_Q8 : KH_PA_TILED_F32;
      launch_pg_attn(a, c.head_size, m->stream);
    } else {
      KhAttnArgs a = fill_attn(m, l);
      a.q = m->pg_q;
      a.out = m->pg_att;
      a.d_pos = nullptr;
      a.ws = m->pg_ws;
      a.tok_stride = c.dim;
      a.ws_tok_bytes = m->pg_ws_tok_bytes;
      // head sizes without an MFMA instantiation (and KH_PG_ATTN=0): the decode kernel, one grid
      // slice per token, 256-thread workgroups (4096 latency-bound (head, token) workgroups: twice
      // as many fit a CU as with the decode width, 28.7 -> 19.6 us per layer)
      launch_attn_decode(a, pos0, KH_WG, m->stream, T, pos0 + T - 1);
    }
    {
      KhPgGemmArgs a{};
      a.w[0] = W.wo;
      a.B = m->pg_att; a.b_tiled = mfma_attn ? 1 : 0; a.out = m->pg_x;  // decode kernel: row-major rows
      a.rows0 = c.dim; a.ldo = c.dim; a.K = c.dim; a.T = T; a.gshift = m->gshift;
      pg_launch<KH_PG_RESID>(m, c.dim, c.dim % 32 == 0, a);
    }
    rmsnorm(W.ffn_norm);
    {
      KhPgGemmArgs a{};
      a.w[0] = W.w1; a.w[1] = W.w3;
      a.B = m->pg_xn; a.b_tiled = 1; a.out = m->pg_h;
      a.rows0 = c.hidden_dim; a.ldo = c.hidden_dim; a.K = c.dim; a.T = T; a.gshift = m->gshift;
      pg_launch<KH_PG_SWIGLU>(m, c.hidden_dim, c.hidden_dim % 32 == 0, a);
    }
    {
      KhPgGemmArgs a{};
      a.w[0] = W.w2;
      a.B = m->pg_h; a.b_tiled = 1; a.out = m->pg_x;
      a.rows0 = c.dim; a.ldo = c.dim; a.K = c.hidden_dim; a.T = T; a.gshift = m->gshift;
      pg_launch<KH_PG_RESID>(m, c.dim, c.dim % 32 == 0, a);
    }
  }
}
}  // namespace

extern "C" int kh_model_prefill_gemm(kh_model* m, const int32_t* h_tokens, int32_t n, int32_t pos0) {
  if (!m || !h_tokens || n <= 0 || pos0 < 0) return KH_ERR_INVALID_ARG;
  const kh_config& c = m->cfg;
  if ((int64_t)pos0 + n > c.cache_len) return KH_ERR_RANGE;
  for (int i = 0; i < n; ++i)
    if (h_tokens[i] < 0 || h_tokens[i] >= c.vocab_size) return KH_ERR_RANGE;
  if (!pg_supported(m)) return KH_ERR_UNSUPPORTED;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  int rc;
  if ((rc = ensure_pg_buffers(m)) != KH_OK) return rc;
  for (int t0 = 0; t0 < n; t0 += KH_PG_TMAX)
    launch_prefill_gemm_chunk(m, h_tokens + t0, n - t0 < KH_PG_TMAX ? n - t0 : KH_PG_TMAX, pos0 + t0);
  return kh_launch_status();
}

// Operator-level entry of the MFMA slice attention (kh_pattn.h): the multi-token form of MHAKernel.
extern "C" int kh_mha_prefill_f32(int32_t pos0, int32_t n_tokens, int32_t head_num, int32_t layer_index,
                                  int32_t seq_len, int32_t kv_dim, int32_t kv_mul, int32_t head_size,
                                  float* mha_out, const float* q, const float* key_cache,
                                  const float* value_cache, void* stream) {
  if (!mha_out || !q || !key_cache || !value_cache) return KH_ERR_INVALID_ARG;
  if (n_tokens <= 0 || pos0 < 0 || head_num <= 0 || layer_index < 0 || seq_len <= 0 || kv_mul <= 0 ||
      head_size <= 0 || kv_dim <= 0)
    return KH_ERR_INVALID_ARG;
  if (head_num % kv_mul || kv_dim != (head_num / kv_mul) * head_size) return KH_ERR_INVALID_ARG;
  if ((int64_t)pos0 + n_tokens > seq_len) return KH_ERR_RANGE;
  if (!pg_attn_supported(head_size)) return KH_ERR_UNSUPPORTED;
  if (((uintptr_t)mha_out | (uintptr_t)q | (uintptr_t)key_cache | (uintptr_t)value_cache) & 15)
    return KH_ERR_INVALID_ARG;  // 16-byte loads / stores
  KhPgAttnArgs a{};
  const size_t layer_off = (size_t)layer_index * seq_len * kv_dim;
  a.q = q; a.kc = key_cache + layer_off; a.vc = value_cache + layer_off; a.out = mha_out;
  a.dim = head_num * head_size; a.kv_dim = kv_dim; a.kv_heads = head_num / kv_mul; a.kv_mul = kv_mul;
  a.T = n_tokens; a.pos0 = pos0; a.layout = KH_PA_ROWS;
  launch_pg_attn(a, head_size, (hipStream_t)stream);
  return kh_launch_status();
}

extern "C" int kh_model_prefill(kh_model* m, const int32_t* h_tokens, int32_t n, int32_t pos0) {
  if (!m || !h_tokens || n <= 0 || pos0 < 0) return KH_ERR_INVALID_ARG;
  const kh_config& c = m->cfg;
  if ((int64_t)pos0 + n > c.cache_len) return KH_ERR_RANGE;
  for (int i = 0; i < n; ++i)
    if (h_tokens[i] < 0 || h_tokens[i] >= c.vocab_size) return KH_ERR_RANGE;
  if (!prefill_supported(m)) return KH_ERR_UNSUPPORTED;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  int rc;
  if ((rc = ensure_prefill_buffers(m)) != KH_OK) return rc;
  const int B = prefill_batch(m);
  for (int t0 = 0; t0 < n; t0 += B)
    launch_prefill_chunk(m, h_tokens + t0, n - t0 < B ? n - t0 : B, pos0 + t0, B);
  return kh_launch_status();
}

// Time the prompt phase alone: n fed-only tokens at positions pos0.., HIP events on the model
// stream around exactly that work (no decode step, no set_state).  mode KH_PREFILL_TOKEN = the
// reference's one forward pass per prompt token (demo/main.cpp:20-22), replayed from the hipGraph;
// KH_PREFILL_GEMV = kh_model_prefill's B-token VALU kernels; KH_PREFILL_GEMM = the MFMA GEMM path.
extern "C" int kh_model_time_prefill(kh_model* m, const int32_t* h_tokens, int32_t n, int32_t pos0,
                                     int32_t mode, float* h_ms) {
  if (!m || !h_tokens || !h_ms || n <= 0 || pos0 < 0) return KH_ERR_INVALID_ARG;
  const kh_config& c = m->cfg;
  if ((int64_t)pos0 + n > c.cache_len) return KH_ERR_RANGE;
  for (int i = 0; i < n; ++i)
    if (h_tokens[i] < 0 || h_tokens[i] >= c.vocab_size) return KH_ERR_RANGE;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  int rc;
  if (mode == KH_PREFILL_TOKEN) {
    if ((rc = ensure_seq_cap(m, pos0 + n + 1)) != KH_OK) return rc;
    std::vector<int32_t> forced((size_t)m->seq_cap + 1, -1);
    for (int i = 0; i < n; ++i) forced[pos0 + i] = h_tokens[i];
    forced[pos0 + n] = h_tokens[n - 1];  // keeps the last timed step in the prompt phase
    KH_CHECK_HIP(hipMemcpyAsync(m->d_forced, forced.data(), forced.size() * sizeof(int32_t),
                                hipMemcpyHostToDevice, m->stream));
    KH_CHECK_HIP(hipStreamSynchronize(m->stream));
    const int n_forced = m->seq_cap + 1;
    if ((rc = ensure_graph(m, n_forced)) != KH_OK) return rc;
    set_state(m, h_tokens[0], pos0);
    KH_CHECK_HIP(hipEventRecord(m->ev0, m->stream));
    for (int s = 0; s < n;) {
      if (n - s >= KH_GRAPH_STEPS) {
        KH_CHECK_HIP(hipGraphLaunch(m->gexecN, m->stream));
        s += KH_GRAPH_STEPS;
      } else {
        KH_CHECK_HIP(hipGraphLaunch(m->gexec, m->stream));
        s += 1;
      }
    }
    KH_CHECK_HIP(hipEventRecord(m->ev1, m->stream));
  } else if (mode == KH_PREFILL_GEMV) {
    if (!prefill_supported(m)) return KH_ERR_UNSUPPORTED;
    if ((rc = ensure_prefill_buffers(m)) != KH_OK) return rc;
    KH_CHECK_HIP(hipEventRecord(m->ev0, m->stream));
    if ((rc = kh_model_prefill(m, h_tokens, n, pos0)) != KH_OK) return rc;
    KH_CHECK_HIP(hipEventRecord(m->ev1, m->stream));
  } else if (mode == KH_PREFILL_GEMM) {
    if (!pg_supported(m)) return KH_ERR_UNSUPPORTED;
    if ((rc = ensure_pg_buffers(m)) != KH_OK) return rc;
    KH_CHECK_HIP(hipEventRecord(m->ev0, m->stream));
    if ((rc = kh_model_prefill_gemm(m, h_tokens, n, pos0)) != KH_OK) return rc;
    KH_CHECK_HIP(hipEventRecord(m->ev1, m->stream));
  } else {
    return KH_ERR_INVALID_ARG;
  }
  KH_CHECK_HIP(hipEventSynchronize(m->ev1));
  KH_CHECK_HIP(hipEventElapsedTime(h_ms, m->ev0, m->ev1));
  if ((rc = kh_launch_status()) != KH_OK) return rc;
  return KH_OK;
}

extern "C" int kh_model_generate(kh_model* m, const int32_t* h_prompt, int32_t n_prompt,
                                 int32_t total_steps, int32_t exec, int32_t* h_words,
                                 int32_t* n_words, float* h_elapsed_ms) {
  return kh_model_generate_until(m, h_prompt, n_prompt, total_steps, exec, nullptr, 0, h_words,
                                 n_words, h_elapsed_ms);
}

namespace {
inline bool is_stop(int32_t t, const int32_t* stop, int n_stop) {
  for (int i = 0; i < n_stop; ++i)
    if (stop[i] == t) return true;
  return false;
}
}  // namespace

extern "C" int kh_model_generate_until(kh_model* m, const int32_t* h_prompt, int32_t n_prompt,
                                       int32_t total_steps, int32_t exec, const int32_t* h_stop,
                                       int32_t n_stop, int32_t* h_words, int32_t* n_words,
                                       float* h_elapsed_ms) {
  if (!m || !h_prompt || n_prompt <= 0 || total_steps <= 0 || !h_words || !n_words ||
      n_stop < 0 || (n_stop > 0 && !h_stop))
    return KH_ERR_INVALID_ARG;
  const kh_config& c = m->cfg;
  if (total_steps > c.cache_len) return KH_ERR_RANGE;
  for (int i = 0; i < n_prompt; ++i)
    if (h_prompt[i] < 0 || h_prompt[i] >= c.vocab_size) return KH_ERR_RANGE;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  int rc;
  *n_words = 0;

  if (exec == KH_EXEC_UNFUSED) {
    // the reference loop verbatim: host drives every step and reads `next` back each time
    KH_CHECK_HIP(hipEventRecord(m->ev0, m->stream));
    int pos = 0, next = -1, nw = 0;
    while (pos < total_steps) {
      const bool is_prompt = pos < n_prompt - 1;
      const int tok = pos <= n_prompt - 1 ? h_prompt[pos] : next;
      int got = -1;
      if ((rc = kh_model_predict(m, tok, pos, is_prompt, KH_EXEC_UNFUSED, &got)) != KH_OK) return rc;
      // demo/main.cpp:30-32: only a sampled token can end the sentence (next == -1 in the prompt)
      if (!is_prompt && is_stop(got, h_stop, n_stop)) break;
      next = is_prompt ? h_prompt[pos + 1] : got;
      h_words[nw++] = next;
      pos += 1;
    }
    KH_CHECK_HIP(hipEventRecord(m->ev1, m->stream));
    KH_CHECK_HIP(hipEventSynchronize(m->ev1));
    if (h_elapsed_ms) KH_CHECK_HIP(hipEventElapsedTime(h_elapsed_ms, m->ev0, m->ev1));
    *n_words = nw;
    return KH_OK;
  }
  if (exec != KH_EXEC_GRAPH && exec != KH_EXEC_FUSED) return KH_ERR_INVALID_ARG;

  if ((rc = ensure_seq_cap(m, total_steps)) != KH_OK) return rc;
  // forced[i] = token fed at position i while inside the prompt, -1 afterwards
  std::vector<int32_t> forced((size_t)m->seq_cap + 1, -1);
  for (int i = 0; i < n_prompt && i <= m->seq_cap; ++i) forced[i] = h_prompt[i];
  KH_CHECK_HIP(hipMemcpyAsync(m->d_forced, forced.data(), forced.size() * sizeof(int32_t),
                              hipMemcpyHostToDevice, m->stream));
  KH_CHECK_HIP(hipStreamSynchronize(m->stream));  // `forced` is a stack-lifetime staging buffer
  const int n_forced = m->seq_cap + 1;
  if (exec == KH_EXEC_GRAPH && (rc = ensure_graph(m, n_forced)) != KH_OK) return rc;

  // prompt phase: the tokens that are only fed (positions 0 .. n_prompt-2) go through the
  // B-tokens-per-weight-pass prefill when there are enough of them; it leaves exactly the K/V
  // rows the token-by-token steps would (bit for bit), and the loop below starts at the last
  // prompt token.  KH_PREFILL=0 keeps the reference's one-token-per-step prompt phase.
  int start = 0;
  KH_CHECK_HIP(hipEventRecord(m->ev0, m->stream));
  if (n_prompt - 1 >= 2 && n_prompt - 1 < total_steps) {
    // KH_PREFILL: "0" = the reference's token-by-token prompt phase, "gemv" = the bit-identical
    // B-token kernels, "gemm" = the MFMA GEMM path; default: GEMM from KH_PG_MIN_TOKENS fed-only
    // tokens on (fp32 tolerance), the B-token kernels below that
    const char* e = getenv("KH_PREFILL");
    const bool want_gemm = e ? !strcmp(e, "gemm") : n_prompt - 1 >= KH_PG_MIN_TOKENS;
    const bool want_gemv = e ? strcmp(e, "0") != 0 : true;
    if (want_gemm && pg_supported(m)) {
      if ((rc = kh_model_prefill_gemm(m, h_prompt, n_prompt - 1, 0)) != KH_OK) return rc;
      start = n_prompt - 1;
    } else if (want_gemv && prefill_supported(m)) {
      if ((rc = kh_model_prefill(m, h_prompt, n_prompt - 1, 0)) != KH_OK) return rc;
      start = n_prompt - 1;
    }
  }
  set_state(m, h_prompt[start], start);
  auto launch_chunk = [&](int s) -> int {  // enqueue the next 1 or KH_GRAPH_STEPS steps
    if (exec == KH_EXEC_GRAPH) {
      if (total_steps - s >= KH_GRAPH_STEPS) {
        if (hipGraphLaunch(m->gexecN, m->stream) != hipSuccess) return -1;
        return KH_GRAPH_STEPS;
      }
      if (hipGraphLaunch(m->gexec, m->stream) != hipSuccess) return -1;
      return 1;
    }
    launch_step_fused(m, 1, n_forced, nullptr);
    return 1;
  };
  int n_out = total_steps;
  if (n_stop == 0) {
    for (int s = start; s < total_steps;) {
      const int n = launch_chunk(s);
      if (n < 0) return (int)hipErrorUnknown;
      s += n;
    }
    KH_CHECK_HIP(hipEventRecord(m->ev1, m->stream));
    if ((rc = kh_launch_status()) != KH_OK) return rc;
    KH_CHECK_HIP(hipMemcpyAsync(h_words, m->d_words, sizeof(int32_t) * total_steps,
                                hipMemcpyDeviceToHost, m->stream));
    KH_CHECK_HIP(hipStreamSynchronize(m->stream));
    for (int i = 0; i < start; ++i) h_words[i] = h_prompt[i + 1];  // forced, main.cpp:36-38
  } else {
    // Stop-token check without a per-step host round trip: the words of every
    // chunk of steps are mirrored into pinned memory behind the chunk, and the host inspects
    // chunk k while chunk k+1 is already queued, so the GPU never waits for the check.  At
    // most two chunks of steps run past the stop token; their words are discarded.
    if ((rc = ensure_pinned_words(m, total_steps)) != KH_OK) return rc;
    struct Chunk { int s0, n; };
    Chunk infl[2];
    int n_infl = 0, head = 0, launched = start, stop_at = -1;
    for (int i = 0; i < start; ++i) m->h_words_pin[i] = h_prompt[i + 1];
    // Once chunks are queued, an early return must not leave graph launches and their D2H copies
    // into h_words_pin in flight (the caller may destroy the model or start another generate that
    // reallocates those buffers): every error path below drains the stream first.
    auto fail = [&](int code) -> int {
      (void)hipStreamSynchronize(m->stream);
      return code;
    };
#define KH_CHECK_DRAIN(expr)                          \
  do {                                                \
    hipError_t _e = (expr);                           \
    if (_e != hipSuccess) return fail((int)_e);       \
  } while (0)
    while (stop_at < 0 && (launched < total_steps || n_infl > 0)) {
      while (launched < total_steps && n_infl < 2) {
        const int n = launch_chunk(launched);
        if (n < 0) return fail((int)hipErrorUnknown);
        const int slot = (head + n_infl) & 1;
        KH_CHECK_DRAIN(hipMemcpyAsync(m->h_words_pin + launched, m->d_words + launched,
                                      sizeof(int32_t) * n, hipMemcpyDeviceToHost, m->stream));
        KH_CHECK_DRAIN(hipEventRecord(m->ev_chunk[slot], m->stream));
        infl[slot] = {launched, n};
        launched += n;
        ++n_infl;
      }
      KH_CHECK_DRAIN(hipEventSynchronize(m->ev_chunk[head]));
      const Chunk c0 = infl[head];
      for (int s = c0.s0; s < c0.s0 + c0.n; ++s)
        if (s >= n_prompt - 1 && is_stop(m->h_words_pin[s], h_stop, n_stop)) {
          stop_at = s;
          break;
        }
      if (stop_at >= 0) {
        // elapsed_ms ends behind the chunks already queued when the stop token was seen: it
        // includes up to 2 x 8 discarded steps past the stop (the reference's timer ends with the
        // step that produced it)
        KH_CHECK_DRAIN(hipEventRecord(m->ev1, m->stream));
      }
      head ^= 1;
      --n_infl;
    }
    if (stop_at < 0) KH_CHECK_DRAIN(hipEventRecord(m->ev1, m->stream));
    if ((rc = kh_launch_status()) != KH_OK) return fail(rc);
#undef KH_CHECK_DRAIN
    KH_CHECK_HIP(hipStreamSynchronize(m->stream));
    n_out = stop_at >= 0 ? stop_at : total_steps;
    memcpy(h_words, m->h_words_pin, sizeof(int32_t) * (size_t)n_out);
  }
  if (h_elapsed_ms) KH_CHECK_HIP(hipEventElapsedTime(h_elapsed_ms, m->ev0, m->ev1));
  *n_words = n_out;
  return KH_OK;
}

extern "C" int kh_model_profile_kernel(kh_model* m, int32_t kclass, int32_t pos, int32_t reps,
                                       float* h_avg_us) {
  if (!m || !h_avg_us || reps <= 0 || kclass < 0 || kclass >= KH_NUM_KCLASS)
    return KH_ERR_INVALID_ARG;
  const kh_config& c = m->cfg;
  if (pos < 0 || pos >= c.cache_len) return KH_ERR_RANGE;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  set_state(m, 1 % c.vocab_size, pos);
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
  if ((rc = ensure_seq_cap(m, pos + 1)) != KH_OK) return rc;
  if ((rc = ensure_graph(m, m->seq_cap + 1)) != KH_OK) return rc;
  for (int r = 0; r < reps; ++r) {
    set_state(m, 1 % m->cfg.vocab_size, pos);
    KH_CHECK_HIP(hipEventRecord(m->ev0, m->stream));
    KH_CHECK_HIP(hipGraphLaunch(m->gexec, m->stream));
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
  const int L = c.layer_num;
  const int nk = 5 * L + 2;
  std::vector<hipEvent_t> ev((size_t)nk + 1);
  for (auto& e : ev) KH_CHECK_HIP(hipEventCreate(&e));
  double acc[KH_NUM_KCLASS] = {0};
  int cnt[KH_NUM_KCLASS] = {0};
  set_state(m, 1 % c.vocab_size, start_pos);
  int rc = KH_OK;
  for (int s = 0; s < n_steps && rc == KH_OK; ++s) {
    launch_step_fused(m, 1, 0, ev.data());
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
