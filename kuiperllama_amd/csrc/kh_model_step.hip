// kh_model_step.hip — the decode step of the model level: launch shapes, the fused (5L+2 launches)
// and unfused (the reference's own sequence) step, hipGraph capture, predict and the generate loop.
// Replaces, for the decode path,
//   LLama2Model::forward / predict                         kuiper/source/model/llama3.cpp:147-167, 642-650
//   generate()                                             demo/main.cpp:5-47
// gfx950 only.  No CPU fallback: every path below launches HIP kernels.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kh_fused.h"
#include "kh_fused_ring.h"
#include "kh_model_internal.h"

namespace khm {

// Launch shape of one GEMV: rows are processed as `pairs` work items of two M-long rows.
//  split: waves sharing a pair (1/2/4) — raised while the launch has < 4096 waves and each wave
//         would still stream >= 8 KiB (fp32) / 4 KiB (int8);
//  u    : 16-byte loads per row per lane in flight (covers the wave's column range when it can);
//  grid : workgroups, <= 1024 (4 per CU), chosen so every wave gets the same number of items.
kh_model::Shape pick_shape(bool quant, int pairs, int M, int max_split, const char* env, int wg,
                           int wg_max, bool many_waves, bool u3) {
  kh_model::Shape sh;
  sh.wg = wg;
  // tuning hook (tools/sweep_shapes.py): KH_SHAPE_<K>="split,u,grid[,wg]" overrides the heuristic
  if (const char* ov = env ? dbg(env) : nullptr) {
    int sp = 0, u = 0, g = 0, w = wg;
    const int nf = sscanf(ov, "%d,%d,%d,%d", &sp, &u, &g, &w);
    if (nf >= 3 && (sp == 1 || sp == 2 || sp == 4) && sp <= max_split &&
        (u == 2 || u == 4 || u == 8 || (quant && u3 && u == 3)) && !(quant && u == 8) && g >= 1 && g <= 4096 &&
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
  // waves to aim for before rows are split.  gemv_pairs walks a wave's tiles (pair, chunk) in ONE loop and
  // requests the next tile in a burst right after the current tile's FMAs (kh_fused.h: ROLL = false everywhere but
  // the int8 QKV kernel), so a wave that walks several chunks of a long row keeps loads in flight across them and
  // fewer, longer-lived waves beat many short ones (r3 sweeps, profiles/r3_shape_sweep.txt: Llama-2-7B w2 int8
  // split 4 -> 2: 12.6 -> 10.8 us, fp32 33.6 -> 31.4; wo int8 split 2 -> 1: 5.6 -> 5.25); rounds 1-2 aimed at
  // twice as many.
  const int target_waves = pair_bytes >= 16384 ? 4096 : 2048;
  while (sh.split < max_split && pairs * sh.split < target_waves &&
         pair_bytes / (sh.split * 2) >= min_bytes)
    sh.split *= 2;
  const int Mc = quant ? M / 16 : M / 4;
  const int per_lane = ((Mc + sh.split - 1) / sh.split + KH_WAVE - 1) / KH_WAVE;
  if (quant) {
    // one chunk when it covers the column range; ranges that need several chunks anyway take the
    // small one (less padding in the last chunk, finer refill: w2 int8 u4 -> u2 11.9 -> 10.8 us)
    sh.u = per_lane > 4 ? 2 : (per_lane >= 3 ? 4 : 2);
    // [r6] w2 only (u3): a column range of 5 or 6 loads per lane as TWO exact tiles of 3 instead of three tiles of
    // 2 - Llama-2-7B int8's w2 (5.4 loads per lane): 10.45 -> 10.25 us, +0.4 % tok/s, four alternations on one box
    // (profiles/r6_w2_u3_ab.txt); ONE tile of 6 is slower (10.95 us: the whole item requested in one burst)
    if (u3 && per_lane > 4 && per_lane <= 6) sh.u = 3;
  } else {
    sh.u = per_lane >= 8 ? 8 : (per_lane >= 3 ? 4 : 2);
  }
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
    else if ((MV) == 2)                                     \
      KH_L3(KERNEL, Q, UU, 2, __VA_ARGS__);                 \
    else if ((MV) == 1)                                     \
      KH_L3(KERNEL, Q, UU, 1, __VA_ARGS__);                 \
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
    else if ((MV) == 2)                                     \
      KH_SEL_SP4(KERNEL, Q, UU, 2, SP, __VA_ARGS__);        \
    else if ((MV) == 1)                                     \
      KH_SEL_SP4(KERNEL, Q, UU, 1, SP, __VA_ARGS__);        \
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
  // step variants 1 and 2 cover positions below the group path's threshold only (step_variant): launch the
  // per-head-only instantiation, whose register count leaves room for two 512-thread workgroups per CU
  a.nsplit_g = m->step_var != 0 ? 0 : m->attn_ns_g;
  a.t_long = m->attn_t_long;
  a.ts_shift = m->attn_ts_shift;
  a.defer = m->step_var == 1 ? 1 : 0;
  a.fenced = m->attn_fenced ? 1 : 0;
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
// wo behind a deferring attention launch (step variant 1): in-register staging of 2 or 4 float4
#define KH_L5C(Q, UU, MV, SP, GRID, LDS, STREAM, ARGS) \
  hipLaunchKernelGGL((k_wo_comb<Q, UU, MV, SP>), dim3(GRID), dim3(kh_launch_wg), LDS, STREAM, ARGS)
#define KH_SEL_SP5C(Q, UU, MV, SP, ...)                     \
  do {                                                      \
    if ((SP) == 4)                                          \
      KH_L5C(Q, UU, MV, 4, __VA_ARGS__);                    \
    else if ((SP) == 2)                                     \
      KH_L5C(Q, UU, MV, 2, __VA_ARGS__);                    \
    else                                                    \
      KH_L5C(Q, UU, MV, 1, __VA_ARGS__);                    \
  } while (0)
#define KH_SEL_MV5C(KERNEL_UNUSED, Q, UU, MV, SP, ...)      \
  do {                                                      \
    if ((MV) == 2)                                          \
      KH_SEL_SP5C(Q, UU, 2, SP, __VA_ARGS__);               \
    else                                                    \
      KH_SEL_SP5C(Q, UU, 4, SP, __VA_ARGS__);               \
  } while (0)
void launch_wo(kh_model* m, int l) {
  const kh_config& c = m->cfg;
  const bool qn = c.is_quant;
  const int kh_launch_wg = m->sh_wo.wg;
  if (m->step_var == 1) {
    KhWoCombArgs a;
    a.g = fill_wo(m, l);
    const AttnSplitWs ws = attn_ws_carve(m->attn_ws, c.head_num, c.head_size, m->attn_ws_stride);
    a.cb.ml = ws.ml;
    a.cb.o = ws.o;
    a.cb.d_pos = m->d_pos;
    a.cb.ns = m->attn_ns;
    a.cb.ts_shift = m->attn_ts_shift;
    a.cb.nsw = m->attn_ws_stride;
    a.cb.heads = c.head_num;
    a.cb.hs = c.head_size;
    const int mv = c.dim <= 2 * 4 * kh_launch_wg ? 2 : 4;
    KH_SEL_U(KH_SEL_MV5C, k_wo_comb, qn, m->sh_wo.u, mv, m->sh_wo.split, m->sh_wo.grid,
             comb_lds_bytes(qn, c.dim, c.head_num), m->stream, a);
    return;
  }
  const KhGemvResArgs a = fill_wo(m, l);
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
  if (m->ring.ffn_r == 2) {  // plan_ring: int8, dim a multiple of 256 floats and at most 16 per thread
    hipLaunchKernelGGL((k_ffn13_ring<2, 4>), dim3(m->ring.ffn_grid), dim3(KH_WG),
                       ring_lds_bytes(c.dim, false, KH_WAVES_PER_WG, 2), m->stream, a);
    return;
  }
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
  const int mv = kh_stage_maxv(c.hidden_dim, kh_launch_wg);
  if (qn && m->sh_w2.u == 3) {  // two exact tiles of three loads per row (pick_shape, u3)
    KH_SEL_MV4X(k_gemv_res, true, 3, mv, m->sh_w2.split, m->sh_w2.grid, fused_lds_bytes(qn, c.hidden_dim), m->stream, a);
    return;
  }
  KH_DISPATCH4X(k_gemv_res, qn, m->sh_w2.u, mv, m->sh_w2.split, m->sh_w2.grid,
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
  if (m->ring.cls_r == 2) {
    hipLaunchKernelGGL((k_cls_ring<2, 4>), dim3(m->ring.cls_grid), dim3(KH_WG),
                       ring_lds_bytes(c.dim, false, KH_WAVES_PER_WG, 2), m->stream, a);
    return;
  }
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

// Which attention / wo pair the steps at positions pos_lo .. pos_hi launch (host decision, per captured
// graph or eager step).  Variant 1 (time splits merged by k_wo_comb) wherever the per-head path has more
// than one split to merge; variant 0 (the attention launch leaves the final vector) below position 256,
// where there is nothing to merge and wo keeps its plain staging, and from the first position of the GQA
// group path on, whose 32 splits per KV group are merged by their last arriver.  Every wo workgroup
// re-reads all nact partials (nact x dim floats from L2: 64 MB over the launch at 16 splits of a 2048-wide
// model, +3.3 us), so shapes whose staging cannot hide under wo's first weight tile (kh_fused.h:
// OVERLAP) defer only up to 4 splits (profiles/r4_attn_defer_ab.txt: Llama-2-7B int8 loses from 8 on).
int step_variant(const kh_model* m, int pos_lo, int pos_hi) {
  (void)pos_lo;
  if (pos_hi < KH_ATTN_MIN_TS) return 0;         // pos + 1 <= 256 everywhere: one split
  if (pos_hi + 1 >= m->attn_t_long) return 0;    // some step runs the group path
  if (m->attn_defer && attn_active_splits(pos_hi, m->attn_ns, m->attn_ts_shift) <= m->attn_defer_max) return 1;
  // Variant 2: the merge stays in the attention launch, but every step of the range is below the group path's
  // threshold, so the launch uses the per-head-only instantiation.  The one that also carries the group path needs
  // 138+ registers (two K/V batches in flight for kv_mul heads): one 512-thread workgroup per CU, which cost the 512
  // (head, split) workgroups at position 4094 +3.6 us per layer (profiles/r4_attn_pipe_ab.txt); per-head-only: 115.
  return m->attn_ns_g > 0 ? 2 : 0;
}
// one fused decode step = 5L + 2 launches.  ev (optional) receives an event after each launch.
void launch_step_fused(kh_model* m, int advance, int n_forced, hipEvent_t* ev, int variant) {
  m->step_var = variant;
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
  if (m->h_forced_pin) (void)hipHostFree(m->h_forced_pin);
  m->d_forced = m->d_words = nullptr;
  m->h_forced_pin = nullptr;
  m->seq_cap = 0;
  int rc;
  if (hipHostMalloc((void**)&m->h_forced_pin, sizeof(int32_t) * ((size_t)n + 1), hipHostMallocDefault) != hipSuccess)
    return (int)hipErrorUnknown;
  if ((rc = dalloc(&m->d_forced, (size_t)n + 1)) != KH_OK) return rc;
  if ((rc = dalloc(&m->d_words, (size_t)n + 1)) != KH_OK) return rc;
  // forced[i] = -1 (0xFFFFFFFF): every position sampled, until a generate uploads its prompt
  KH_CHECK_HIP(hipMemsetAsync(m->d_forced, 0xFF, sizeof(int32_t) * ((size_t)n + 1), m->stream));
  m->forced_hwm = 0;
  m->forced_in_flight = false;
  m->seq_cap = n;
  // the graph captured pointers/capacity: rebuild
  destroy_step_graphs(m);
  return KH_OK;
}
void destroy_step_graphs(kh_model* m) {
  for (int v = 0; v < KH_STEP_VARIANTS; ++v)
    for (int k = 0; k < 4; ++k) {
      kh_model::StepGraph& sg = m->sg[v][k];
      if (sg.e) (void)hipGraphExecDestroy(sg.e);
      if (sg.g) (void)hipGraphDestroy(sg.g);
      sg = kh_model::StepGraph{};
    }
}

int capture_steps(kh_model* m, int n_forced, int steps, int variant, hipGraph_t* g, hipGraphExec_t* ge) {
  KH_CHECK_HIP(hipStreamBeginCapture(m->stream, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < steps; ++i) launch_step_fused(m, /*advance=*/1, n_forced, nullptr, variant);
  hipError_t e = hipStreamEndCapture(m->stream, g);
  if (e != hipSuccess) return (int)e;
  KH_CHECK_HIP(hipGraphInstantiate(ge, *g, nullptr, nullptr, 0));
  return KH_OK;
}
int step_graph_n(kh_model* m, int n_forced, int variant, int nsteps, hipGraphExec_t* out) {
  if (variant < 0 || variant >= KH_STEP_VARIANTS) return KH_ERR_INVALID_ARG;
  const int k = nsteps == 1 ? 0 : nsteps == 2 ? 1 : nsteps == 4 ? 2 : nsteps == KH_GRAPH_STEPS ? 3 : -1;
  if (k < 0) return KH_ERR_INVALID_ARG;
  kh_model::StepGraph& sg = m->sg[variant][k];
  if (!sg.e) {
    const int rc = capture_steps(m, n_forced, nsteps, variant, &sg.g, &sg.e);
    if (rc != KH_OK) return rc;
    // push the executable graph to the device now: otherwise its FIRST launch pays for it (a 20-step run behind a
    // 5-step warm-up launched its 8-step graph for the first time inside the timed region: 1037-1046 tok/s by wall
    // clock where repeated runs gave 1060)
    (void)hipGraphUpload(sg.e, m->stream);
  }
  *out = sg.e;
  return KH_OK;
}
int step_graph(kh_model* m, int n_forced, int variant, bool steps8, hipGraphExec_t* out) {
  return step_graph_n(m, n_forced, variant, steps8 ? KH_GRAPH_STEPS : 1, out);
}

void plan_decode_shapes(bool quant, int dim, int hidden_dim, int kv_dim, int vocab_size,
                        kh_model::Shape (&out)[5]) {
  out[0] = pick_shape(quant, (dim + 2 * kv_dim) / 2, dim, 2, "KH_SHAPE_QKV", KH_WG, KH_WG_MAX);
  out[1] = pick_shape(quant, dim / 2, dim, 4, "KH_SHAPE_WO", KH_WG, KH_WG_MAX, true);
  out[2] = pick_shape(quant, hidden_dim, dim, 1, "KH_SHAPE_FFN", KH_WG, KH_WG_MAX);
  // w2 re-stages the hidden-sized input in every workgroup: 512-thread workgroups halve that
  // L2 -> LDS traffic for the same number of waves (measured 14.1 -> 11.8 us on Llama-3.2-1B)
  out[3] = pick_shape(quant, dim / 2, hidden_dim, 4, "KH_SHAPE_W2", KH_WG_MAX, KH_WG_MAX, true, /*u3=*/true);
  out[4] = pick_shape(quant, (vocab_size + 1) / 2, dim, 1, "KH_SHAPE_CLS", quant ? KH_WG : KH_WG_MAX, KH_WG_MAX);
}

// The int8 ffn13 and classifier launches run on the LDS-DMA ring kernels (kh_fused_ring.h) when the geometry fits
// them: at most 16 floats of the input vector per staging thread (the MAXV = 4 staging of the 256-thread kernels,
// which the ring kernels reproduce bit for bit), a power-of-two group of at least 16 weights (one scale per lane and
// piece).  Same-box A/B on Llama-2-7B int8, 32 distinct slabs per
// graph (tools/mb_q8ring.hip, profiles/r5_int8_ring_ab.txt): ffn13 17.7-17.9 -> 16.6-16.7 us (-6...-7 %), cls
// 24.35 -> 23.5 (-3.5 %); qkv +0.7 %, w2 -1.3 %, wo +5.8 % - those three stay on the register-tile kernels.
// Two ring slots per wave and two 256-thread workgroups per CU measured best (deeper rings and more waves per CU
// are slower: 3 slots +1 %, 4 slots +5 %, three workgroups per CU +3 %).  KH_RING=0 turns the ring kernels off.
void plan_ring(bool quant, int dim, int hidden_dim, int vocab_size, int group_size, kh_model::RingPlan* out) {
  *out = kh_model::RingPlan();
  if (!quant || dbg_off("KH_RING")) return;
  if (dim % 16 != 0 || !kh_stage_fits4(dim, KH_WG)) return;
  if (group_size < 16 || (group_size & (group_size - 1)) != 0 || dim % group_size != 0) return;
  if (ring_lds_bytes(dim, false, KH_WAVES_PER_WG, 2) > 64 * 1024) return;  // no dynamic-LDS opt-in on this path
  auto grid_of = [](int items) {
    const int need = (items + KH_WAVES_PER_WG - 1) / KH_WAVES_PER_WG;
    return need < 512 ? need : 512;
  };
  // a KH_SHAPE_FFN / KH_SHAPE_CLS hook asks for a specific register-tile launch: honour it (the B-token prefill
  // follows the same hook, and its bit-identity with decode needs the same workgroup width on both sides)
  if (!dbg("KH_SHAPE_FFN")) {
    out->ffn_r = 2;
    out->ffn_grid = grid_of(hidden_dim);
  }
  if (!dbg("KH_SHAPE_CLS")) {
    out->cls_r = 2;
    out->cls_grid = grid_of((vocab_size + 1) / 2);
  }
}
extern "C" int kh_plan_decode_ring(int32_t dim, int32_t hidden_dim, int32_t vocab_size, int32_t is_quant,
                                   int32_t group_size, int32_t* out4) {
  if (!out4 || dim <= 0 || hidden_dim <= 0 || vocab_size <= 0) return KH_ERR_INVALID_ARG;
  kh_model::RingPlan r;
  plan_ring(is_quant != 0, dim, hidden_dim, vocab_size, group_size, &r);
  out4[0] = r.ffn_r;
  out4[1] = r.ffn_grid;
  out4[2] = r.cls_r;
  out4[3] = r.cls_grid;
  return KH_OK;
}

// Host-only view of that plan for tools and the CPU test-suite: out[5][4] = {split, u, grid, wg} of qkv, wo,
// ffn13, w2, cls for a geometry (no device is touched).
extern "C" int kh_plan_decode_shapes(int32_t dim, int32_t hidden_dim, int32_t kv_dim, int32_t vocab_size,
                                     int32_t is_quant, int32_t* out20) {
  if (!out20 || dim <= 0 || hidden_dim <= 0 || kv_dim <= 0 || vocab_size <= 0) return KH_ERR_INVALID_ARG;
  kh_model::Shape sh[5];
  plan_decode_shapes(is_quant != 0, dim, hidden_dim, kv_dim, vocab_size, sh);
  for (int i = 0; i < 5; ++i) {
    out20[4 * i] = sh[i].split;
    out20[4 * i + 1] = sh[i].u;
    out20[4 * i + 2] = sh[i].grid;
    out20[4 * i + 3] = sh[i].wg;
  }
  return KH_OK;
}

int configure_step_kernels(kh_model* m) {
  const kh_config& c = m->cfg;
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
    KH_ATTR(true, 3, 1); KH_ATTR(true, 3, 2); KH_ATTR(true, 3, 4);
    KH_ATTR(true, 2, 1); KH_ATTR(true, 2, 2); KH_ATTR(true, 2, 4);
#undef KH_ATTR
  }
  return KH_OK;
}

}  // namespace khm
using namespace khm;

extern "C" int kh_model_predict(kh_model* m, int32_t token, int32_t pos, int32_t is_prompt,
                                int32_t exec, int32_t* h_next) {
  if (!m || !h_next) return KH_ERR_INVALID_ARG;
  const kh_config& c = m->cfg;
  if (token < 0 || token >= c.vocab_size || pos < 0 || pos >= c.cache_len) return KH_ERR_RANGE;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  int rc = kv_ensure(m, pos + 1);  // cache rows 0 .. pos backed by HBM before the step is enqueued
  if (rc != KH_OK) return rc;
  set_state(m, token, pos);  // embedding() + fill_input (llama3.cpp:578-598, model.cpp:245-263)
  if (exec == KH_EXEC_UNFUSED) {
    rc = launch_step_unfused(m, pos);
  } else if (exec == KH_EXEC_FUSED || exec == KH_EXEC_GRAPH) {
    launch_step_fused(m, /*advance=*/0, /*n_forced=*/0, nullptr, step_variant(m, pos, pos));
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
  // every cache row this call can reach is backed by HBM before its first launch (the dry launches of fresh graphs
  // below touch rows 0 .. 7); mapping happens here, on the host, outside the event bracket of the step loop
  if ((rc = kv_ensure(m, total_steps < KH_GRAPH_STEPS ? KH_GRAPH_STEPS : total_steps)) != KH_OK) return rc;
  // pinned mirror of the words, sized like the device buffers so that a longer run later does not re-allocate it (a
  // hipHostMalloc inside the step loop's event bracket stalled the first 20-step run behind a 5-step one by 0.3 ms)
  if ((rc = ensure_pinned_words(m, m->seq_cap)) != KH_OK) return rc;
  // forced[i] = token fed at position i while inside the prompt, -1 afterwards.  Staged in the model's pinned
  // buffer: the upload is ordered before the steps by the stream and needs no host-side wait.  A generate that
  // returned early on an error may have left its upload in flight: the buffer is refilled only behind a stream
  // sync in that case (forced_in_flight; free when the previous call ended normally - it synchronised itself).
  // Only the first total_steps + 1 entries are written; whatever an earlier, longer run left beyond them goes back
  // to -1 (time_step / profile_step at deeper positions must not feed a stale prompt token).
  {
    if (m->forced_in_flight) KH_CHECK_HIP(hipStreamSynchronize(m->stream));
    const int nf = total_steps + 1 <= m->seq_cap + 1 ? total_steps + 1 : m->seq_cap + 1;
    for (int i = 0; i < nf; ++i) m->h_forced_pin[i] = i < n_prompt ? h_prompt[i] : -1;
    m->forced_in_flight = true;
    KH_CHECK_HIP(hipMemcpyAsync(m->d_forced, m->h_forced_pin, (size_t)nf * sizeof(int32_t), hipMemcpyHostToDevice,
                                m->stream));
    if (m->forced_hwm > nf)
      KH_CHECK_HIP(hipMemsetAsync(m->d_forced + nf, 0xFF, (size_t)(m->forced_hwm - nf) * sizeof(int32_t), m->stream));
    m->forced_hwm = nf;
  }
  const int n_forced = m->seq_cap + 1;
  if (exec == KH_EXEC_GRAPH) {
    // all four graphs of variant 0 (1 / 2 / 4 / 8 steps) exist after the first generate of a model, whatever its
    // length (a warm-up run of 5 steps must leave the 8-step graph behind: capturing 656 nodes costs ~0.8 ms, which a
    // 20-step run would otherwise pay inside its timed loop); the other variants are captured when a run first
    // reaches position 256
    // ... and each has been LAUNCHED once: the first launch of an instantiated graph costs ~0.1-0.3 ms on this
    // runtime even after hipGraphUpload (a 20-step run behind a 5-step warm-up: 1012 tok/s, every later one 1028 on
    // the same model instance).  Every dry launch starts at position 0 (set_state before each one), so together they
    // write cache rows / words 0 .. 7 and read forced[1 .. 8] - uploaded above, or -1 beyond this call's prompt:
    // rows this very call rewrites (generate always starts a new sequence at position 0) unless total_steps < 8, in
    // which case rows total_steps .. 7 hold the K/V of a throw-away continuation afterwards (kuiper_hip.h says so:
    // a generate owns rows [0, max(total_steps, 8)) of the cache).  Skipped when the cache is shorter than that.
    // (Launching only the graphs of at most total_steps steps - the first r5 form - left the 8-step graph's first
    // launch inside the timed loop of a 20-step run behind a 5-step warm-up: 1018 instead of 1036-1042 tok/s.)
    bool fresh[4] = {false, false, false, false};
    hipGraphExec_t ge = nullptr;
    for (int n = 1, k = 0; n <= KH_GRAPH_STEPS; n *= 2, ++k) {
      fresh[k] = m->sg[0][k].e == nullptr;
      if ((rc = step_graph_n(m, n_forced, 0, n, &ge)) != KH_OK) return rc;
    }
    bool dry = false;
    if (c.cache_len >= KH_GRAPH_STEPS && m->seq_cap >= KH_GRAPH_STEPS)
      for (int k = 3; k >= 0; --k)
        if (fresh[k]) {
          set_state(m, h_prompt[0], 0);
          KH_CHECK_HIP(hipGraphLaunch(m->sg[0][k].e, m->stream));
          dry = true;
        }
    if (dry) KH_CHECK_HIP(hipStreamSynchronize(m->stream));
  }

  // prompt phase: the tokens that are only fed (positions 0 .. n_prompt-2).  KH_PREFILL selects how:
  //   "0" / "token"  the reference's one forward pass per prompt token (demo/main.cpp:20-22)
  //   "gemv"         B-token VALU kernels: K/V rows bit-identical to the token-by-token ones
  //   "gemm"         fp32-MFMA GEMM prefill: rows equal to fp32 round-off (tolerance, NOT bit-identity:
  //                  greedy tokens can differ from the token-by-token path at near-ties)
  //   unset          KH_FLAG_PREFILL_EXACT: "gemv" always; otherwise "gemm" from KH_PG_MIN_TOKENS fed-only tokens
  //                  on, "gemv" below that
  // any other value is an error (KH_ERR_INVALID_ARG), not a silent choice.
  int start = 0;
  KH_CHECK_HIP(hipEventRecord(m->ev0, m->stream));
  if (n_prompt - 1 >= 2 && n_prompt - 1 < total_steps) {
    const char* e = dbg("KH_PREFILL");
    bool want_gemm = n_prompt - 1 >= KH_PG_MIN_TOKENS && !(m->opts.flags & KH_FLAG_PREFILL_EXACT), want_gemv = true;
    if (e && *e) {
      if (!strcmp(e, "0") || !strcmp(e, "token")) want_gemm = want_gemv = false;
      else if (!strcmp(e, "gemv")) want_gemm = false;
      else if (!strcmp(e, "gemm")) want_gemm = true;
      else return KH_ERR_INVALID_ARG;
    }
    if (want_gemm && pg_supported(m)) {
      if ((rc = kh_model_prefill_gemm(m, h_prompt, n_prompt - 1, 0)) != KH_OK) return rc;
      start = n_prompt - 1;
      m->first_mode = 2;
    } else if (want_gemv && prefill_supported(m)) {
      if ((rc = kh_model_prefill(m, h_prompt, n_prompt - 1, 0)) != KH_OK) return rc;
      start = n_prompt - 1;
      m->first_mode = 1;
    }
  }
  // near-tie report (kh_model_first_sample): with a prefill the first sampled step runs on its own and its logits
  // are put aside before the next step overwrites them
  m->first_pos = -1;
  if (start > 0) {
    if (!m->first_logits) KH_CHECK_HIP(hipMalloc(&m->first_logits, sizeof(float) * (size_t)c.vocab_size));
    m->first_pos = start;
  }
  set_state(m, h_prompt[start], start);
  auto launch_chunk = [&](int s) -> int {  // enqueue the next 1 or KH_GRAPH_STEPS steps (positions s ..)
    if (exec == KH_EXEC_GRAPH) {
      // the largest of 8 / 4 / 2 / 1 steps that still fits (a single-step launch costs ~15 us of graph-launch gap:
      // the 20-step form of the bench ran 8 + 8 + 1 + 1 + 1 + 1 and lost 0.3 % to it; now 8 + 8 + 4)
      int n = KH_GRAPH_STEPS;
      while (n > total_steps - s) n >>= 1;
      const bool keep = s == start && start > 0;  // first sampled step behind a prefill: alone, logits kept
      if (keep) n = 1;
      hipGraphExec_t ge = nullptr;
      if (step_graph_n(m, n_forced, step_variant(m, s, s + n - 1), n, &ge) != KH_OK) return -1;
      if (hipGraphLaunch(ge, m->stream) != hipSuccess) return -1;
      if (keep && hipMemcpyAsync(m->first_logits, m->logits, sizeof(float) * (size_t)c.vocab_size,
                                 hipMemcpyDeviceToDevice, m->stream) != hipSuccess)
        return -1;
      return n;
    }
    launch_step_fused(m, 1, n_forced, nullptr, step_variant(m, s, s));
    if (s == start && start > 0 &&
        hipMemcpyAsync(m->first_logits, m->logits, sizeof(float) * (size_t)c.vocab_size, hipMemcpyDeviceToDevice,
                       m->stream) != hipSuccess)
      return -1;
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
    // through the pinned mirror: a device-to-pageable copy is staged and synchronised by the runtime on top of ours
    KH_CHECK_HIP(hipMemcpyAsync(m->h_words_pin, m->d_words, sizeof(int32_t) * total_steps,
                                hipMemcpyDeviceToHost, m->stream));
    KH_CHECK_HIP(hipStreamSynchronize(m->stream));
    memcpy(h_words, m->h_words_pin, sizeof(int32_t) * (size_t)total_steps);
    for (int i = 0; i < start; ++i) h_words[i] = h_prompt[i + 1];  // forced, main.cpp:36-38
  } else {
    // Stop-token check without a per-step host round trip: the words of every
    // chunk of steps are mirrored into pinned memory behind the chunk, and the host inspects
    // chunk k while chunk k+1 is already queued, so the GPU never waits for the check.  At
    // most two chunks of steps run past the stop token; their words are discarded.
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
  m->forced_in_flight = false;  // both branches above end with a stream sync
  if (h_elapsed_ms) KH_CHECK_HIP(hipEventElapsedTime(h_elapsed_ms, m->ev0, m->ev1));
  *n_words = n_out;
  return KH_OK;
}
