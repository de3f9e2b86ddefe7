// kh_model_internal.h — state and cross-unit helpers of the model level of the C-ABI.
// The model level is split into four translation units:
//   kh_model_load.hip     .bin image -> HBM arena, weight table, buffers, create / destroy, cache I/O
//   kh_model_step.hip     launch shapes, the fused and unfused decode step, hipGraph capture,
//                         predict / generate (kh_fused.h kernels are instantiated here)
//   kh_model_prefill.hip  B-token VALU prefill and the MFMA GEMM prefill (kh_prefill.h, kh_gemm.h,
//                         kh_pattn.h kernels)
//   kh_model_profile.hip  per-kernel / per-step timing entry points
// gfx950 only.  No CPU fallback: every path launches HIP kernels.
#pragma once
#include <new>
#include <thread>
#include <stdint.h>

#include <vector>

#include "kh_attn.h"
#include "kh_common.h"

namespace khm {
// No C++ exception crosses the C ABI: the entry points that allocate host containers or start threads run their
// body through this (std::bad_alloc -> hipErrorOutOfMemory, anything else -> KH_ERR_INTERNAL).
template <class F>
static inline int kh_api_guard(F&& body) noexcept {
  try {
    return body();
  } catch (const std::bad_alloc&) {
    return (int)hipErrorOutOfMemory;
  } catch (...) {
    return KH_ERR_INTERNAL;
  }
}
// joins a helper thread on every way out of a scope (an exception while it runs would otherwise terminate)
struct KhJoinOnExit {
  std::thread& t;
  ~KhJoinOnExit() {
    if (t.joinable()) t.join();
  }
};

struct LayerW {
  KhLin wq, wk, wv, wo, w1, w2, w3;
  const float* att_norm;
  const float* ffn_norm;
};

}  // namespace khm
using khm::LayerW;

#define KH_STEP_VARIANTS 3  // kh_model_step.hip::step_variant
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
  // KV cache on reserved addresses, physical memory mapped on demand (kh_model_load.hip::kv_ensure): the contiguous
  // [layer, cache_len, kv_dim] addressing of llama3.cpp:469-472 with HBM committed only for the rows a sequence reached
  struct KvVmm {
    bool on = false;
    size_t chunk = 0;      // mapping unit in bytes (a multiple of the allocation granularity)
    size_t reserved = 0;   // bytes reserved per cache (K and V each)
    size_t mapped = 0;     // bytes mapped, K + V
    std::vector<uint8_t> have[2];  // per chunk of the K / V reservation: mapped?
    struct Run {
      void* va;
      size_t len;
      hipMemGenericAllocationHandle_t h;
    };
    std::vector<Run> runs;
    int rows_all = 0;      // rows [0, rows_all) of EVERY layer are mapped
  };
  KvVmm kv;
  float load_ms = 0.f;      // host image -> HBM upload time (kh_model_get_load_ms)
  std::thread unmap_thread;  // kh_model_create_from_file: munmap of the file off the critical path, joined by destroy
  void* attn_ws = nullptr;  // split-T attention partials + tickets (kh_attn.h)
  int attn_ns = 1;
  int attn_ns_g = 0;        // GQA long-context path: splits per KV group (0 = path off)
  int attn_ws_stride = 1;   // split slots per head in attn_ws
  int attn_t_long = 1 << 30;
  int attn_wg = KH_WG;
  int attn_ts_shift = 8;     // log2 of the per-head split quantum (kh_attn.h::attn_ts_shift_for)
  bool attn_fenced = false;  // KH_FLAG_ATTN_MERGE_FENCED / KH_ATTN_FENCED: fences around the in-launch split merge
  bool attn_defer = false;  // variant 1 exists: split partials combined by kh_fused.h::k_wo_comb
  int attn_defer_max = 0;   // ... up to this many active splits (more: the in-launch merge is as fast or faster)
  int step_var = 0;         // variant the launch_* helpers use right now (set by launch_step_fused / profile)
  int32_t *d_pos = nullptr, *d_token = nullptr, *d_next = nullptr, *d_forced = nullptr,
          *d_words = nullptr;
  int seq_cap = 0;  // capacity of d_forced / d_words
  // prefill (kh_prefill.h): residual / q / attention / hidden rows of up to KH_PF_BMAX prompt tokens
  float *pf_x = nullptr, *pf_q = nullptr, *pf_att = nullptr, *pf_h = nullptr;
  void* pf_ws = nullptr;        // KH_PF_BMAX attention split workspaces
  size_t pf_ws_tok_bytes = 0;
  // GEMM prefill (kh_gemm.h): slabs of KH_PG_TMAX token rows
  float *pg_x = nullptr, *pg_xn = nullptr, *pg_q = nullptr, *pg_att = nullptr, *pg_h = nullptr;
  float* pg_part = nullptr;     // partial rows of residual GEMMs that split K across workgroups
  void* pg_ws = nullptr;        // KH_PG_TMAX attention split workspaces
  size_t pg_ws_tok_bytes = 0;
  bool pf_ready = false, pg_ready = false;  // set when ALL prefill slabs exist (allocation can fail half-way)
  bool pg_launch_failed = false;
  int32_t* h_words_pin = nullptr;  // pinned mirror of d_words (stop-token check, final copy-out)
  int32_t* h_forced_pin = nullptr; // pinned staging of d_forced [seq_cap + 1]: the upload needs no host sync
  int forced_hwm = 0;              // entries of d_forced that may differ from -1 (the last generate's upload)
  bool forced_in_flight = false;   // a generate returned before its final stream sync: h_forced_pin may be under DMA
  // near-tie report (kh_model_first_sample): logits of the first sampled step of the last generate with a prefill
  float* first_logits = nullptr;  // [vocab], allocated on first use
  int first_pos = -1;             // -1: the last generate had no prefill phase
  int first_mode = 0;             // 1 = kh_model_prefill, 2 = kh_model_prefill_gemm
  int pin_cap = 0;
  hipEvent_t ev_chunk[2] = {nullptr, nullptr};
  // launch geometry
  struct Shape {
    int u = 2, split = 1, grid = 1, wg = KH_WG;
  };
  Shape sh_qkv, sh_wo, sh_ffn, sh_w2, sh_cls;
  // int8 ffn13 / cls on the LDS-DMA ring kernels (kh_fused_ring.h): ring slots per wave (0 = the register-tile
  // kernel of sh_ffn / sh_cls) and workgroups (256 threads each); chosen by plan_ring
  struct RingPlan {
    int ffn_r = 0, ffn_grid = 0, cls_r = 0, cls_grid = 0;
  };
  RingPlan ring;
  // graph
  // the decode step captured once as a 1-step graph and once as a KH_GRAPH_STEPS-step graph:
  // consecutive hipGraphLaunch calls leave the GPU idle for ~8 us (measured), so the long
  // graph amortises that gap over several tokens
  // ... each in three VARIANTS of the attention / wo pair (kh_model_step.hip::step_variant): 0 = the
  // attention launch merges its time splits itself (valid at every position; nothing to merge below
  // position 256), 1 = the splits are merged by k_wo_comb (positions on the per-head path only), 2 = as 0
  // with the per-head-only attention instantiation (positions below the group path, merge not deferrable)
  struct StepGraph {
    hipGraph_t g = nullptr;
    hipGraphExec_t e = nullptr;
  };
  StepGraph sg[KH_STEP_VARIANTS][4];  // [variant][log2 steps]: graphs of 1, 2, 4 and KH_GRAPH_STEPS = 8 steps
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
};

#define KH_GRAPH_STEPS 8
#define KH_PG_MIN_TOKENS 16  // prompts with fewer fed-only tokens stay on the bit-identical path

namespace khm {
template <typename T>
int dalloc(T** p, size_t n) {
  hipError_t e = hipMalloc((void**)p, n * sizeof(T));
  return e == hipSuccess ? KH_OK : (int)e;
}

// ---- kh_model_step.hip ----------------------------------------------------------------------
kh_model::Shape pick_shape(bool quant, int pairs, int M, int max_split, const char* env, int wg = KH_WG,
                           int wg_max = KH_WG, bool many_waves = false, bool u3 = false);
// launch geometry of the five GEMV kernels of a decode step: qkv, wo, ffn13, w2, cls (host-only)
void plan_decode_shapes(bool quant, int dim, int hidden_dim, int kv_dim, int vocab_size, kh_model::Shape (&out)[5]);
// which int8 GEMVs of a decode step run on the LDS-DMA ring kernels, and their launch geometry (host-only)
void plan_ring(bool quant, int dim, int hidden_dim, int vocab_size, int group_size, kh_model::RingPlan* out);
int configure_step_kernels(kh_model* m);  // >64 KiB dynamic-LDS opt-in of the hidden-sized GEMVs
KhAttnArgs fill_attn(kh_model* m, int l);
void launch_qkv(kh_model* m, int l);
void launch_attn(kh_model* m, int l);
void launch_wo(kh_model* m, int l);  // follows m->step_var like launch_attn
void launch_ffn13(kh_model* m, int l);
void launch_w2(kh_model* m, int l);
void launch_cls(kh_model* m);
void launch_sample(kh_model* m, int advance, int n_forced);
// variant (see kh_model::sg): which attention / wo pair the launches of a step use
int step_variant(const kh_model* m, int pos_lo, int pos_hi);
void launch_step_fused(kh_model* m, int advance, int n_forced, hipEvent_t* ev, int variant);
int launch_step_unfused(kh_model* m, int pos);
void set_state(kh_model* m, int token, int pos);
int ensure_pinned_words(kh_model* m, int n);
int ensure_seq_cap(kh_model* m, int n);
void destroy_step_graphs(kh_model* m);
// the captured graph of 1 (steps8 = false) or KH_GRAPH_STEPS decode steps in `variant`, captured on first use
int step_graph(kh_model* m, int n_forced, int variant, bool steps8, hipGraphExec_t* out);
// the same for a graph of `nsteps` in {1, 2, 4, 8} steps (the tail of a run: 20 steps = 8 + 8 + 4)
int step_graph_n(kh_model* m, int n_forced, int variant, int nsteps, hipGraphExec_t* out);
// ---- kh_model_load.hip ----------------------------------------------------------------------
// Make rows [0, rows) of the K / V cache usable before anything that touches them is enqueued: every layer, or one
// (layer >= 0).  No-op for rows that are mapped already and for caches that are plainly allocated.  Newly mapped
// memory is zeroed on the model's stream.
int kv_ensure(kh_model* m, int rows, int layer = -1);
// ---- kh_model_selftest.hip ------------------------------------------------------------------
// ring kernels vs register tiles, fence-free vs fenced split merge: once at the end of kh_model_create_*
int run_selftests(kh_model* m);
// ---- kh_model_prefill.hip -------------------------------------------------------------------
bool prefill_supported(const kh_model* m);  // B-token VALU path
bool pg_supported(const kh_model* m);       // MFMA GEMM path
}  // namespace khm
