// kh_attn.h — single-query (decode) attention over the contiguous KV cache, one workgroup per
// query head.  reference: cuda/mha_kernel.cu:47-110 (+ softmax_gpu :7-45), cpu/mha_kernel.cpp:5-61.
//
// KV layout is the reference's: cache[layer, t, kv_dim]; head h reads columns
// (h/kv_mul)*hs .. +hs of every row t <= pos (row stride kv_dim floats).
//
// Mapping (wave64): G = pow2 >= hs/4 lanes cooperate on one timestep, each owning one float4
// of the head vector, so a wave instruction reads 64/G whole K (or V) head-rows of hs*4
// contiguous bytes — coalesced, unlike the reference's thread-per-timestep walk with a
// kv_dim stride.  The q.k reduction is log2(G) shuffles.  Scores never touch global memory
// (the reference round-trips them through kScoreStorage); they live in an LDS chunk of
// KH_ATTN_TC timesteps, with flash-style running (max, sum) rescaling across chunks so any
// pos < seq_len works.  P.V uses the same lane mapping: every lane accumulates its float4 of
// the output over its timesteps, then timestep-groups are folded with shuffles + one LDS
// exchange across the 4 waves.
#pragma once
#include <stdlib.h>

#include "kh_common.h"

#define KH_ATTN_TC 2048  // timesteps per LDS score chunk (8 KiB)

// Phase stamps of the decode launch (tools/mb_attn_phases.hip builds this header with -DKH_ATTN_TRACE and reads the
// buffer back: where a launch's microseconds go, per workgroup).  Compiled out of the product.
#ifdef KH_ATTN_TRACE
__device__ unsigned long long* kh_attn_trace_buf;  // [gridDim.x][32] of the 100-MHz constant clock
#define KH_ATTN_STAMP(i)                                                                     \
  do {                                                                                       \
    if (threadIdx.x == 0) kh_attn_trace_buf[(size_t)blockIdx.x * 32 + (i)] = wall_clock64(); \
  } while (0)
// per wave (slots 8 + 8 * i + wave, i = 0: first batch consumed, 1: last batch consumed)
#define KH_ATTN_STAMP_W(i)                                                                                         \
  do {                                                                                                             \
    if ((threadIdx.x & 63) == 0)                                                                                   \
      kh_attn_trace_buf[(size_t)blockIdx.x * 32 + 8 + 8 * (i) + (threadIdx.x >> 6)] = wall_clock64();              \
  } while (0)
#else
#define KH_ATTN_STAMP(i) do { } while (0)
#define KH_ATTN_STAMP_W(i) do { } while (0)
#endif

static inline size_t attn_lds_bytes(int head_size, int wg = KH_WG) {
  return (size_t)(KH_ATTN_TC + 8 + (wg / KH_WAVE) * head_size) * sizeof(float);
}

// q_h: [hs] ; k_base/v_base: cache + layer offset + head column offset ; kv_stride = kv_dim.
// out_h: [hs].  score_out (optional, global [>= pos+1]): receives the softmax probabilities like
// the reference's score tensor.  smem: attn_lds_bytes(hs) bytes, 16-B aligned.
__device__ __forceinline__ void attn_head_decode(const float* __restrict__ q_h,
                                                 const float* __restrict__ k_base,
                                                 const float* __restrict__ v_base, int kv_stride,
                                                 int hs, int pos, float* __restrict__ out_h,
                                                 float* __restrict__ score_out, float* smem) {
  float* sc = smem;                    // [KH_ATTN_TC]
  float* red = smem + KH_ATTN_TC;      // [8]
  float* opart = red + 8;              // [4][hs]
  const int tid = threadIdx.x;
  const int hs4 = hs >> 2;
  int G = 1;
  while (G < hs4) G <<= 1;             // lanes per timestep (<= 64 since hs <= 256)
  const int TPI = kh_wg() / G;         // timesteps per workgroup iteration
  const int tg = tid / G, dl = tid - tg * G;
  const bool active = dl < hs4;
  const int stride4 = kv_stride >> 2;
  const f32x4* K4 = (const f32x4*)k_base;
  const f32x4* V4 = (const f32x4*)v_base;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const f32x4 q4 = active ? ((const f32x4*)q_h)[dl] : zero4;
  const float scale = 1.0f / sqrtf((float)hs);   // cpu/mha_kernel.cpp:11
  const int nT = pos + 1;
  const bool single = nT <= KH_ATTN_TC;

  float m_run = -INFINITY, l_run = 0.f;
  f32x4 o4 = zero4;

  for (int t0 = 0; t0 < nT; t0 += KH_ATTN_TC) {
    const int tc = min(KH_ATTN_TC, nT - t0);
    // ---- scores: s[t] = (q . K[t]) * scale ---------------------------------------------
    for (int tb = 0; tb < tc; tb += TPI * 4) {
      f32x4 kv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = tb + u * TPI + tg;
        const int tt = t < tc ? t : tc - 1;
        kv[u] = active ? K4[(size_t)(t0 + tt) * stride4 + dl] : zero4;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = tb + u * TPI + tg;
        float d = fma4(q4, kv[u], 0.f);
        for (int off = G >> 1; off > 0; off >>= 1) d += __shfl_xor(d, off, KH_WAVE);
        if (dl == 0 && t < tc) {
          const float s = d * scale;
          sc[t] = s;
          if (score_out) score_out[t0 + t] = s;  // raw score; normalised in the last pass
        }
      }
    }
    __syncthreads();
    // ---- running softmax statistics --------------------------------------------------------
    float mx = -INFINITY;
    for (int t = tid; t < tc; t += kh_wg()) mx = fmaxf(mx, sc[t]);
    mx = block_max(mx, red);
    const float m_new = fmaxf(m_run, mx);
    float s = 0.f;
    for (int t = tid; t < tc; t += kh_wg()) {
      const float e = expf(sc[t] - m_new);
      sc[t] = e;
      s += e;
    }
    s = block_sum(s, red);
    const float alpha = expf(m_run - m_new);  // exp(-inf) = 0 on the first chunk
    l_run = l_run * alpha + s;
    o4 = o4 * alpha;
    m_run = m_new;
    if (single) {
      // one chunk: normalise BEFORE the weighted sum like cpu/softmax_kernel.cpp:13-14
      for (int t = tid; t < tc; t += kh_wg()) sc[t] = sc[t] / s;
      __syncthreads();
    }
    // ---- o += sum_t p[t] * V[t] --------------------------------------------------------------
    for (int tb = 0; tb < tc; tb += TPI * 4) {
      f32x4 vv[4];
      float p[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int t = tb + u * TPI + tg;
        const int tt = t < tc ? t : tc - 1;
        vv[u] = active ? V4[(size_t)(t0 + tt) * stride4 + dl] : zero4;
        p[u] = t < tc ? sc[tt] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        o4.x = __builtin_fmaf(p[u], vv[u].x, o4.x);
        o4.y = __builtin_fmaf(p[u], vv[u].y, o4.y);
        o4.z = __builtin_fmaf(p[u], vv[u].z, o4.z);
        o4.w = __builtin_fmaf(p[u], vv[u].w, o4.w);
      }
    }
    __syncthreads();  // sc is rewritten by the next chunk
  }

  // ---- fold the timestep groups: lanes with equal dl inside a wave, then the 4 waves --------
  for (int off = G; off < KH_WAVE; off <<= 1) {
    o4.x += __shfl_xor(o4.x, off, KH_WAVE);
    o4.y += __shfl_xor(o4.y, off, KH_WAVE);
    o4.z += __shfl_xor(o4.z, off, KH_WAVE);
    o4.w += __shfl_xor(o4.w, off, KH_WAVE);
  }
  const int lane = tid & 63, wave = tid >> 6;
  if (G <= KH_WAVE && lane < G && active) ((f32x4*)(opart + wave * hs))[dl] = o4;
  __syncthreads();
  if (tid < hs) {
    float r = 0.f;
    // when G == 256/TPI... every wave holds a partial for every d (TPI >= 4 <=> G <= 64)
    const int nw = kh_nwaves();
#pragma unroll
    for (int w = 0; w < KH_WAVES_MAX; ++w) r += w < nw ? opart[(w < nw ? w : 0) * hs + tid] : 0.f;
    out_h[tid] = single ? r : r / l_run;
  }
  if (score_out) {
    // probabilities as the reference leaves them in the score tensor
    const float inv_l = l_run;
    for (int t = tid; t < nT; t += kh_wg()) score_out[t] = expf(score_out[t] - m_run) / inv_l;
  }
}

// ---------------------------------------------------------------------------------------------
// Low-latency variant for the fused decode step (no score output).  At decode positions of a
// few hundred the K/V bytes are tiny (<= 64 KiB per head) and the kernel is pure latency, so:
//  * q, K and V loads of a batch of 4 timesteps per lane-group are all issued up front — ONE
//    memory round trip instead of K-then-V;
//  * every G-lane group keeps its own running (max, sum, o) in registers (flash-style), so
//    there is no score buffer and no workgroup-wide max/sum pass; the q.k reduction is DPP;
//  * groups are merged once at the end: permlane swaps inside the wave against the wave's own maximum, one
//    LDS exchange across the waves (ONE barrier in the whole kernel since round 6).
// G = lanes per timestep = pow2 >= hs/4, must be 16, 32 or 64 (hs 33..256).
//
// Long contexts: the grid carries NS workgroups per head ("splits"); at run time the first
// nact = ceil((pos+1)/TS) of them each take TS timesteps (TS >= 256, so positions < 256 use
// exactly one workgroup per head and none of the machinery below) and the rest exit.  With
// nact > 1 every split leaves its unnormalised (M, L, o[hs]) in the workspace.  Two ways to combine them:
//  * DEFERRED (the fused decode step between position 256 and the GQA group path): the split
//    workgroups store and STOP; the kernel that consumes the attention output anyway (wo,
//    kh_fused.h::k_wo_comb) combines the <= 16 partials of every head in fixed order while it stages
//    its input vector.  No ticket, no fence, no last arriver: the kernel boundary that exists already
//    carries the visibility (r3: 12.8 us per layer at pos 4095, of which ~9 were the publish -> ticket ->
//    acquire -> merge chain).  The host picks this pair of kernels per captured graph from the
//    position range the graph covers (kh_model_step.hip::step_variant); the arithmetic is the in-launch
//    merger's, bit for bit.
//  * IN-LAUNCH (operator entry points, prefill slices, the GQA group path whose 32 splits per group
//    are too many to re-read in every wo workgroup): publish with agent-scope write-through stores
//    (global_store sc1: no release fence, guide G16 R1), every storing wave drains vmcnt, barrier, one
//    lane takes a relaxed agent ticket; the LAST arriver merges with agent-scope loads (sc1: served
//    past L1, so no acquire fence either).  Placement-independent (splits of a head land on different
//    XCDs) and never waits on another workgroup, so it cannot hang.  The merger re-arms the ticket.
#ifndef KH_ATTN_UB
#define KH_ATTN_UB 4
#endif
#ifndef KH_ATTN_MIN_TS
#define KH_ATTN_MIN_TS 256  // timesteps a head keeps in ONE split; also the split quantum of the GQA group path
#endif
#define KH_ATTN_MAX_NS 16
// Per-head path: split quantum 1 << ts_shift timesteps, chosen per geometry (attn_ts_shift_for).  A split streams
// 2 * quantum * head_size * 4 bytes of K/V from ONE workgroup, and splitting costs a merge (2.3-3 us on Llama-2-7B,
// in the wo kernel or in the last arriver): a head stays in one split up to KH_ATTN_MIN_TS timesteps whatever the
// quantum, beyond that the quantum that measured best keeps a split at about 128 KiB - 256 timesteps at head size
// 64 (Llama-3.2-1B: 128 loses 2 % at position 2047 and 2.5 % at 4095, profiles/r4_attn_min_ts.txt), 128 at head
// size 128 (Llama-2-7B: -3.6 % per token at position 383, -3.9 % at 511, -2 % at 1023, -0.6 % at 2047;
// profiles/r5_attn_7b_ts.txt).  Hook KH_ATTN_TS = 64 | 128 | 256 overrides.
#define KH_ATTN_TS_SHIFT_MAX 8
#ifndef KH_ATTN_TSG_SHIFT
#define KH_ATTN_TSG_SHIFT 8  // GQA group path: log2 of its split quantum
#endif
#define KH_ATTN_TLONG_DEFAULT 4096  // pos + 1 from which GQA models switch to the group path
#ifndef KH_ATTN_MAX_NS_G
// splits per KV group (the last arriver merges them all).  16 / 24 / 48 / 64 measured: 48 / 64 (two workgroups per CU) far worse
// from position 16383 on: 22.6 -> 34-35 us, 93.9 -> 110-120 us at 131071 ; 16 / 24 lose from 32767 on (profiles/r3_attn_nsg.txt)
#define KH_ATTN_MAX_NS_G 32
#endif
#define KH_ATTN_MIN_GROUPS 4         // fewer KV heads than this: too few workgroups, stay per-head
static inline size_t attn_fast_lds_bytes(int head_size, int wg = KH_WG) {
  return (size_t)(8 + 8 + (wg / KH_WAVE) * head_size) * sizeof(float);
}
static inline int attn_ts_shift_for(int head_size) {
  if (const char* e = khm::dbg("KH_ATTN_TS")) {
    const int v = atoi(e);
    if (v == 64) return 6;
    if (v == 128) return 7;
    if (v == 256) return 8;
  }
  return head_size >= 128 ? 7 : 8;
}
// splits per head carried by the grid for a cache of `cache_len` rows
static inline int attn_num_splits(int cache_len, int ts_shift = KH_ATTN_TS_SHIFT_MAX) {
  if (cache_len <= KH_ATTN_MIN_TS) return 1;
  int ns = (cache_len + (1 << ts_shift) - 1) >> ts_shift;
  if (ns < 1) ns = 1;
  if (ns > KH_ATTN_MAX_NS) ns = KH_ATTN_MAX_NS;
  return ns;
}
// workspace for the split merge: [heads] int tickets | [heads*NS*2] (M,L) | [heads*NS*hs] o, every region rounded up
// to 16 bytes: k_wo_comb's CombStager reads `o` with dwordx4 loads (head counts that are not a multiple of four -
// Qwen2.5-0.5B: 14 - used to leave it 8 bytes off), and a per-token workspace stride stays 16-byte aligned
__host__ __device__ static inline size_t attn_ws_cnt_words(int heads) { return ((size_t)heads + 3) & ~(size_t)3; }
__host__ __device__ static inline size_t attn_ws_ml_words(int heads, int ns) {
  return ((size_t)heads * ns * 2 + 3) & ~(size_t)3;
}
static inline size_t attn_ws_bytes(int heads, int head_size, int ns) {
  if (ns <= 1) return 0;
  const size_t words = attn_ws_cnt_words(heads) + attn_ws_ml_words(heads, ns) + (size_t)heads * ns * head_size;
  return ((words + 3) & ~(size_t)3) * sizeof(float);
}

// number of splits that own timesteps at position `pos` (uniform over the grid)
__host__ __device__ static inline int attn_split_len(int nT, int NS, int ts_shift = KH_ATTN_TS_SHIFT_MAX) {
  int TS = (nT + NS - 1) / NS;
  TS = (TS + 63) & ~63;
  return TS < (1 << ts_shift) ? (1 << ts_shift) : TS;
}
// Split length TS and number of splits that own timesteps (uniform over the grid) at position `pos`:
//   pos + 1 <= 256 (and a quantum below 256): one split - nothing to merge inside the headline window;
//   up to NS quanta: every split is one quantum long - shifts, no division (this runs on the device too, between
//   the arrival of the position and the first address of the attention launch / of k_wo_comb's staging);
//   beyond: NS splits of ceil((pos + 1) / NS) timesteps rounded up to 64.
__host__ __device__ static inline void attn_split_geometry(int pos, int NS, int ts_shift, int& TS, int& nact) {
  const int nT = pos + 1;
  if (ts_shift < KH_ATTN_TS_SHIFT_MAX && nT <= KH_ATTN_MIN_TS) {
    TS = KH_ATTN_MIN_TS;
    nact = 1;
  } else if (nT <= (NS << ts_shift)) {
    TS = 1 << ts_shift;
    nact = (nT + TS - 1) >> ts_shift;
  } else {
    TS = attn_split_len(nT, NS, ts_shift);
    nact = (nT + TS - 1) / TS;
  }
}
__host__ __device__ static inline int attn_active_splits(int pos, int NS, int ts_shift = KH_ATTN_TS_SHIFT_MAX) {
  int TS, nact;
  attn_split_geometry(pos, NS, ts_shift, TS, nact);
  return nact;
}

struct AttnSplitWs {
  int* cnt;        // [heads]   arrival tickets, zero between launches
  float* ml;       // [heads, NS, 2]
  float* o;        // [heads, NS, hs]
};
__host__ __device__ static inline AttnSplitWs attn_ws_carve(void* ws, int heads, int head_size,
                                                             int ns) {
  AttnSplitWs w;
  w.cnt = (int*)ws;
  w.ml = (float*)(w.cnt + attn_ws_cnt_words(heads));
  w.o = w.ml + attn_ws_ml_words(heads, ns);
  (void)head_size;
  return w;
}

// Attention of head h over timesteps [t_begin, t_end): leaves, for threads tid < hs, the
// unnormalised output r = sum_t exp(s_t - M) v_t[tid] and L = sum_t exp(s_t - M); returns M.
template <int G>
__device__ __forceinline__ float attn_fast_partial(const float* q_h, const float* k_base,
                                                   const float* v_base, int kv_stride, int hs,
                                                   int t_begin, int t_end, float* smem,
                                                   float& r_out, float& L_out) {
  static_assert(G == 16 || G == 32 || G == 64, "G must be 16, 32 or 64");
  const int TPI = kh_wg() / G;
  float* red = smem;          // [8]
  float* lpart = smem + 8;    // [8]
  float* opart = smem + 16;   // [4][hs]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tg = tid / G, dl = tid - tg * G;
  const int hs4 = hs >> 2;
  const bool active = dl < hs4;
  const int dlc = active ? dl : 0;  // clamped lane offset: loads are unconditional, masked at the dot product
  const int stride4 = kv_stride >> 2;
  const f32x4* K4 = (const f32x4*)k_base;
  const f32x4* V4 = (const f32x4*)v_base;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float scale = 1.0f / sqrtf((float)hs);
  f32x4 q4 = zero4;
  float m = -INFINITY, l = 0.f;
  f32x4 o = zero4;
  // one batch = KH_ATTN_UB timesteps per lane group (K and V rows of all of them requested together)
  auto load_batch = [&](f32x4 (&kv)[KH_ATTN_UB], f32x4 (&vv)[KH_ATTN_UB], int tb) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < KH_ATTN_UB; ++u) {
      const int t = tb + u * TPI;
      const int tt = t < t_end ? t : t_end - 1;  // clamped address, masked in use_batch
      // unconditional (lanes past the head vector read element 0 and are dropped in use_batch): a load behind an
      // exec-mask branch is a CFG join, and the s_waitcnt at the loop header then degrades to vmcnt(0)
      kv[u] = K4[(size_t)tt * stride4 + dlc];
      vv[u] = V4[(size_t)tt * stride4 + dlc];
    }
  };
  auto use_batch = [&](const f32x4 (&kv)[KH_ATTN_UB], const f32x4 (&vv)[KH_ATTN_UB], int tb)
                       __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < KH_ATTN_UB; ++u) {
      const int t = tb + u * TPI;
      // lanes past the head vector hold element 0's data: their product is dropped HERE, at the first use, not by
      // zeroing q after its load (that select was hoisted above the batch's remaining loads together with its wait)
      const float s = group_sum<G>(active ? fma4(q4, kv[u], 0.f) : 0.f) * scale;  // all lanes: DPP, no branch
      // Branch-free update.  A timestep past the end scores -inf: m_new = m, alpha = exp(0) = 1, p = exp(-inf) = 0, so
      // l and o come out bit-identical to skipping it (its V row is a clamped re-read, finite).  Before the first
      // valid timestep m = m_new = -inf; the exponent is then taken against 0 instead (alpha = p = 0 on l = o = 0)
      // rather than forming -inf - -inf.  Valid timesteps compute exactly the expressions of the branching form.
      // (With the update behind `if (t < t_end)` the compiler sank the batch's last V load INTO the branch, behind
      // the first wait - a second memory round trip at every headline position.)
      const float sm = t < t_end ? s : -INFINITY;
      const float m_new = fmaxf(m, sm);
      const float mref = m_new == -INFINITY ? 0.f : m_new;
      const float alpha = expf(m - mref);  // exp(-inf) = 0 on the first timestep
      const float p = expf(sm - mref);
      l = l * alpha + p;
      o.x = __builtin_fmaf(p, vv[u].x, o.x * alpha);
      o.y = __builtin_fmaf(p, vv[u].y, o.y * alpha);
      o.z = __builtin_fmaf(p, vv[u].z, o.z * alpha);
      o.w = __builtin_fmaf(p, vv[u].w, o.w * alpha);
      m = m_new;
    }
  };
  const int step = TPI * KH_ATTN_UB;
  if (t_end - t_begin <= step) {  // uniform: a single batch (every headline position) - one memory round trip
    f32x4 kv[KH_ATTN_UB], vv[KH_ATTN_UB];
    q4 = ((const f32x4*)q_h)[dlc];  // same round trip as the K/V rows
    load_batch(kv, vv, t_begin + tg);
    __builtin_amdgcn_sched_barrier(0);  // every load has left before the first instruction that waits for one
    use_batch(kv, vv, t_begin + tg);
    KH_ATTN_STAMP(2);
    KH_ATTN_STAMP(3);
    KH_ATTN_STAMP_W(1);
  } else {
    // [r4] several batches: TWO register sets, the loads of batch i+2 leave when batch i has been consumed, so
    // two batches are in flight while one is computed (the plain loop issued, waited, computed: a 256-timestep
    // split cost two full memory round trips, a 1024-timestep one eight).  Same batches in the same order per
    // lane group: bit-identical.  The trip count is uniform so the reloads sit in straight-line code (exact
    // s_waitcnt); the last two batches are consumed after the loop without reloads (clamped re-reads past the end
    // in every workgroup were 4x the real traffic at position 4094: 8.4 -> 8.8 us).
    f32x4 ka[KH_ATTN_UB], va[KH_ATTN_UB], kb[KH_ATTN_UB], vb[KH_ATTN_UB];
    q4 = ((const f32x4*)q_h)[dlc];
    load_batch(ka, va, t_begin + tg);
    load_batch(kb, vb, t_begin + tg + step);
    __builtin_amdgcn_sched_barrier(0);
    int base = t_begin + tg;
    // while a third batch exists (uniform): consume, reload.  The scheduler must not interleave a batch's reloads
    // with the other batch's arithmetic: loads moved up between the uses turned the progressive waits into
    // vmcnt(0..1) on freshly issued loads (seen in the ISA).
    for (; base - tg + 2 * step < t_end; base += 2 * step) {  // a third batch exists
      use_batch(ka, va, base);
      if (base - tg == t_begin) { KH_ATTN_STAMP(2); KH_ATTN_STAMP_W(0); }
      __builtin_amdgcn_sched_barrier(0);
      load_batch(ka, va, base + 2 * step);
      __builtin_amdgcn_sched_barrier(0);
      use_batch(kb, vb, base + step);
      __builtin_amdgcn_sched_barrier(0);
      load_batch(kb, vb, base + 3 * step);  // past the end once when the batch count is odd: clamped re-reads
      __builtin_amdgcn_sched_barrier(0);
    }
    // the last two batches, nothing left to request (a 256-timestep split is exactly this tail); with an odd count
    // the second one is the clamped batch - skipped (uniform), not computed through its masks.  (Tried: separate
    // straight-line tails for "two left" / "three left" so that nothing is requested in vain - 40 % more code in a
    // kernel whose cost is latency, slower than this at every position but the odd-count ones.)
    use_batch(ka, va, base);
    if (base - tg == t_begin) { KH_ATTN_STAMP(2); KH_ATTN_STAMP_W(0); }
    __builtin_amdgcn_sched_barrier(0);
    if (base - tg + step < t_end) use_batch(kb, vb, base + step);
    KH_ATTN_STAMP(3);
    KH_ATTN_STAMP_W(1);
  }
  // ---- merge the TPI groups: ONE barrier [r6] ---------------------------------------------------------------------
  // Every wave folds its own lane groups against the WAVE's maximum - no workgroup-wide maximum first, hence no
  // barrier (and no LDS round trip) in front of the rescale -, leaves (max_w, l_w, o_w[hs]) in LDS, and the reader
  // weighs the wave partials by c_w = exp(max_w - M).  The coefficients are computed by lanes 0..7 of every wave (one
  // expf per lane instead of eight per thread) and broadcast through SGPRs (v_readlane).  Rounds 1-5 folded in two
  // steps (workgroup maximum through LDS, barrier, rescale + wave sums, LDS, barrier, plain sums): same box as round
  // 5's library, alternating three times, the launch went 3.52 -> 3.37 us (Llama-3.2-1B), 4.93 -> 4.64 (Llama-2-7B),
  // same tokens (profiles/r6_attn_fold_ab.txt, r6_attn_fold_reader_ab.txt).
  const float mw = across_groups_max<G>(m);
  const float mwr = mw == -INFINITY ? 0.f : mw;  // a wave whose groups saw no timestep: every factor is exp(-inf) = 0
  const float f = expf(m - mwr);
  l = across_groups_sum<G>(l * f);
  o.x = across_groups_sum<G>(o.x * f);
  o.y = across_groups_sum<G>(o.y * f);
  o.z = across_groups_sum<G>(o.z * f);
  o.w = across_groups_sum<G>(o.w * f);
  if (lane < G && active) ((f32x4*)(opart + wave * hs))[dl] = o;
  if (lane == 0) {
    red[wave] = mw;
    lpart[wave] = l;
  }
  __syncthreads();
  static_assert(KH_WAVES_MAX == 8, "red[] is read as two float4");
  const int nw = kh_nwaves();
  const f32x4 ra = ((const f32x4*)red)[0], rb = ((const f32x4*)red)[1];
  const float mv[KH_WAVES_MAX] = {ra.x, ra.y, ra.z, ra.w, rb.x, rb.y, rb.z, rb.w};
  float M = mv[0], mine = mv[0];
#pragma unroll
  for (int w = 1; w < KH_WAVES_MAX; ++w) {
    M = fmaxf(M, w < nw ? mv[w] : -INFINITY);  // words of absent waves were never written
    mine = lane == w ? mv[w] : mine;
  }
  const float cl = lane < nw ? expf(mine - M) : 0.f;  // M is finite: every split owns a valid timestep
  // Branch-free reader: every thread reads (threads past the head vector a clamped element; their r is never used),
  // so the sixteen LDS reads leave together.  (With the reads behind `if (tid < hs)` inside the unrolled loop the
  // compiler built eight exec-masked regions, each with its own s_waitcnt lgkmcnt(0): eight LDS round trips in a row,
  // +0.3 us on the launch - seen in the ISA after a same-box run against round 5 read 3.68 us where the experiment
  // build of this fold had read 3.36.)
  const int te = tid < hs ? tid : 0;
  const f32x4 la = ((const f32x4*)lpart)[0], lb = ((const f32x4*)lpart)[1];
  const float lv[KH_WAVES_MAX] = {la.x, la.y, la.z, la.w, lb.x, lb.y, lb.z, lb.w};
  float ov[KH_WAVES_MAX];
#pragma unroll
  for (int w = 0; w < KH_WAVES_MAX; ++w) ov[w] = opart[(w < nw ? w : 0) * hs + te];
  float r = 0.f, L = 0.f;
#pragma unroll
  for (int w = 0; w < KH_WAVES_MAX; ++w) {
    const float c = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(cl), w));  // 0 for an absent wave
    r = __builtin_fmaf(c, ov[w], r);
    L = __builtin_fmaf(c, w < nw ? lv[w] : 0.f, L);  // words of absent waves were never written
  }
  r_out = r;
  L_out = L;
  return M;
}

// Merge of the split partials in the LAST ARRIVER (all of its threads call this; uniform).  The partials were
// written by other workgroups: agent-scope relaxed atomic loads (vector path to L2, never a stale L1 / scalar-cache
// line).  NH heads per workgroup (1, or the KV group's kv_mul), first head h0; thread (j, e) - `mine` - returns
// element e of head h0 + j.
// [r5] (profiles/r5_attn_phases.txt: the merge of 32 splits took 2.9 us of a 12-us launch, of 16 splits 1.8)
//  * ONE memory round trip: every load of the merge is requested before the first is used.  Rounds 3-4 ran two loops
//    - the common maximum, then the weighted sums - of 16 splits per batch; agent-scope loads are not reordered by
//    the compiler, so 32 splits paid four dependent trips;
//  * the coefficient exp(M_k - max M) of split k is computed ONCE per (head, split) - thread j * MAXS + k loads
//    (M, L), the maximum and the coefficients go through LDS - instead of by each of the head's hs threads for each
//    of the nact splits (32 expf in sequence per thread were half of those 2.9 us).
// Same terms in the same order as before and as kh_fused.h::CombStager (the deferred merge): max over the splits,
// expf(M_k - max), one fma per split into num and den, ascending k.
// lds: 3 * NH * MAXS floats, not overlapping the ticket word.  Requires NH * MAXS <= workgroup width.
#define KH_ATTN_MB 16
__device__ __forceinline__ float2 ld_agent_f2(const float* p) {  // p 8-byte aligned
  const unsigned long long w =
      __hip_atomic_load((const unsigned long long*)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  float2 r;
  r.x = __uint_as_float((unsigned)(w & 0xffffffffull));
  r.y = __uint_as_float((unsigned)(w >> 32));
  return r;
}
static inline size_t attn_merge_lds_floats(int nh, int maxs) { return 16 + (size_t)3 * nh * maxs; }
template <int NH, int MAXS, int MB>
__device__ __forceinline__ float attn_merge_lds(const AttnSplitWs& ws, int h0, int j, int e, bool mine, int hs,
                                                int nact, int NSW, float* lds) {
  static_assert((MAXS & (MAXS - 1)) == 0 && MB <= MAXS, "split slots per head in LDS: a power of two");
  float* Ms = lds;
  float* Ls = lds + NH * MAXS;
  float* Cf = Ls + NH * MAXS;
  const int tid = threadIdx.x;
  float ov[MB];
  if (mine) {
    const size_t base = (size_t)(h0 + j) * NSW;
#pragma unroll
    for (int u = 0; u < MB; ++u)  // clamped: a re-read of the last split, dropped below
      ov[u] = __hip_atomic_load(&ws.o[(base + (u < nact ? u : nact - 1)) * hs + e], __ATOMIC_RELAXED,
                                __HIP_MEMORY_SCOPE_AGENT);
  }
  if (tid < NH * MAXS) {
    const int jj = tid / MAXS, kk = tid % MAXS;
    float2 ml;
    ml.x = -INFINITY;
    ml.y = 0.f;
    if (kk < nact) ml = ld_agent_f2(&ws.ml[((size_t)(h0 + jj) * NSW + kk) * 2]);
    Ms[tid] = ml.x;
    Ls[tid] = ml.y;
  }
  __syncthreads();
  if (tid < NH * MAXS) {
    const int jj = tid / MAXS;
    float Mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < MAXS; ++k) Mx = fmaxf(Mx, Ms[jj * MAXS + k]);
    Cf[tid] = expf(Ms[tid] - Mx);  // slots past nact: exp(-inf) = 0, never read
  }
  __syncthreads();
  float num = 0.f, den = 0.f;
  if (mine) {
    // ascending k, one term per split.  No `if (u < nact)` around a term [r6]: a slot past nact carries f = 0 and
    // L = 0 (written above) and a clamped, finite o, so its term adds an exact zero - and without the (uniform)
    // branches the 2 MB LDS reads leave together instead of one round trip per split (seen in the ISA: a chain of
    // ds_read + s_waitcnt lgkmcnt(0) pairs, the bulk of the merge's 1.6-2.1 us at 16-32 splits)
#pragma unroll
    for (int u = 0; u < MB; ++u) {
      const float f = Cf[j * MAXS + u];
      num = __builtin_fmaf(ov[u], f, num);
      den = __builtin_fmaf(Ls[j * MAXS + u], f, den);
    }
  }
  return num / den;
}
template <int NH, int MAXS>
__device__ __forceinline__ float attn_merge_splits(const AttnSplitWs& ws, int h0, int j, int e, bool mine, int hs,
                                                   int nact, int NSW, float* lds) {
  static_assert(MAXS == KH_ATTN_MB || MAXS == 2 * KH_ATTN_MB, "merge batch sizes");
  if (MAXS == KH_ATTN_MB || nact <= KH_ATTN_MB)  // uniform
    return attn_merge_lds<NH, MAXS, KH_ATTN_MB>(ws, h0, j, e, mine, hs, nact, NSW, lds);
  return attn_merge_lds<NH, MAXS, MAXS>(ws, h0, j, e, mine, hs, nact, NSW, lds);  // the group path's 17 ... 32
}

// agent-scope relaxed store = global_store ... sc1 (write-through): visible to every XCD once the
// issuing wave's vmcnt has drained, without a buffer_wbl2
__device__ __forceinline__ void st_agent(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Publication of a split's partial and the hand-over to the last arriver, two forms:
//  * default - the guide's G16 R1 form ("sc1 stores AND sc1 loads", MI355X_MICROARCH.md, correctness boundaries):
//    every storing wave drains its write-through stores (s_waitcnt vmcnt(0)), barrier, one lane takes a relaxed
//    agent-scope ticket; the last arriver reads the partials with agent-scope (sc1) loads, which are served past
//    its L1.  No buffer_wbl2, no buffer_inv: 1-2 us less on the 4 k-32 k positions.  It relies on
//    gfx950 behaviour that the HIP memory model does not spell out: relaxed agent stores are write-through,
//    vmcnt(0) means they are acknowledged at the device coherence point, and an sc1 load is not served from a
//    stale line.  tests/test_ops_gpu.py::test_mha_decode_split_merge_stress hammers it (uneven load, every XCD,
//    warm readers, thousands of launches, every word checked);
//  * fenced - KH_FLAG_ATTN_MERGE_FENCED / hook KH_ATTN_FENCED=1: the formally ordered form, release fence before
//    the ticket and acquire fence in the last arriver (buffer_wbl2 sc1 / buffer_inv sc1), for a part, a compiler or
//    a partition mode where the above has not been verified.
__device__ __forceinline__ void attn_publish_barrier(bool fenced) {
  if (fenced) {
    // every storing wave drains its stores here too: a workgroup-scope barrier need not wait for vmcnt, and the
    // release fence below is issued by one lane only - with the wait the fenced form is a strict superset of the default
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the compiler may drop the wait behind buffer_wbl2 (guide G16)
    }
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its stores
    __syncthreads();
  }
}
__device__ __forceinline__ void attn_acquire(bool fenced) {  // in the last arriver, before the merge (uniform)
  if (fenced) {
    if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
  }
}

// One workgroup = (head h, split s) of a grid of heads*NS workgroups.  ws may be null iff NS==1.
// defer: leave the partial in the workspace and return, whatever nact is (the consumer combines).
// Returns true in the workgroup that wrote the head's final output.
template <int G>
__device__ __forceinline__ bool attn_head_decode_fast(const float* q_h, const float* k_base,
                                                      const float* v_base, int kv_stride, int hs,
                                                      int pos, float* out_h, float* smem, int h,
                                                      int s, int NS, AttnSplitWs ws, int NSW = 0,
                                                      bool defer = false, bool fenced = false,
                                                      int ts_shift = KH_ATTN_TS_SHIFT_MAX) {
  if (NSW <= 0) NSW = NS;  // slot stride of the workspace (>= NS)
  const int tid = threadIdx.x;
  const int nT = pos + 1;
  int TS, nact;
  attn_split_geometry(pos, NS, ts_shift, TS, nact);  // uniform over the grid; shifts only at the positions that matter
  if (s >= nact) return false;
  KH_ATTN_STAMP(1);
  const int t_begin = s * TS;
  const int t_end = t_begin + TS < nT ? t_begin + TS : nT;
  float r, L;
  const float M = attn_fast_partial<G>(q_h, k_base, v_base, kv_stride, hs, t_begin, t_end, smem, r, L);
  KH_ATTN_STAMP(4);
  const size_t slot = (size_t)h * NSW + s;
  if (defer) {  // plain stores (also with ONE active split); the next kernel on the stream reads them
    if (tid < hs) ws.o[slot * hs + tid] = r;
    if (tid == 0) {
      ws.ml[slot * 2] = M;
      ws.ml[slot * 2 + 1] = L;
    }
    KH_ATTN_STAMP(5);
    return false;
  }
  if (nact == 1) {
    if (tid < hs) out_h[tid] = r / L;
    KH_ATTN_STAMP(5);
    return true;
  }
  // ---- publish this split's partial (write-through), take a ticket ------------------------------
  if (tid < hs) st_agent(&ws.o[slot * hs + tid], r);
  if (tid == 0) {
    st_agent(&ws.ml[slot * 2], M);
    st_agent(&ws.ml[slot * 2 + 1], L);
  }
  attn_publish_barrier(fenced);
  KH_ATTN_STAMP(5);
  int* flag = (int*)smem;  // red[] is free again after the barriers inside attn_fast_partial
  if (tid == 0)
    flag[0] = __hip_atomic_fetch_add(&ws.cnt[h], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  KH_ATTN_STAMP(6);
  if (flag[0] != nact - 1) return false;  // not the last arriver
  attn_acquire(fenced);
  // ---- last arriver: merge every split's partial (agent-scope loads inside) ------------------------
  {
    const float v = attn_merge_splits<1, KH_ATTN_MAX_NS>(ws, h, 0, tid < hs ? tid : 0, tid < hs, hs, nact, NSW, smem + 16);
    if (tid < hs) out_h[tid] = v;
  }
  if (tid == 0) __hip_atomic_store(&ws.cnt[h], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  KH_ATTN_STAMP(7);
  return true;
}


// =============================================================================================
// GQA long-context core.  One workgroup = (kv group g, time split s) computes ALL KVM query
// heads of the group from ONE pass over the group's K/V rows: at long contexts the per-head
// kernel re-reads every K/V row kv_mul times (from L2 at best), which is what bounds it
// (2.6 TB/s of K/V bytes at pos 131071 on Llama-3.2-1B); here every K/V byte crosses the memory
// system once.  The cost is KVM times the arithmetic per loaded byte, so the softmax runs in the
// log2 domain on the hardware exp2 (v_exp_f32) and the rescale factor is computed once per
// batch of KH_ATTN_UB timesteps instead of once per timestep.  Used only when pos + 1 >=
// t_long (short contexts are latency-bound and prefer one workgroup per head).
// smem: red[KH_WAVES_MAX*KVM] | lpart[KH_WAVES_MAX*KVM] | opart[KH_WAVES_MAX*KVM*hs]
static inline size_t attn_group_lds_bytes(int head_size, int kvm) {
  return (size_t)KH_WAVES_MAX * kvm * (2 + head_size) * sizeof(float);
}
// splits per KV group carried by the grid: enough workgroups to cover every CU twice
static inline int attn_group_splits(int cache_len, int kv_heads) {
  int ns = (cache_len + (1 << KH_ATTN_TSG_SHIFT) - 1) >> KH_ATTN_TSG_SHIFT;
  int want = (512 + kv_heads - 1) / kv_heads;
  if (want > KH_ATTN_MAX_NS_G) want = KH_ATTN_MAX_NS_G;
  if (ns > want) ns = want;
  return ns < 1 ? 1 : ns;
}

template <int G, int KVM>
__device__ __forceinline__ void attn_group_partial(const float* q_g, const float* k_base,
                                                   const float* v_base, int kv_stride, int hs,
                                                   int t_begin, int t_end, float* smem,
                                                   float& r_out, float& L_out, float& M_out) {
  static_assert(G == 16 || G == 32 || G == 64, "G must be 16, 32 or 64");
  const int TPI = kh_wg() / G;
  const int nw = kh_nwaves();
  float* red = smem;
  float* lpart = red + KH_WAVES_MAX * KVM;
  float* opart = lpart + KH_WAVES_MAX * KVM;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tg = tid / G, dl = tid - tg * G;
  const int hs4 = hs >> 2;
  const bool active = dl < hs4;
  const int dlc = active ? dl : 0;  // clamped lane offset: loads are unconditional, masked at the dot product
  const int stride4 = kv_stride >> 2;
  const f32x4* K4 = (const f32x4*)k_base;
  const f32x4* V4 = (const f32x4*)v_base;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  const float sc2 = 1.4426950408889634f / sqrtf((float)hs);  // scores in the log2 domain
  f32x4 q4[KVM], o[KVM];
  float m[KVM], l[KVM];
#pragma unroll
  for (int j = 0; j < KVM; ++j) {
    q4[j] = ((const f32x4*)(q_g + (size_t)j * hs))[dlc];  // lanes past the head vector: dropped at the dot product
    o[j] = zero4;
    m[j] = -INFINITY;
    l[j] = 0.f;
  }
  auto load_batch = [&](f32x4 (&kv)[KH_ATTN_UB], f32x4 (&vv)[KH_ATTN_UB], int tb) __attribute__((always_inline)) {
#pragma unroll
    for (int u = 0; u < KH_ATTN_UB; ++u) {
      const int t = tb + u * TPI;
      const int tt = t < t_end ? t : t_end - 1;
      kv[u] = ld_nt(K4 + (size_t)tt * stride4 + dlc);  // unconditional, see attn_fast_partial
      vv[u] = ld_nt(V4 + (size_t)tt * stride4 + dlc);
    }
  };
  // tb < t_end required (u = 0 valid): a batch without any timestep would form exp2(-inf - -inf)
  auto use_batch = [&](const f32x4 (&kv)[KH_ATTN_UB], const f32x4 (&vv)[KH_ATTN_UB], int tb)
                       __attribute__((always_inline)) {
#pragma unroll
    for (int j = 0; j < KVM; ++j) {
      float sv[KH_ATTN_UB];
      float mx = m[j];
#pragma unroll
      for (int u = 0; u < KH_ATTN_UB; ++u) {
        sv[u] = group_sum<G>(active ? fma4(q4[j], kv[u], 0.f) : 0.f) * sc2;  // all lanes: DPP, no branch
        if (tb + u * TPI < t_end) mx = fmaxf(mx, sv[u]);       // u = 0 is always valid
      }
      const float alpha = __builtin_amdgcn_exp2f(m[j] - mx);  // exp2(-inf) = 0 on the first batch
      float lj = l[j] * alpha;
      f32x4 oj = o[j] * alpha;
#pragma unroll
      for (int u = 0; u < KH_ATTN_UB; ++u) {
        // branch-free: a timestep past the end contributes p = 0 (its V row is a clamped re-read, finite), which
        // leaves lj and oj bit-identical to skipping it; mx is finite here (u = 0 is valid)
        const float p = tb + u * TPI < t_end ? __builtin_amdgcn_exp2f(sv[u] - mx) : 0.f;
        lj += p;
        oj.x = __builtin_fmaf(p, vv[u].x, oj.x);
        oj.y = __builtin_fmaf(p, vv[u].y, oj.y);
        oj.z = __builtin_fmaf(p, vv[u].z, oj.z);
        oj.w = __builtin_fmaf(p, vv[u].w, oj.w);
      }
      l[j] = lj;
      o[j] = oj;
      m[j] = mx;
    }
  };
  // [r4] two register sets, see attn_fast_partial: the loads of batch i+2 leave when batch i has been consumed
  // (this path starts at 4096 timesteps, i.e. always has many batches per workgroup); uniform trip count.
  {
    const int step = TPI * KH_ATTN_UB;
    f32x4 ka[KH_ATTN_UB], va[KH_ATTN_UB], kb[KH_ATTN_UB], vb[KH_ATTN_UB];
    load_batch(ka, va, t_begin + tg);
    load_batch(kb, vb, t_begin + tg + step);
    __builtin_amdgcn_sched_barrier(0);
    int base = t_begin + tg;
    for (; base - tg + 2 * step < t_end; base += 2 * step) {  // while a third batch exists (uniform)
      if (base < t_end) use_batch(ka, va, base);
      if (base - tg == t_begin) { KH_ATTN_STAMP(2); KH_ATTN_STAMP_W(0); }
      __builtin_amdgcn_sched_barrier(0);
      load_batch(ka, va, base + 2 * step);
      __builtin_amdgcn_sched_barrier(0);
      if (base + step < t_end) use_batch(kb, vb, base + step);
      __builtin_amdgcn_sched_barrier(0);
      load_batch(kb, vb, base + 3 * step);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (base < t_end) use_batch(ka, va, base);  // the last two batches: nothing left to request
    if (base - tg == t_begin) { KH_ATTN_STAMP(2); KH_ATTN_STAMP_W(0); }
    __builtin_amdgcn_sched_barrier(0);
    if (base + step < t_end) use_batch(kb, vb, base + step);
    KH_ATTN_STAMP(3);
    KH_ATTN_STAMP_W(1);
  }
  // ---- merge the lane groups of a wave, then the waves: ONE barrier [r6], as attn_fast_partial -------------------
  // every wave folds its lane groups against its own maximum, (max, l, o) of the wave go to LDS, the reader weighs
  // the wave partials by exp2(max_w - M) (the hardware exp2: eight per thread cost less than the barrier + LDS round
  // trip of the two-step fold of rounds 1-5)
#pragma unroll
  for (int j = 0; j < KVM; ++j) {
    const float mw = across_groups_max<G>(m[j]);
    const float mwr = mw == -INFINITY ? 0.f : mw;  // a wave without a timestep: every factor exp2(-inf) = 0
    const float f = __builtin_amdgcn_exp2f(m[j] - mwr);
    const float lw = across_groups_sum<G>(l[j] * f);
    f32x4 ow;
    ow.x = across_groups_sum<G>(o[j].x * f);
    ow.y = across_groups_sum<G>(o[j].y * f);
    ow.z = across_groups_sum<G>(o[j].z * f);
    ow.w = across_groups_sum<G>(o[j].w * f);
    if (lane < G && active) ((f32x4*)(opart + (size_t)(wave * KVM + j) * hs))[dl] = ow;
    if (lane == 0) {
      red[wave * KVM + j] = mw;
      lpart[wave * KVM + j] = lw;
    }
  }
  __syncthreads();
  // branch-free reader (threads past the group's KVM * hs outputs read head 0, element 0; their results are never
  // used): all 24 LDS reads leave before the first is consumed - behind a mask the compiler made three round trips of
  // them, and inside an unrolled loop it makes one per iteration (see attn_fast_partial)
  float r = 0.f, L = 0.f, M2 = -INFINITY;
  {
    const bool mine = tid < KVM * hs;
    const int j = mine ? tid / hs : 0, e = mine ? tid - j * hs : 0;
    float mv[KH_WAVES_MAX], lv[KH_WAVES_MAX], ov[KH_WAVES_MAX];
#pragma unroll
    for (int w = 0; w < KH_WAVES_MAX; ++w) {
      const int wc = w < nw ? w : 0;  // words of absent waves were never written
      mv[w] = red[wc * KVM + j];
      lv[w] = lpart[wc * KVM + j];
      ov[w] = opart[(size_t)(wc * KVM + j) * hs + e];
    }
#pragma unroll
    for (int w = 0; w < KH_WAVES_MAX; ++w) {
      mv[w] = w < nw ? mv[w] : -INFINITY;
      M2 = fmaxf(M2, mv[w]);
    }
    // M2 is finite: every split owns a valid timestep (use_batch's precondition)
#pragma unroll
    for (int w = 0; w < KH_WAVES_MAX; ++w) {
      const float c = __builtin_amdgcn_exp2f(mv[w] - M2);  // 0 for a wave without a timestep / an absent wave
      r = __builtin_fmaf(c, ov[w], r);
      L = __builtin_fmaf(c, lv[w], L);
    }
    if (!mine) M2 = -INFINITY;
  }
  r_out = r;
  L_out = L;
  M_out = M2 * 0.6931471805599453f;  // back to the natural-log domain of the split merge
  KH_ATTN_STAMP(4);
}

// One workgroup = (kv group g, split s); heads g*KVM .. g*KVM+KVM-1.  Workspace slots and the
// merge are those of the per-head path (slot = h*NSW + s); the arrival ticket of the group is
// the ticket of its first head.  Requires KVM*hs <= blockDim.x.
template <int G, int KVM>
__device__ __forceinline__ void attn_group_decode(const float* q_g, const float* k_base,
                                                  const float* v_base, int kv_stride, int hs,
                                                  int pos, float* out_g, float* smem, int g, int s,
                                                  int NS, int NSW, AttnSplitWs ws, bool fenced) {
  const int tid = threadIdx.x;
  const int nT = pos + 1;
  int TS, nact;
  attn_split_geometry(pos, NS, KH_ATTN_TSG_SHIFT, TS, nact);  // uniform over the grid
  if (s >= nact) return;
  KH_ATTN_STAMP(1);
  const int t_begin = s * TS;
  const int t_end = t_begin + TS < nT ? t_begin + TS : nT;
  float r, L, M;
  attn_group_partial<G, KVM>(q_g, k_base, v_base, kv_stride, hs, t_begin, t_end, smem, r, L, M);
  const bool mine = tid < KVM * hs;
  const int j = mine ? tid / hs : 0, e = tid - j * hs;
  const int h = g * KVM + j;
  if (nact == 1) {
    if (mine) out_g[(size_t)j * hs + e] = r / L;
    return;
  }
  const size_t slot = (size_t)h * NSW + s;
  if (mine) {  // write-through publication, see attn_head_decode_fast
    st_agent(&ws.o[slot * hs + e], r);
    if (e == 0) {
      st_agent(&ws.ml[slot * 2], M);
      st_agent(&ws.ml[slot * 2 + 1], L);
    }
  }
  attn_publish_barrier(fenced);
  KH_ATTN_STAMP(5);
  int* flag = (int*)smem;
  if (tid == 0)
    flag[0] = __hip_atomic_fetch_add(&ws.cnt[g * KVM], 1, __ATOMIC_RELAXED,
                                     __HIP_MEMORY_SCOPE_AGENT);
  __syncthreads();
  KH_ATTN_STAMP(6);
  if (flag[0] != nact - 1) return;  // not the last arriver
  attn_acquire(fenced);
  {
    const float v = attn_merge_splits<KVM, KH_ATTN_MAX_NS_G>(ws, g * KVM, j, mine ? e : 0, mine, hs, nact, NSW, smem + 16);
    if (mine) out_g[(size_t)j * hs + e] = v;
  }
  if (tid == 0)
    __hip_atomic_store(&ws.cnt[g * KVM], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  KH_ATTN_STAMP(7);
}

// =============================================================================================
// The decode-attention launch shared by the fused step (kh_model_step.hip) and the operator-level
// entry point kh_mha_decode_f32 (kh_ops.hip).
struct KhAttnArgs {
  const float* q;          // [dim]
  const float* kcache_layer;
  const float* vcache_layer;
  float* out;              // [dim]
  const int32_t* d_pos;
  int kv_dim, kv_mul, head_size;
  int kv_heads, nsplit;    // per-head path: kv_heads * kv_mul * nsplit workgroups
  void* ws;                // attn_ws_bytes(heads, head_size, ws_stride), tickets zeroed
  int ws_stride;           // split slots per head in the workspace (>= nsplit, >= nsplit_g)
  int nsplit_g;            // GQA long-context path: kv_heads * nsplit_g workgroups (0 = off)
  int t_long;              // the group path runs when pos + 1 >= t_long
  int ts_shift;            // per-head path: log2 of the split quantum (attn_ts_shift_for)
  int defer;               // per-head path: leave the split partials (also a single one) for k_wo_comb
  int fenced;              // in-launch merge with release / acquire fences (see attn_publish_barrier)
  // prefill (kh_prefill.h): gridDim.y tokens per launch, token t at position pos + t, its q /
  // out rows tok_stride floats apart, its split workspace ws_tok_bytes apart (decode: y = 1)
  int tok_stride;
  size_t ws_tok_bytes;
  int kvh_shift, kvm_shift;  // log2 of kv_heads / kv_mul when they are powers of two, else -1 (set by launch_attn_decode)
};

// per-head path: block -> (kv group g, head-in-group j, split s).  Blocks are placed on XCD
// b % 8, so with g = b % kv_heads the kv_mul heads that share K/V rows share an XCD's L2.
template <int G>
__device__ __forceinline__ void attn_head_block(const KhAttnArgs& a, float* smem, int b, int pos) {
  // power-of-two head counts (every Llama-family config): shifts instead of three integer divisions (~40
  // instructions each, in front of the first address of a launch whose whole body is ~2 us)
  int g, j, s;
  if (a.kvh_shift >= 0 && a.kvm_shift >= 0) {  // uniform
    g = b & (a.kv_heads - 1);
    j = (b >> a.kvh_shift) & (a.kv_mul - 1);
    s = b >> (a.kvh_shift + a.kvm_shift);
  } else {
    g = b % a.kv_heads;
    j = (b / a.kv_heads) % a.kv_mul;
    s = b / (a.kv_heads * a.kv_mul);
  }
  const int h = g * a.kv_mul + j;
  const size_t head_off = (size_t)g * a.head_size;
  attn_head_decode_fast<G>(
      a.q + (size_t)h * a.head_size, a.kcache_layer + head_off, a.vcache_layer + head_off,
      a.kv_dim, a.head_size, pos, a.out + (size_t)h * a.head_size, smem, h, s, a.nsplit,
      attn_ws_carve(a.ws, a.kv_heads * a.kv_mul, a.head_size, a.ws_stride), a.ws_stride, a.defer != 0,
      a.fenced != 0, a.ts_shift);
}

// KVM = 0: per-head workgroups only.  KVM = kv_mul > 1: per-head workgroups at short contexts,
// one workgroup per (kv group, split) computing the group's KVM heads from one pass over K/V
// once pos + 1 >= t_long.  The choice is uniform over the grid (it depends on the position
// only), so one captured launch serves every position; workgroups beyond the active path's
// count leave immediately.
// (Measured and not kept, profiles/r4_attn_pipe_ab.txt: the split-0 workgroups requesting q and their first batch -
// rows tg + u * TPI whatever the position is - BEFORE the device scalar *d_pos has arrived.  The position then has to
// come through a vector load (scalar loads return out of order, so the kernel-argument waits would wait for it), the
// rows past the position are cold HBM rows instead of clamped re-reads, and the wait covers both: 3.9 -> 4.4-4.6 us
// at position 63.)
template <int G, int KVM>
__global__ __launch_bounds__(KH_WG_MAX) void k_attn_decode(KhAttnArgs a, int host_pos) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  KH_ATTN_STAMP(0);
  int pos = a.d_pos ? *a.d_pos : host_pos;
  if (gridDim.y > 1) {  // uniform: one grid slice per prompt token
    const int t = (int)blockIdx.y;
    pos += t;
    a.q += (size_t)t * a.tok_stride;
    a.out += (size_t)t * a.tok_stride;
    a.ws = (char*)a.ws + (size_t)t * a.ws_tok_bytes;
  }
  const int b = (int)blockIdx.x;
  if (KVM == 0 || pos + 1 < a.t_long) {
    if (b < a.kv_heads * a.kv_mul * a.nsplit) attn_head_block<G>(a, (float*)smem_raw, b, pos);
    return;
  }
  if constexpr (KVM > 0) {
    if (b >= a.kv_heads * a.nsplit_g) return;
    const int g = b % a.kv_heads, s = b / a.kv_heads;
    const size_t head_off = (size_t)g * a.head_size;
    attn_group_decode<G, KVM>(a.q + (size_t)g * KVM * a.head_size, a.kcache_layer + head_off,
                              a.vcache_layer + head_off, a.kv_dim, a.head_size, pos,
                              a.out + (size_t)g * KVM * a.head_size, (float*)smem_raw, g, s,
                              a.nsplit_g, a.ws_stride,
                              attn_ws_carve(a.ws, a.kv_heads * KVM, a.head_size, a.ws_stride), a.fenced != 0);
  }
}

// lanes cooperating on one timestep for this head size
static inline int attn_lanes(int head_size) {
  int G = 1;
  while (G < head_size / 4) G <<= 1;
  return G < 16 ? 16 : G;
}
// is the GQA group path instantiated for this geometry (and does it fit a `wg`-thread workgroup)?
static inline bool attn_group_supported(int head_size, int kv_mul, int wg) {
  const int G = attn_lanes(head_size);
  if (kv_mul * head_size > wg) return false;
  if (kv_mul * KH_ATTN_MAX_NS_G > wg) return false;  // attn_merge_lds: one thread per (head, split slot)
  if (G == 16) return kv_mul == 2 || kv_mul == 4 || kv_mul == 7 || kv_mul == 8;
  if (G == 32) return kv_mul == 2 || kv_mul == 4;
  return false;
}
// Split geometry of the decode launch for a cache of `cache_len` rows: the ONE place that decides it
// (model level: kh_model_load.hip::finish_create; operator level: kh_mha_decode_f32).
//   ns      time splits per head on the per-head path (1 for head_size <= 32: generic kernel)
//   ns_g    splits per KV group on the GQA group path, 0 = path off
//   stride  split slots per head in the workspace
//   t_long  the group path runs when pos + 1 >= t_long
// t_long_override < 0: default policy - models with few KV heads (Qwen2.5-0.5B: 2) cannot fill the chip
// with (group, split) workgroups and stay per-head; 0 = never; > 0 = that threshold.
struct AttnPlan {
  int ns, ns_g, stride, t_long, ts_shift;
};
static inline AttnPlan attn_plan(int head_num, int kv_mul, int head_size, int cache_len, int wg,
                                 int t_long_override) {
  AttnPlan p;
  p.ts_shift = attn_ts_shift_for(head_size);
  p.ns = head_size > 32 ? attn_num_splits(cache_len, p.ts_shift) : 1;
  p.ns_g = 0;
  p.stride = p.ns;
  p.t_long = 1 << 30;
  if (kv_mul > 1 && head_size > 32 && head_num % kv_mul == 0 &&
      attn_group_supported(head_size, kv_mul, wg)) {
    int tl = head_num / kv_mul >= KH_ATTN_MIN_GROUPS ? KH_ATTN_TLONG_DEFAULT : 0;
    if (t_long_override >= 0) tl = t_long_override;
    if (tl > 0 && tl <= cache_len) {
      p.ns_g = attn_group_splits(cache_len, head_num / kv_mul);
      p.t_long = tl;
      if (p.ns_g > p.stride) p.stride = p.ns_g;
    }
  }
  return p;
}
// Deferred merge (defer mode above): can the consumer's staging combine the partials?  The combiner
// (kh_fused.h::CombStager) keeps wo's input vector in registers (<= 16 floats per thread of a `wg`-wide
// workgroup), computes the heads * KH_ATTN_MAX_NS merge coefficients in <= 4 passes of the workgroup and
// maps a float4 of the vector to its head with a shift (power-of-two head sizes).
#define KH_COMB_CP 4
static inline bool comb_supported(int dim, int heads, int head_size, int wg) {
  return head_size > 32 && (head_size & (head_size - 1)) == 0 && dim == heads * head_size && dim <= 16 * wg &&
         heads * KH_ATTN_MAX_NS <= KH_COMB_CP * wg;
}
// KH_ATTN_TLONG hook -> t_long_override
static inline int attn_tlong_hook() {
  const char* e = khm::dbg("KH_ATTN_TLONG");
  return e ? atoi(e) : -1;
}

// Launch.  a.nsplit_g == 0 disables the group path; head_size > 32 required (callers route
// smaller heads to the generic LDS-score kernel).
// pos_hi >= 0 (host-positioned launches only): the highest position among the ntok tokens; the grid
// then carries only the splits that own timesteps at those positions instead of all cache_len / 256
// of them (a 128-token prefill slice at the start of a 131072-row cache launched 16x too many
// workgroups, all of which left immediately - measured 129 us per layer of dispatch).
static inline void launch_attn_decode(KhAttnArgs a, int host_pos, int wg, hipStream_t s,
                                      int ntok = 1, int pos_hi = -1) {
  const int G = attn_lanes(a.head_size);
  auto log2_or_neg = [](int v) {
    int sh = 0;
    while ((1 << sh) < v) ++sh;
    return (1 << sh) == v ? sh : -1;
  };
  a.kvh_shift = log2_or_neg(a.kv_heads);
  a.kvm_shift = log2_or_neg(a.kv_mul);
  // host-positioned launch whose positions all stay below the group path's threshold: the per-head-only
  // instantiation (fewer registers: two 512-thread workgroups per CU instead of one, which matters when 16 splits
  // x 32 heads are in flight).  Device-positioned callers clear nsplit_g themselves when they know the range.
  if (!a.d_pos && (pos_hi >= 0 ? pos_hi : host_pos) + 1 < a.t_long) a.nsplit_g = 0;
  const bool grp = a.nsplit_g > 0 && attn_group_supported(a.head_size, a.kv_mul, wg);
  if (!grp) a.nsplit_g = 0;
  int head_splits = a.nsplit, group_splits = grp ? a.nsplit_g : 0;
  if (pos_hi >= 0 && !a.d_pos) {
    head_splits = 0;
    group_splits = 0;
    for (int p = host_pos; p <= pos_hi; ++p) {
      if (grp && p + 1 >= a.t_long) {
        const int n = attn_active_splits(p, a.nsplit_g, KH_ATTN_TSG_SHIFT);
        if (n > group_splits) group_splits = n;
      } else {
        const int n = attn_active_splits(p, a.nsplit, a.ts_shift);
        if (n > head_splits) head_splits = n;
      }
    }
  }
  int grid = a.kv_heads * a.kv_mul * head_splits;
  size_t lds = attn_fast_lds_bytes(a.head_size, wg);
  if (grp) {
    if (a.kv_heads * group_splits > grid) grid = a.kv_heads * group_splits;
    const size_t l2 = attn_group_lds_bytes(a.head_size, a.kv_mul);
    if (l2 > lds) lds = l2;
  }
  if (grid < 1) grid = 1;
#define KH_ATTN_LAUNCH(GG, KK) \
  hipLaunchKernelGGL((k_attn_decode<GG, KK>), dim3(grid, ntok), dim3(wg), lds, s, a, host_pos)
  const int kvm = grp ? a.kv_mul : 0;
  if (G == 16) {
    switch (kvm) {
      case 2: KH_ATTN_LAUNCH(16, 2); break;
      case 4: KH_ATTN_LAUNCH(16, 4); break;
      case 7: KH_ATTN_LAUNCH(16, 7); break;
      case 8: KH_ATTN_LAUNCH(16, 8); break;
      default: KH_ATTN_LAUNCH(16, 0); break;
    }
  } else if (G == 32) {
    switch (kvm) {
      case 2: KH_ATTN_LAUNCH(32, 2); break;
      case 4: KH_ATTN_LAUNCH(32, 4); break;
      default: KH_ATTN_LAUNCH(32, 0); break;
    }
  } else {
    KH_ATTN_LAUNCH(64, 0);
  }
#undef KH_ATTN_LAUNCH
}
