// kh_merged.h — EXPERIMENTAL (opt-in: KH_FLAG_MERGE / KH_MERGE=1; measured neutral to slightly
// negative, see the end of this comment) — one launch for the three small, latency-bound stages
// of a layer:
//
//     [ qkv (rmsnorm + wq|wk|wv + bias + RoPE) | attention | wo + residual ]
//
// As separate kernels each stage pays ~3 us of launch ramp / tail on top of its bytes (DESIGN.md
// §5: qkv 7.8 + attn 4.8 + wo 4.8 us for 42 MB on Llama-3.2-1B).  Here the stages are consecutive
// ranges of ONE grid and hand their vectors over inside the launch (guide G16, recipe R1):
//
//   producer  sc1 (write-through) stores -> every wave s_waitcnt vmcnt(0) -> barrier -> one lane
//             relaxed agent-scope atomic add on a counter
//   consumer  issues everything that does not depend on the producer first (wo: its first weight
//             chunk and the residual), then ONE lane polls the counter (relaxed, s_sleep, bounded),
//             barrier, sc1 loads of the payload.  No fences on either side.
//
// qkv -> attention: one counter per KV group (a group's q, k and v rows are 48-96 arrivals);
// attention -> wo: one counter, head_num arrivals.  Counters are re-armed by k_sample at the end
// of every decode step.  Progress: a stage only ever waits for stages with LOWER block indices,
// which the dispatcher starts first and which never wait themselves, so waiting workgroups
// cannot starve their producers; every wait is bounded anyway and raises an error word instead
// of hanging (kh_model_generate then reports KH_ERR_SYNC).
//
// Measured (MI355X, 128-step decode): the merged launch takes 15.8 us vs 7.8 + 4.8 + 4.8 + two
// 0.3 us gaps = 18.0 us for the three kernels, yet end to end it is 1009 vs 1010 tok/s on
// Llama-3.2-1B fp32 and 501 vs 519 tok/s on Llama-2-7B int8: each hand-off (drain + memory-side
// atomic + poll period + sc1 reads) costs the ~3 us a kernel boundary costs, and the pollers
// compete with the weight stream.  First attempt, all pollers on one word: 52 us per launch —
// requests to one line serialise at the memory side (hence one 128-B line per counter, 8
// replicas of the attention->wo counter, 0.2-0.85 us poll periods).  Kept because results are
// bitwise identical (tests) and it is the scaffold for a persistent layer kernel; not the default.
#pragma once
#include "kh_fused.h"

struct KhLayerAArgs {
  KhQkvArgs qkv;
  KhAttnArgs attn;
  KhGemvResArgs wo;
  KhSync sync;
  int n_qkv, n_attn, n_wo;  // workgroups per stage; grid = sum
};

template <bool QUANT, int UQ, int SQ, int G, int UW, int SW>
__global__ __launch_bounds__(KH_WG) void k_layer_a(const KhLayerAArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int b = (int)blockIdx.x;
  if (b < a.n_qkv) {
    qkv_body<QUANT, UQ, 4, SQ, true>(a.qkv, smem_raw, b, a.n_qkv, a.sync);
  } else if (b < a.n_qkv + a.n_attn) {
    attn_body<G, true>(a.attn, smem_raw, b - a.n_qkv, a.sync);
  } else {
    gemv_res_body<QUANT, UW, 4, SW, true>(a.wo, smem_raw, b - a.n_qkv - a.n_attn, a.n_wo, a.sync);
  }
}

// The instantiated combinations (launch shapes of the BASELINE configs, kh_model.hip::pick_shape);
// any other shape runs the three stand-alone kernels.
//   id 0: fp32 dim 2048 (Llama-3.2-1B, TinyLlama)   qkv<U4,S2> attn<16> wo<U4,S2>
//   id 1: int8 dim 4096 (Llama-2-7B int8)           qkv<U4,S1> attn<32> wo<U2,S2>
//   id 2: fp32 dim 4096 (Llama-2-7B fp32)           qkv<U8,S2> attn<32> wo<U8,S2>
//   id 3: fp32 dim 896  (Qwen2.5-0.5B)              qkv<U4,S1> attn<16> wo<U4,S1>
static inline int merged_combo_id(bool quant, int uq, int sq, int g, int uw, int sw) {
  if (!quant && uq == 4 && sq == 2 && g == 16 && uw == 4 && sw == 2) return 0;
  if (quant && uq == 4 && sq == 1 && g == 32 && uw == 2 && sw == 2) return 1;
  if (!quant && uq == 8 && sq == 2 && g == 32 && uw == 8 && sw == 2) return 2;
  if (!quant && uq == 4 && sq == 1 && g == 16 && uw == 4 && sw == 1) return 3;
  return -1;
}
static inline void launch_layer_a(int combo, int grid, size_t lds, hipStream_t s,
                                  const KhLayerAArgs& a) {
  switch (combo) {
    case 0:
      hipLaunchKernelGGL((k_layer_a<false, 4, 2, 16, 4, 2>), dim3(grid), dim3(KH_WG), lds, s, a);
      break;
    case 1:
      hipLaunchKernelGGL((k_layer_a<true, 4, 1, 32, 2, 2>), dim3(grid), dim3(KH_WG), lds, s, a);
      break;
    case 2:
      hipLaunchKernelGGL((k_layer_a<false, 8, 2, 32, 8, 2>), dim3(grid), dim3(KH_WG), lds, s, a);
      break;
    case 3:
      hipLaunchKernelGGL((k_layer_a<false, 4, 1, 16, 4, 1>), dim3(grid), dim3(KH_WG), lds, s, a);
      break;
    default:
      break;
  }
}
