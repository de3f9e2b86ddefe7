// kh_model_prefill.hip — prompt phase of the model level: the B-token VALU prefill (kh_prefill.h,
// bit-identical to token-by-token) and the fp32-MFMA GEMM prefill (kh_gemm.h, kh_pattn.h).  The
// reference feeds the prompt one token per forward pass (demo/main.cpp:20-22).
// gfx950 only.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "kh_gemm.h"
#include "kh_model_internal.h"
#include "kh_pattn.h"
#include "kh_prefill.h"

using namespace khm;

namespace khm {
// The B-token kernels mirror the decode kernels' arithmetic only for the staging variant the
// decode path uses at these sizes (in-register, at most 4 float4 per thread) and for the fast attention core.
bool prefill_supported(const kh_model* m) {
  const kh_config& c = m->cfg;
  if (c.head_size <= 32) return false;
  if (!kh_stage_fits4(c.dim, m->sh_qkv.wg) || !kh_stage_fits4(c.dim, m->sh_ffn.wg)) return false;
  if (m->sh_qkv.split > 2 || m->sh_ffn.split != 1) return false;
  if (pf_lds_bytes(c.is_quant, c.dim, 4) > 160 * 1024) return false;
  if (pf_lds_bytes(c.is_quant, c.hidden_dim, 2) > 160 * 1024) return false;
  return true;
}
bool pg_supported(const kh_model* m) {
  const kh_config& c = m->cfg;
  if (c.head_size <= 32) return false;  // attention: the fast multi-token decode kernel
  const int kq = c.is_quant ? 64 : 16;  // K granule of one MFMA operand load
  if (c.dim % kq || c.hidden_dim % kq || c.dim % 16 || c.kv_dim % 16 || c.hidden_dim % 16) return false;
  if (c.is_quant && m->gshift != 6) return false;
  return true;
}
}  // namespace khm

// ---- prompt prefill (kh_prefill.h) ---------------------------------------------------------------
namespace {
// tokens per pass of the dim-input matrices (qkv, wo, ffn13)
int prefill_batch(const kh_model* m) {
  const kh_config& c = m->cfg;
  if (!c.is_quant && pf_lds_bytes(false, c.dim, 8) <= 80 * 1024) return 8;
  return 4;
}
int ensure_prefill_buffers(kh_model* m) {
  if (m->pf_ready) return KH_OK;
  for (float** p : {&m->pf_x, &m->pf_q, &m->pf_att, &m->pf_h}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  if (m->pf_ws) (void)hipFree(m->pf_ws);
  m->pf_ws = nullptr;
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
  m->pf_ready = true;
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
    // the prompt phase leaves K/V rows and nothing else (no logits): the last layer's K/V rows are
    // written, its attention, wo and FFN feed nothing
    if (l == c.layer_num - 1) break;
    {
      KhAttnArgs a = fill_attn(m, l);
      a.defer = 0;  // multi-token slices merge in the launch
      a.nsplit_g = m->attn_ns_g;  // not the decode step's variant: launch_attn_decode decides from the slice's positions
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
int ensure_pg_buffers(kh_model* m) {
  if (m->pg_ready) return KH_OK;
  // a previous attempt may have failed half-way (out of memory): start from a clean slate so a
  // later call never launches GEMMs on null slabs
  for (float** p : {&m->pg_x, &m->pg_xn, &m->pg_q, &m->pg_att, &m->pg_h, &m->pg_part}) {
    if (*p) (void)hipFree(*p);
    *p = nullptr;
  }
  if (m->pg_ws) (void)hipFree(m->pg_ws);
  m->pg_ws = nullptr;
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
  if ((rc = zalloc(&m->pg_part, (size_t)KH_PG_KZ_MAX * T * c.dim)) != KH_OK) return rc;  // K-slice partial rows
  m->pg_ready = true;
  return KH_OK;
}
// does the prompt-slice attention run on the MFMA kernel (kh_pattn.h)?  KH_PG_ATTN=0 forces the fallback.
bool pg_mfma_attn(const kh_model* m) { return !dbg_off("KH_PG_ATTN") && pg_attn_supported(m->cfg.head_size); }
// The split workspace of the decode-attention FALLBACK (one per token of a pass, ~270 KB per token at
// Llama-3.2-1B geometry: 136 MB for a 512-token pass).  Every BASELINE head size has an MFMA
// instantiation, so it is allocated on the first pass that actually takes the fallback.
int ensure_pg_ws(kh_model* m) {
  if (m->pg_ws || pg_mfma_attn(m)) return KH_OK;
  m->pg_ws_tok_bytes = (attn_ws_bytes(m->cfg.head_num, m->cfg.head_size, m->attn_ws_stride) + 255) & ~(size_t)255;
  if (!m->pg_ws_tok_bytes) return KH_OK;
  KH_CHECK_HIP(hipMalloc(&m->pg_ws, m->pg_ws_tok_bytes * (size_t)KH_PG_TMAX));
  KH_CHECK_HIP(hipMemsetAsync(m->pg_ws, 0, m->pg_ws_tok_bytes * (size_t)KH_PG_TMAX, m->stream));
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
//     (these penalties were fitted with the phase scheme in every fp32 loop; since the uniform
//     operand rings of kh_gemm.h a second wave per SIMD costs less - (w1,w3) with 8 instead of 4
//     waves per workgroup: 3.01 vs 2.75 ms per 128-token pass - but still loses, and a re-sweep of
//     13 forced shapes leaves every choice of the model in place, profiles/r3_prefill_shapes512.txt);
//   * bigger register tiles need fewer operand bytes per MFMA (small factor), more token slices
//     re-read the weights from L2 (small factor), padding tokens are wasted MFMAs;
//   * fp32 passes of more than 128 tokens launch several times 256 workgroups, which would sit two
//     or more to a CU - two or more MFMA-bound waves per SIMD.  `solo` keeps ONE workgroup per CU at
//     a time (a dynamic-LDS request above half the CU's 160 KiB; the workgroups of a CU then run back
//     to back at the one-wave-per-SIMD rate) and is priced as such: Llama-3.2-1B, 512-token pass,
//     56.4 k prompt tok/s with it, 38.0 k without (profiles/r3_prefill_wide.txt).  Passes of <= 128
//     tokens use it for the MFMA-bound (2,8) tile only (Llama-2-7B fp32: 344 / 384 workgroups); their
//     small latency-bound tiles gain from a partner wave, and whole rounds of 256 cost more than
//     they save (Qwen2.5, (1,4) tile in 608 workgroups: 45.9 k shared, 41.8 k solo).
//   * the residual GEMMs (wo, w2) may split K across `kz` workgroups as well (blockIdx.z): with 2048
//     rows x 128 tokens there are only 64 (2,8) tiles, so a chip-filling launch either takes small
//     tiles or more K slices than a workgroup has waves.  The slices store partial rows and the
//     RMSNorm kernel that follows adds them in fixed order (kh_gemm.h) - no ticket, no second launch.
struct PgShape {
  int R, NT, ks, slices;
  bool solo;
  int kz;
};
// hooks read per call (kh_debug_set takes effect on the next launch plan)
static inline bool pg_solo_on() { return !dbg_off("KH_PG_SOLO"); }
static inline bool pg_kz_on() { return !dbg_off("KH_PG_KZ"); }
PgShape pg_shape(int T, int rows_total, bool r2_ok, int nm, int kblocks, int min_blocks, bool quant,
                 bool allow_kz = false) {
  const int nt_all = (T + 15) / 16;
  const bool wide = nt_all > 8;  // a pass of more than 128 tokens
  static const int cand[4][3] = {{2, 8, 2}, {2, 4, 3}, {2, 2, 4}, {1, 4, 4}};  // R, NT, waves/SIMD that fit
  PgShape best{1, 4, 1, (nt_all + 3) / 4, false, 1};
  double best_cost = -1.0;
  for (const auto& c : cand) {
    const int R = c[0], NT = c[1], occ = c[2];
    if (R == 2 && !r2_ok) continue;
    if (R == 1 && quant && r2_ok) continue;  // int8: the 32-row tile measured better wherever it fits
    if (NT > 4 && nt_all <= 4) continue;     // no 128-token tile for <= 64 tokens
    const int slices = (nt_all + NT - 1) / NT;
    for (int kz = 1; kz <= (allow_kz && pg_kz_on() ? KH_PG_KZ_MAX : 1); kz *= 2)
    for (int ks = 1; ks * nm * 64 <= KH_PG_WG_MAX(quant); ks *= 2) {
      const long wgs = (long)(rows_total / (16 * R)) * slices * kz;
      if (ks * kz > 1 && kblocks / (ks * kz) < min_blocks) break;  // keep a useful K range per wave
      // waves per SIMD on the busiest CU.  A workgroup counts as one wave on EACH SIMD it touches:
      // two 2-wave workgroups on a CU were measured on the same two SIMDs (Llama-3.2-1B, 256 tokens:
      // the (2,8) SwiGLU tile with 2-wave workgroups 255 us against 2 x 83 us for two 128-token
      // launches of 4-wave workgroups)
      const long cu_wgs = (wgs + 255) / 256;
      long wps = cu_wgs * ((nm * ks + 3) / 4);
      const long rounds = (wps + occ - 1) / occ;                    // beyond the register file: queued
      if (wps > occ) wps = occ;
      const double pen = quant ? (wps <= 1 ? 1.0 : (wps == 2 ? 0.72 : 0.62))
                               : (wps <= 1 ? 1.0 : (wps == 2 ? 1.6 : 1.8));
      const double per_wave = (double)((kblocks + ks * kz - 1) / (ks * kz)) * R * NT;
      double cost = per_wave * (double)(wps * rounds) * pen;
      cost *= 1.0 + 0.04 * (double)(kz - 1);  // partial rows written, and read back by the RMSNorm
      cost *= 1.0 + 0.15 * (double)(R + NT) / (double)(R * NT);
      // every token slice re-reads the weights from L2 - and, int8, dequantises them again
      cost *= 1.0 + (quant ? 0.15 : 0.05) * (double)(slices - 1);
      cost *= (double)(slices * NT) / (double)nt_all;
      bool solo = false;
      if (!quant && (wide || R * NT >= 16) && pg_solo_on() && wgs > 256) {
        const long wg_wps = (nm * ks + 3) / 4;
        const double c1 = per_wave * (double)(((wgs + 255) / 256) * wg_wps) * (wg_wps <= 1 ? 1.0 : (wg_wps == 2 ? 1.6 : 1.8));
        if (c1 < per_wave * (double)(wps * rounds) * pen) {
          cost *= c1 / (per_wave * (double)(wps * rounds) * pen);
          solo = true;
        }
      }
      if (best_cost < 0 || cost < best_cost) {
        best_cost = cost;
        best = PgShape{R, NT, ks, slices, solo, kz};
      }
    }
  }
  return best;
}
template <bool Q, int EPI>
bool pg_launch_cfg(const PgShape& sh, int tiles, int wg, hipStream_t s, const KhPgGemmArgs& a) {
  size_t lds = pg_lds_bytes(wg / 64, sh.NT);
  if (sh.solo && lds < KH_PG_SOLO_LDS) lds = KH_PG_SOLO_LDS;  // one workgroup per CU at a time
  auto go = [&](auto kern) {
    if (lds > 48 * 1024) {
      // the >48 KiB dynamic-LDS opt-in is per (device, kernel): set once, not on every launch of
      // the prefill hot path; a refusal is reported here, not as a generic launch error later
      struct Done { int dev; const void* fn; size_t lds; };
      static thread_local std::vector<Done> done;
      int dev = 0;
      (void)hipGetDevice(&dev);
      bool hit = false;
      for (const auto& d : done) hit = hit || (d.dev == dev && d.fn == (const void*)kern && d.lds >= lds);
      if (!hit) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) {
          (void)hipGetLastError();
          return false;
        }
        done.push_back({dev, (const void*)kern, lds});
      }
    }
    hipLaunchKernelGGL(kern, dim3(tiles, sh.slices, sh.kz), dim3(wg), lds, s, a);
    return true;
  };
  if (sh.R == 2 && sh.NT == 8) return go(k_pg_gemm<Q, 2, 8, EPI>);
  if (sh.R == 2 && sh.NT == 4) return go(k_pg_gemm<Q, 2, 4, EPI>);
  if (sh.R == 2) return go(k_pg_gemm<Q, 2, 2, EPI>);
  return go(k_pg_gemm<Q, 1, 4, EPI>);
}
// returns whether the QKV epilogue rotates q / k itself (else k_pg_rope has to follow); RESID: the
// number of K slices whose partial rows the next RMSNorm has to add (0 = the epilogue added itself)
template <int EPI>
int pg_launch(kh_model* m, int rows_total, bool r2_ok, KhPgGemmArgs a) {
  const bool q = m->cfg.is_quant;
  const int nm = EPI == KH_PG_SWIGLU ? 2 : 1;
  PgShape sh = pg_shape(a.T, rows_total, r2_ok, nm, a.K / (q ? 64 : 16), q ? 4 : 16, q, EPI == KH_PG_RESID);
  {  // tuning hook: KH_PG_SHAPE_<QKV|RESID|SWIGLU>="R,NT,ks" overrides the heuristic
    static const char* const names[3] = {"KH_PG_SHAPE_QKV", "KH_PG_SHAPE_RESID", "KH_PG_SHAPE_SWIGLU"};
    const char* const ov = dbg(names[EPI]);
    if (ov) {
      int R = 0, NT = 0, ks = 0, kz = 1;
      if (sscanf(ov, "%d,%d,%d,%d", &R, &NT, &ks, &kz) >= 3 && ((R == 2 && (NT == 2 || NT == 4 || NT == 8) && r2_ok) || (R == 1 && NT == 4)) &&
          (ks == 1 || ks == 2 || ks == 4 || ks == 8) && ks * nm * 64 <= KH_PG_WG_MAX(q) &&
          (kz == 1 || (EPI == KH_PG_RESID && (kz == 2 || kz == 4))))
      {
        const int slices = ((a.T + 15) / 16 + NT - 1) / NT;
        sh = PgShape{R, NT, ks, slices,
                     !q && (a.T > 128 || R * NT >= 16) && pg_solo_on() && (long)(rows_total / (16 * R)) * slices * kz > 256, kz};
      }
    }
  }
  if (dbg("KH_PG_DEBUG"))
    fprintf(stderr, "[pg] epi %d rows %d K %d T %d -> R %d NT %d slices %d ks %d kz %d (%d wgs x %d waves%s)\n", EPI,
            rows_total, a.K, a.T, sh.R, sh.NT, sh.slices, sh.ks, sh.kz, rows_total / (16 * sh.R) * sh.slices * sh.kz,
            nm * sh.ks, sh.solo ? ", solo" : "");
  if (EPI == KH_PG_QKV && a.rope == KH_PG_ROPE_TILES && (sh.R != 2 || sh.NT > 4))
    a.rope = KH_PG_ROPE_OFF;  // no partner tile in the wave / no registers to hold it: k_pg_rope follows
  const int tiles = rows_total / (16 * sh.R);
  const bool ok = q ? pg_launch_cfg<true, EPI>(sh, tiles, nm * sh.ks * 64, m->stream, a)
                    : pg_launch_cfg<false, EPI>(sh, tiles, nm * sh.ks * 64, m->stream, a);
  if (!ok) m->pg_launch_failed = true;  // reported by kh_model_prefill_gemm
  if (EPI == KH_PG_RESID) return sh.kz > 1 ? sh.kz : 0;
  return EPI == KH_PG_QKV && a.rope != KH_PG_ROPE_OFF;
}
// forward of T (<= KH_PG_TMAX) prompt tokens at positions pos0..: fills their K/V cache rows
void launch_prefill_gemm_chunk(kh_model* m, const int32_t* toks, int T, int pos0) {
  const kh_config& c = m->cfg;
  const bool q = c.is_quant;
  const int tcap = pg_tcap(T);  // token stride of this chunk's tiled slabs
  (void)kh_embedding_f32_host(toks, T, m->tok_emb, m->pg_x, c.dim, c.vocab_size, (void*)m->stream);
  int pending_kz = 0;  // K slices of the last residual GEMM still to be added to pg_x (by the next RMSNorm)
  auto rmsnorm = [&](const float* w) {
    if (q) hipLaunchKernelGGL(k_pg_rmsnorm<true>, dim3(T), dim3(KH_WG), 0, m->stream, m->pg_x, w, m->pg_xn, c.dim, c.rms_eps, tcap, (const float*)m->pg_part, pending_kz);
    else hipLaunchKernelGGL(k_pg_rmsnorm<false>, dim3(T), dim3(KH_WG), 0, m->stream, m->pg_x, w, m->pg_xn, c.dim, c.rms_eps, tcap, (const float*)m->pg_part, pending_kz);
    pending_kz = 0;
  };
  // attention of the slice: MFMA kernel (kh_pattn.h) unless KH_PG_ATTN=0 or an odd head size
  const bool mfma_attn = pg_mfma_attn(m);
  const bool rope_fuse_env = !dbg_off("KH_PG_ROPE_FUSE");
  for (int l = 0; l < c.layer_num; ++l) {
    const LayerW& W = m->layers[l];
    float* kc = m->kcache + (size_t)l * c.cache_len * c.kv_dim;
    float* vc = m->vcache + (size_t)l * c.cache_len * c.kv_dim;
    rmsnorm(W.att_norm);
    bool rope_fused = false;
    {
      KhPgGemmArgs a{};
      a.w[0] = W.wq; a.w[1] = W.wk; a.w[2] = W.wv;
      a.B = m->pg_xn; a.b_tiled = 1; a.tcap = tcap; a.out = m->pg_q; a.kc = kc; a.vc = vc;
      a.rows0 = c.dim; a.rows1 = c.kv_dim; a.ldo = c.dim; a.K = c.dim; a.T = T; a.pos0 = pos0;
      a.gshift = m->gshift;
      // RoPE in the epilogue: interleaved pairs sit in one lane's float4; half-mode partners need the
      // paired-tile mapping (R = 2, head size a multiple of 32).  KH_PG_ROPE_FUSE=0: separate kernel.
      a.head_size = c.head_size; a.sin_cache = m->sin_cache; a.cos_cache = m->cos_cache;
      a.rope = !rope_fuse_env ? KH_PG_ROPE_OFF
               : (c.rope_mode == KH_ROPE_HALF ? (c.head_size % 32 == 0 ? KH_PG_ROPE_TILES : KH_PG_ROPE_OFF)
                                              : KH_PG_ROPE_PAIRS);
      rope_fused = pg_launch<KH_PG_QKV>(m, c.dim + 2 * c.kv_dim, c.dim % 32 == 0 && c.kv_dim % 32 == 0, a) != 0;
    }
    if (!rope_fused)
      hipLaunchKernelGGL(k_pg_rope, dim3(T), dim3(KH_WG), 0, m->stream, m->pg_q, kc, m->sin_cache,
                         m->cos_cache, c.dim, c.kv_dim, c.head_size, pos0, c.rope_mode);
    // the prompt phase leaves K/V rows and nothing else (no logits): the last layer's K/V rows are
    // written, its attention, wo and FFN feed nothing (1/L of the pass minus one QKV GEMM)
    if (l == c.layer_num - 1) break;
    if (mfma_attn) {
      KhPgAttnArgs a{};
      a.q = m->pg_q; a.kc = kc; a.vc = vc; a.out = m->pg_att;
      a.dim = c.dim; a.kv_dim = c.kv_dim; a.kv_heads = c.kv_head_num; a.kv_mul = c.kv_mul;
      a.T = T; a.pos0 = pos0; a.layout = q ? KH_PA_TILED_Q8 : KH_PA_TILED_F32; a.tcap = tcap;
      launch_pg_attn(a, c.head_size, m->stream);
    } else {
      KhAttnArgs a = fill_attn(m, l);
      a.defer = 0;  // multi-token slices merge in the launch
      a.nsplit_g = m->attn_ns_g;  // not the decode step's variant: launch_attn_decode decides from the slice's positions
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
      a.B = m->pg_att; a.b_tiled = mfma_attn ? 1 : 0; a.tcap = tcap; a.out = m->pg_x;  // decode kernel: row-major rows
      a.rows0 = c.dim; a.ldo = c.dim; a.K = c.dim; a.T = T; a.gshift = m->gshift; a.part = m->pg_part;
      pending_kz = pg_launch<KH_PG_RESID>(m, c.dim, c.dim % 32 == 0, a);
    }
    rmsnorm(W.ffn_norm);
    {
      KhPgGemmArgs a{};
      a.w[0] = W.w1; a.w[1] = W.w3;
      a.B = m->pg_xn; a.b_tiled = 1; a.tcap = tcap; a.out = m->pg_h;
      a.rows0 = c.hidden_dim; a.ldo = c.hidden_dim; a.K = c.dim; a.T = T; a.gshift = m->gshift;
      pg_launch<KH_PG_SWIGLU>(m, c.hidden_dim, c.hidden_dim % 32 == 0, a);
    }
    {
      KhPgGemmArgs a{};
      a.w[0] = W.w2;
      a.B = m->pg_h; a.b_tiled = 1; a.tcap = tcap; a.out = m->pg_x;
      a.rows0 = c.dim; a.ldo = c.dim; a.K = c.hidden_dim; a.T = T; a.gshift = m->gshift; a.part = m->pg_part;
      pending_kz = pg_launch<KH_PG_RESID>(m, c.dim, c.dim % 32 == 0, a);  // (the next layer's RMSNorm adds them)
    }
  }
}
}  // namespace

// Host-only view of the prefill GEMM launch plan (pg_shape) for tools and the CPU test-suite: epi 0 = QKV
// (rows = dim + 2 kv_dim), 1 = residual GEMM (wo / w2: rows = dim), 2 = SwiGLU pair (rows = hidden_dim); K =
// contraction length; r2_ok = the 32-row register tile divides the row blocks.  out[7] = {R, NT, ks, token
// slices, solo, kz, workgroups}.  No device is touched; the KH_PG_SHAPE_* overrides are not applied.
extern "C" int kh_plan_prefill_shape(int32_t epi, int32_t T, int32_t rows, int32_t K, int32_t is_quant,
                                     int32_t r2_ok, int32_t* out7) {
  if (!out7 || epi < 0 || epi > 2 || T <= 0 || T > KH_PG_TMAX || rows <= 0 || K <= 0) return KH_ERR_INVALID_ARG;
  const bool q = is_quant != 0;
  if (rows % 16 || K % (q ? 64 : 16)) return KH_ERR_UNSUPPORTED;
  const PgShape sh = pg_shape(T, rows, r2_ok != 0, epi == KH_PG_SWIGLU ? 2 : 1, K / (q ? 64 : 16), q ? 4 : 16, q,
                              epi == KH_PG_RESID);
  out7[0] = sh.R; out7[1] = sh.NT; out7[2] = sh.ks; out7[3] = sh.slices; out7[4] = sh.solo ? 1 : 0; out7[5] = sh.kz;
  out7[6] = rows / (16 * sh.R) * sh.slices * sh.kz;
  return KH_OK;
}

extern "C" int kh_model_prefill_gemm(kh_model* m, const int32_t* h_tokens, int32_t n, int32_t pos0) {
  if (!m || !h_tokens || n <= 0 || pos0 < 0) return KH_ERR_INVALID_ARG;
  const kh_config& c = m->cfg;
  if ((int64_t)pos0 + n > c.cache_len) return KH_ERR_RANGE;
  for (int i = 0; i < n; ++i)
    if (h_tokens[i] < 0 || h_tokens[i] >= c.vocab_size) return KH_ERR_RANGE;
  if (!pg_supported(m)) return KH_ERR_UNSUPPORTED;
  KH_CHECK_HIP(hipSetDevice(m->opts.device));
  int rc;
  if ((rc = kv_ensure(m, pos0 + n)) != KH_OK) return rc;
  if ((rc = ensure_pg_buffers(m)) != KH_OK) return rc;
  if ((rc = ensure_pg_ws(m)) != KH_OK) return rc;
  m->pg_launch_failed = false;
  // tuning / test hook: KH_PG_CHUNK=<tokens per weight pass> (16 .. KH_PG_TMAX), read per call
  int chunk = KH_PG_TMAX;
  if (const char* e = dbg("KH_PG_CHUNK")) {
    const int v = atoi(e);
    chunk = v < 16 ? 16 : (v > KH_PG_TMAX ? KH_PG_TMAX : v);
  }
  for (int t0 = 0; t0 < n; t0 += chunk)
    launch_prefill_gemm_chunk(m, h_tokens + t0, n - t0 < chunk ? n - t0 : chunk, pos0 + t0);
  if (m->pg_launch_failed) return KH_ERR_UNSUPPORTED;  // a GEMM shape's LDS opt-in was refused
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
  if ((rc = kv_ensure(m, pos0 + n)) != KH_OK) return rc;
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
  if ((rc = kv_ensure(m, pos0 + n + 1)) != KH_OK) return rc;
  if (mode == KH_PREFILL_TOKEN) {
    if ((rc = ensure_seq_cap(m, pos0 + n + 1)) != KH_OK) return rc;
    std::vector<int32_t> forced((size_t)m->seq_cap + 1, -1);
    for (int i = 0; i < n; ++i) forced[pos0 + i] = h_tokens[i];
    forced[pos0 + n] = h_tokens[n - 1];  // keeps the last timed step in the prompt phase
    KH_CHECK_HIP(hipMemcpyAsync(m->d_forced, forced.data(), forced.size() * sizeof(int32_t),
                                hipMemcpyHostToDevice, m->stream));
    KH_CHECK_HIP(hipStreamSynchronize(m->stream));
    const int n_forced = m->seq_cap + 1;
    // capture (first use) outside the timed region
    hipGraphExec_t ge = nullptr;
    for (int s = 0; s < n;) {
      const bool n8 = n - s >= KH_GRAPH_STEPS;
      const int k = n8 ? KH_GRAPH_STEPS : 1;
      if ((rc = step_graph(m, n_forced, step_variant(m, pos0 + s, pos0 + s + k - 1), n8, &ge)) != KH_OK) return rc;
      s += k;
    }
    set_state(m, h_tokens[0], pos0);
    KH_CHECK_HIP(hipEventRecord(m->ev0, m->stream));
    for (int s = 0; s < n;) {
      const bool n8 = n - s >= KH_GRAPH_STEPS;
      const int k = n8 ? KH_GRAPH_STEPS : 1;
      if ((rc = step_graph(m, n_forced, step_variant(m, pos0 + s, pos0 + s + k - 1), n8, &ge)) != KH_OK) return rc;
      KH_CHECK_HIP(hipGraphLaunch(ge, m->stream));
      s += k;
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
    if ((rc = ensure_pg_ws(m)) != KH_OK) return rc;
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
