// kh_pattn.h — causal attention of a prompt slice on the matrix cores (GEMM prefill).
//
// The reference computes attention one query token at a time (cpu/mha_kernel.cpp:5-61,
// cuda/mha_kernel.cu:47-110: score = q.k * 1/sqrt(hs), softmax over 0..pos, out = sum p_t v_t).
// With T prompt tokens in flight, q.K^T and P.V are real GEMMs: here one workgroup owns (head h,
// 16 query tokens) and its four or eight waves split the key/value timesteps in 16-position tiles; both
// contractions run on v_mfma_f32_16x16x4_f32 (exact fp32 fmaf chains), the softmax is the online
// form the decode kernel already uses (running max m, running sum l, rescale by exp(m - m')).
//
// The transposed products make the operands fall out of the registers without any shuffle:
//   S^T[pos, tok] = K[pos, :] . Q[tok, :]      A = K rows (lane i = l&15 -> position), B = Q rows
//       D layout: lane holds column tok = l&15, rows pos = 4*(l>>4) + reg            (reg 0..3)
//   O^T[d, tok]  += V^T[d, pos] . P^T[pos, tok]
//       MFMA #s contracts the position quad {s, 4+s, 8+s, 12+s}: its B operand for k = l>>4 is the
//       lane's OWN register s of P^T - the D layout of the first product IS the B layout of the
//       second.  A = V[pos = 4*(l>>4)+s][d]: the lane loads a float4 V[pos][64c + 4i .. +4] and feeds
//       component a to the accumulator tile (c, a), whose row i stands for d = 64c + 4i + a; in the
//       D layout a lane then holds d = 64c + 16*(l>>4) + 4*reg + a, i.e. 16 consecutive outputs of
//       its token - four float4 stores.
// As in kh_gemm.h the K-side contraction visits the head dimension in the order of the lanes'
// float4 loads (k-quad {16b+s, 16b+4+s, ...}); any order is a valid dot product.
//
// Row statistics: a token's scores sit in 4 registers x 4 lane groups (l>>4); the tile maximum is
// combined across the groups with two ds_bpermute, so every lane of a token uses the same m and
// the P values fed to the matrix core are consistently scaled.  The row sum is kept per lane and
// combined once at the end.  The waves' (m, l, O) partials merge through LDS in fixed order.
//
// Output goes straight into the tiled activation slab the wo GEMM reads (pg_tiled_index).
#pragma once
#include "kh_gemm.h"

struct KhPgAttnArgs {
  const float* q;   // [T][dim] row-major, RoPE applied
  const float* kc;  // this layer's K cache rows [cache_len][kv_dim]; rows pos0 .. pos0+T-1 just written
  const float* vc;
  float* out;       // layout 0 / 1: tiled slab [dim x tcap] of an fp32 / int8 model
                    // (pg_tiled_index, T <= tcap); layout 2: row-major [T][dim]
  int dim, kv_dim, kv_heads, kv_mul, T, pos0, layout, tcap;
};
enum { KH_PA_TILED_F32 = 0, KH_PA_TILED_Q8 = 1, KH_PA_ROWS = 2 };

template <int HB /* head_size / 16 */, int NW /* waves that split the timesteps: 4 or 8 */,
          int QT /* 16-token query tiles per workgroup: every K/V fragment a wave loads feeds QT tiles */>
__global__ __launch_bounds__(64 * NW) void k_pg_attn(const KhPgAttnArgs a) {
  constexpr int HS = 16 * HB;
  constexpr int NC = (HS + 63) / 64;  // 64-wide chunks of the head dimension on the P.V side
  constexpr int LDO = HS + 4;         // LDS row stride (floats): float4 rows of 16 tokens hit distinct banks
  __shared__ __attribute__((aligned(16))) float o_lds[NW][16][LDO];
  __shared__ float m_lds[NW][16], l_lds[NW][16];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lj = lane >> 4;
  // workgroup -> (kv group g, head-in-group j, token tile tt); b % 8 is the XCD: the kv_mul heads
  // (and all token tiles) that share K/V rows share an XCD's L2.  Long tiles are dispatched first.
  const int b = (int)blockIdx.x;
  const int g = b % a.kv_heads;
  const int rest = b / a.kv_heads;
  const int j = rest % a.kv_mul;
  const int n_tt = (a.T + 16 * QT - 1) / (16 * QT);
  const int tt = n_tt - 1 - rest / a.kv_mul;
  const int h = g * a.kv_mul + j;
  const int t0 = tt * 16 * QT;
  const int t_last = t0 + 16 * QT - 1 < a.T - 1 ? t0 + 16 * QT - 1 : a.T - 1;
  const int p_last = a.pos0 + t_last;  // last timestep any token of this workgroup attends to
  const int n_pt = (p_last >> 4) + 1;  // 16-position tiles, aligned at timestep 0
  // scores are kept in the log2 domain (score * 1/sqrt(hs) * log2(e)) so that the softmax runs on the
  // hardware exp2 (v_exp_f32), like the GQA decode path of kh_attn.h: exp(s - m) == exp2(s2 - m2)
  const float scale = (1.0f / sqrtf((float)HS)) * 1.4426950408889634f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  const float* kbase = a.kc + (size_t)g * HS + 4 * lj;
  const float* vbase = a.vc + (size_t)g * HS + 4 * li;
  auto load_k = [&](f32x4(&kf)[HB], int pt) __attribute__((always_inline)) {
    int p = pt * 16 + li;
    p = p < p_last ? p : p_last;  // rows past the slice are not written yet: clamp, masked below
    const float* r = kbase + (size_t)p * a.kv_dim;
#pragma unroll
    for (int bb = 0; bb < HB; ++bb) kf[bb] = *(const f32x4*)(r + 16 * bb);
  };
  auto load_v = [&](f32x4(&vf)[NC][4], int pt) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      int p = pt * 16 + lj * 4 + s;
      p = p < p_last ? p : p_last;
      const float* r = vbase + (size_t)p * a.kv_dim;
#pragma unroll
      for (int c = 0; c < NC; ++c)
        vf[c][s] = (64 * c + 4 * li < HS) ? *(const f32x4*)(r + 64 * c) : zero4;
    }
  };

  f32x4 kf[2][HB], vf[2][NC][4];
  if (wave < n_pt) {  // wave-uniform
    load_k(kf[0], wave);
    load_v(vf[0], wave);
  }
  // query tile qi: this lane's query token (column of both products; padding columns repeat the last
  // token) and the last timestep it attends to
  f32x4 qf[QT][HB];
  int my_pos[QT];
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) {
    const int t = t0 + 16 * qi + li;
    const int tq = t < a.T ? t : a.T - 1;
    my_pos[qi] = a.pos0 + tq;
    const float* qrow = a.q + (size_t)tq * a.dim + (size_t)h * HS + 4 * lj;
#pragma unroll
    for (int bb = 0; bb < HB; ++bb) qf[qi][bb] = *(const f32x4*)(qrow + 16 * bb);
  }
  float m[QT], l[QT];
  f32x4 oacc[QT][NC][4];
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) {
    m[qi] = -INFINITY;
    l[qi] = 0.f;
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) oacc[qi][c][e] = zero4;
  }

  auto tile = [&](const f32x4(&kc_)[HB], const f32x4(&vc_)[NC][4], int pt) __attribute__((always_inline)) {
    const int pb = pt * 16 + lj * 4;
#pragma unroll
    for (int qi = 0; qi < QT; ++qi) {
      // position tiles wholly above the diagonal of this query tile: nothing to add (wave-uniform)
      if (QT > 1 && pt * 16 > a.pos0 + t0 + 16 * qi + 15) continue;
      f32x4 sacc = zero4;
#pragma unroll
      for (int bb = 0; bb < HB; ++bb)
#pragma unroll
        for (int s = 0; s < 4; ++s) sacc = mfma16(pg_comp(kc_[bb], s), pg_comp(qf[qi][bb], s), sacc);
      float sv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) sv[r] = (pb + r <= my_pos[qi]) ? pg_comp(sacc, r) * scale : -INFINITY;
      float tm = fmaxf(fmaxf(sv[0], sv[1]), fmaxf(sv[2], sv[3]));
      tm = fmaxf(tm, __shfl_xor(tm, 16));
      tm = fmaxf(tm, __shfl_xor(tm, 32));
      const float m_new = fmaxf(m[qi], tm);
      const float m_ref = m_new == -INFINITY ? 0.f : m_new;  // a token that has seen no timestep yet
      const float alpha = __builtin_amdgcn_exp2f(m[qi] - m_ref);  // exp2(-inf) = 0 on its first timestep
      float p[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) p[r] = __builtin_amdgcn_exp2f(sv[r] - m_ref);
      l[qi] = l[qi] * alpha + ((p[0] + p[1]) + (p[2] + p[3]));
#pragma unroll
      for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          f32x4 o = oacc[qi][c][e];
          o.x *= alpha; o.y *= alpha; o.z *= alpha; o.w *= alpha;
#pragma unroll
          for (int s = 0; s < 4; ++s) o = mfma16(pg_comp(vc_[c][s], e), p[s], o);
          oacc[qi][c][e] = o;
        }
      m[qi] = m_new;
    }
  };

  for (int pt = wave; pt < n_pt; pt += 2 * NW) {
    const int p1 = pt + NW, p2 = pt + 2 * NW;
    if (p1 < n_pt) {
      load_k(kf[1], p1);
      load_v(vf[1], p1);
    }
    __builtin_amdgcn_sched_barrier(0);
    tile(kf[0], vf[0], pt);
    if (p1 < n_pt) {
      if (p2 < n_pt) {
        load_k(kf[0], p2);
        load_v(vf[0], p2);
      }
      __builtin_amdgcn_sched_barrier(0);
      tile(kf[1], vf[1], p1);
    }
  }

  // ---- merge the waves' partials (fixed order), one query tile at a time through the same LDS area -----
#pragma unroll
  for (int qi = 0; qi < QT; ++qi) {
    if (qi > 0) __syncthreads();
    float lq = l[qi];
    lq += __shfl_xor(lq, 16);
    lq += __shfl_xor(lq, 32);
    if (lj == 0) {
      m_lds[wave][li] = m[qi];
      l_lds[wave][li] = lq;
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const int d0 = 64 * c + 16 * lj;
      if (d0 < HS) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          f32x4 v = {pg_comp(oacc[qi][c][0], r), pg_comp(oacc[qi][c][1], r), pg_comp(oacc[qi][c][2], r),
                     pg_comp(oacc[qi][c][3], r)};
          *(f32x4*)&o_lds[wave][li][d0 + 4 * r] = v;
        }
      }
    }
    __syncthreads();
    const int tok = tid & 15;
    const int t = t0 + 16 * qi + tok;
    if (t < a.T) {
      float M = m_lds[0][tok];
#pragma unroll
      for (int w = 1; w < NW; ++w) M = fmaxf(M, m_lds[w][tok]);  // finite: a token sees its own timestep
      float f[NW], L = 0.f;
#pragma unroll
      for (int w = 0; w < NW; ++w) {
        f[w] = __builtin_amdgcn_exp2f(m_lds[w][tok] - M);  // waves without a timestep: m = -inf -> 0
        L += f[w] * l_lds[w][tok];
      }
      for (int grp = tid >> 4; grp < HS / 4; grp += 4 * NW) {
        f32x4 r = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < NW; ++w) {
          const f32x4 o = *(const f32x4*)&o_lds[w][tok][4 * grp];
          r.x += f[w] * o.x; r.y += f[w] * o.y; r.z += f[w] * o.z; r.w += f[w] * o.w;
        }
        r.x /= L; r.y /= L; r.z /= L; r.w /= L;
        const int k = h * HS + 4 * grp;
        const size_t at = a.layout == KH_PA_ROWS ? (size_t)t * a.dim + k
                                                 : pg_tiled_index(a.layout == KH_PA_TILED_Q8, k, t, a.tcap);
        *(f32x4*)(a.out + at) = r;
      }
    }
  }
}

static inline bool pg_attn_supported(int head_size) {
  return head_size == 48 || head_size == 64 || head_size == 128;
}
// Query tiles per workgroup: every doubling halves the K/V bytes a workgroup pulls through L2 per MFMA
// (256 B per MFMA with one tile).  Measured (profiles/r3_pattn_qt.txt, Llama-3.2-1B, 512-token pass): the
// kernel is closer to MFMA-bound than to L2-bound - 4 tiles take a pass at start position 32256 from 31.95
// to 28.9 ms (attention 1.42 -> 1.23 ms per layer = 111 TFLOP/s of the ~135 the matrix cores deliver in
// fp32), 2 % at position 4096, and LOSE 2 % at position 0 (fewer, longer workgroups on the causal
// triangle); with fewer than 256 workgroups they lose outright (128-token pass at 32256: 8.8 -> 13.8 ms).
// So: more than one tile only from start position 2048 on and only while the launch keeps >= 256
// workgroups.  Head size 128 stops at 2 (registers).  KH_PG_ATTN_QT forces a value.
static inline int pg_attn_qt(const KhPgAttnArgs& a, int head_size) {
  const int heads = a.kv_heads * a.kv_mul, maxqt = head_size == 128 ? 2 : 4;
  const char* e = khm::dbg("KH_PG_ATTN_QT");
  const int forced = e ? atoi(e) : 0;
  if (forced == 1 || forced == 2 || forced == 4) return forced < maxqt ? forced : maxqt;
  if (a.pos0 < 2048) return 1;
  int qt = 1;
  while (qt < maxqt && heads * ((a.T + 32 * qt - 1) / (32 * qt)) >= 256) qt *= 2;
  return qt;
}
static inline void launch_pg_attn(const KhPgAttnArgs& a, int head_size, hipStream_t s) {
  const int qt = pg_attn_qt(a, head_size);
  const int grid = a.kv_heads * a.kv_mul * ((a.T + 16 * qt - 1) / (16 * qt));
  // 8 waves (two per SIMD: one wave's softmax fills the other's MFMA shadow) where the registers
  // allow it; head size 128 keeps 4
#define KH_PA_GO(HB, NW, QT) hipLaunchKernelGGL((k_pg_attn<HB, NW, QT>), dim3(grid), dim3(64 * NW), 0, s, a)
  switch (head_size) {
    case 48: if (qt == 4) KH_PA_GO(3, 8, 4); else if (qt == 2) KH_PA_GO(3, 8, 2); else KH_PA_GO(3, 8, 1); break;
    case 64: if (qt == 4) KH_PA_GO(4, 8, 4); else if (qt == 2) KH_PA_GO(4, 8, 2); else KH_PA_GO(4, 8, 1); break;
    case 128: if (qt >= 2) KH_PA_GO(8, 4, 2); else KH_PA_GO(8, 4, 1); break;
    default: break;
  }
#undef KH_PA_GO
}
