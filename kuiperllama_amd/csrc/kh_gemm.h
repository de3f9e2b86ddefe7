// kh_gemm.h — prompt prefill as real GEMMs on the matrix cores.
//
// The reference feeds the prompt one token per forward pass (demo/main.cpp:20-22): every prompt
// token streams all weights.  kh_prefill.h already shares one weight pass between 8 (fp32) / 4
// (int8) tokens on the VALU; here up to KH_PG_TMAX = 128 prompt tokens share ONE pass and the
// contraction runs on MFMA:  C[rows, T] = W[rows, K] . Xn[T, K]^T  with
//     v_mfma_f32_16x16x4_f32   (f32 in, f32 accumulate: bit-for-bit a k-ordered fmaf chain,
//                               MI355X guide §3 — exact fp32, no reduced-precision path)
// Tokens sit on the MFMA N dimension, weight rows on M.  Arithmetic intensity at T = 128 is
// 64 flop per weight byte (fp32) against a machine balance of ~22, so the GEMMs are MFMA-bound
// (157 TF f32 peak) and the weights cross HBM once per 128 tokens.
//
// Mapping (one wave = one 16-row weight tile x NT 16-token tiles x a K slice):
//   * A fragment of 16x16x4: lane (i = l&15, h = l>>4) supplies W[row i][k].  The lane loads ONE
//     float4 = W[i][16b + 4h .. +4] per 16-column block b straight from HBM into VGPRs (non-temporal:
//     every weight byte is read by exactly one wave) and feeds its four components to four MFMAs;
//     MFMA #s therefore contracts the k-quad {16b+s, 16b+4+s, 16b+8+s, 16b+12+s} - any k order is as good
//     as any other for a dot product, and this one makes both operands plain 16-byte loads.
//   * B fragment: lane (j = l&15, h) supplies Xn[token j][same k], one float4 per token tile per
//     block, read from the (L2-resident, <= 4 MiB) normalised activation slab.
//   * C/D: lane (j, hq = l>>4) holds rows 4hq..4hq+3 of the tile for token j: a float4 of four
//     consecutive output rows of one token, which is exactly what the epilogues store.
//   * a workgroup = ks waves that split K (fixed-order LDS reduction, deterministic), times two
//     for the (w1, w3) SwiGLU pair.  ks is chosen per GEMM so that every launch has >= ~1-2 waves
//     per SIMD although a 16-row tile x 128 tokens is a big unit.
//   * weights are prefetched 8 blocks (~8 k MFMA cycles) ahead through a register ring, the
//     activation operand one block ahead.
// int8 (group 64): the lane's 16-byte load is 16 consecutive weights of one 64-group; they are
// converted once to scale * float(w) - the reference's per-element dequant (cuda/matmul_kernel.cu:73)
// - and feed 16 MFMAs per token tile.
//
// Numerics: not bit-identical to the decode GEMVs (different summation order); parity is held to the
// fp32 tolerance against the oracle (K/V rows 5e-6, following logits 2e-5, same greedy tokens;
// tests/test_model_gpu.py::test_gemm_prefill_*).  The bit-exact B-token path stays available
// (kh_model_prefill / KH_PREFILL=gemv).
#pragma once
#include "kh_fused.h"

#define KH_PG_TMAX 128           // prompt tokens per weight pass (8 MFMA token tiles)
#define KH_PG_WG_MAX 512         // <= 8 waves per workgroup: 256 VGPRs per lane stay available
#define KH_PG_RING 8             // weight blocks per register ring (two rings per wave)

enum { KH_PG_QKV = 0, KH_PG_RESID = 1, KH_PG_SWIGLU = 2 };

struct KhPgGemmArgs {
  KhLin w[3];        // QKV: wq, wk, wv ; RESID: w[0] ; SWIGLU: w1, w3
  const float* B;    // [KH_PG_TMAX][K] activation slab (rows >= T hold finite garbage)
  float* out;        // QKV: Q [T][ldo] ; RESID: X [T][ldo] (+=) ; SWIGLU: H [T][ldo]
  float* kc;         // QKV: K cache rows of this layer, row (pos0 + t) * kv_dim
  float* vc;
  int rows0, rows1;  // QKV: rows of wq, rows of wk (= rows of wv); else rows0 = rows
  int ldo, K, T, pos0, gshift;
};

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// ---- K loops -------------------------------------------------------------------------------------
// A wave has ONE in-order counter for its vector-memory loads (s_waitcnt vmcnt): a wait for a young
// L2-latency load (the activation operand of the next block) also waits for every OLDER load, in
// particular for HBM-latency weight loads issued before it.  A ring that refills one weight block
// per iteration therefore stalls every iteration for an HBM round trip.  Here the weights arrive in
// PHASES: two register rings of KH_PG_RING blocks; while ring `cur` is consumed, ring `nxt` is
// requested in ONE batch right behind the first activation prefetch of the phase, so its latency
// is exposed at most once per phase (8 blocks = 8 k MFMA cycles at 8 token tiles) instead of once
// per block, and only for what exceeds the two iterations that run before the next young load is
// awaited.
// The activation operand ping-pongs between two register sets xb[0] / xb[1] (no copies: a copy
// would make the compiler wait for the prefetch as soon as it is issued).
template <int NT, int SUB /* operand sub-steps per block: 1 fp32, 4 int8 */, class LoadA, class LoadB,
          class Mfma>
__device__ __forceinline__ void pg_phases(int b0, int b1, LoadA&& load_a, LoadB&& load_b,
                                          Mfma&& mfma_sub) {
  static_assert((KH_PG_RING * SUB) % 2 == 0, "ping-pong parity must restart with every phase");
  const int last = b1 - 1;
  load_a(0, b0);  // ring 0 <- blocks b0 .. b0+R-1
  f32x4 xb[2][NT];
  load_b(xb[0], b0, 0);
  for (int b = b0; b < b1; b += 2 * KH_PG_RING) {
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int base = b + ph * KH_PG_RING;
      if (base < b1) {  // wave-uniform (b0, b1 are scalars)
#pragma unroll
        for (int d = 0; d < KH_PG_RING; ++d) {
          const int bb = base + d;
          if (bb < b1) {
#pragma unroll
            for (int q = 0; q < SUB; ++q) {
              const int cur = (d * SUB + q) & 1;
              // next sub-step's activations first (young, L2 latency) ...
              const int nb_ = q + 1 < SUB ? bb : (bb + 1 < last ? bb + 1 : last);
              load_b(xb[cur ^ 1], nb_, q + 1 < SUB ? q + 1 : 0);
              // ... then, once per phase, the whole next weight ring (HBM latency)
              if (d == 0 && q == 0) load_a(1 - ph, base + KH_PG_RING < last ? base + KH_PG_RING : last);
              __builtin_amdgcn_sched_barrier(0);  // the prefetches are issued before the MFMAs
              mfma_sub(ph, d, q, xb[cur]);
            }
          }
        }
      }
    }
  }
}

// fp32 weights: blocks of 16 columns; lane (i, h) owns W[i][16b + 4h .. +4]
template <int NT>
__device__ __forceinline__ void pg_kloop_f32(const float* __restrict__ wrow /* row i, + 4h */,
                                             const float* __restrict__ brow /* token i, + 4h */,
                                             size_t tile_stride /* 16 * K floats */, int b0, int b1,
                                             f32x4 (&acc)[NT]) {
  f32x4 a[2][KH_PG_RING];
  const int last = b1 - 1;
  auto load_a = [&](int ring, int base) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < KH_PG_RING; ++d) {
      const int bb = base + d < last ? base + d : last;
      a[ring][d] = ld_nt((const f32x4*)(wrow + (size_t)bb * 16));
    }
  };
  auto load_b = [&](f32x4 (&x)[NT], int bb, int) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) x[nt] = *(const f32x4*)(brow + nt * tile_stride + (size_t)bb * 16);
  };
  auto sub = [&](int ph, int d, int, const f32x4 (&x)[NT]) __attribute__((always_inline)) {
    const f32x4 av = a[ph][d];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(av.x, x[nt].x, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(av.y, x[nt].y, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(av.z, x[nt].z, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(av.w, x[nt].w, acc[nt]);
  };
  pg_phases<NT, 1>(b0, b1, load_a, load_b, sub);
}

// int8 group-64 weights: blocks of 64 columns.  Lane (i, h) owns the 16 weights
// W8[i][64b + 16h .. +16] (one dwordx4, one group -> one scale); quarter q of them pairs with
// Xn[token][64b + 16h + 4q .. +4].
template <int NT>
__device__ __forceinline__ void pg_kloop_q8(const int8_t* __restrict__ wrow /* row i, + 16h */,
                                            const float* __restrict__ srow /* scales of row i */,
                                            const float* __restrict__ brow /* token i, + 16h */,
                                            size_t tile_stride, int b0, int b1, f32x4 (&acc)[NT]) {
  i32x4 qw[2][KH_PG_RING];
  float sc[2][KH_PG_RING];
  const int last = b1 - 1;
  auto load_a = [&](int ring, int base) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < KH_PG_RING; ++d) {
      const int bb = base + d < last ? base + d : last;
      qw[ring][d] = ld_nt((const i32x4*)(wrow + (size_t)bb * 64));
      sc[ring][d] = srow[bb];
    }
  };
  auto load_b = [&](f32x4 (&x)[NT], int bb, int q) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      x[nt] = *(const f32x4*)(brow + nt * tile_stride + (size_t)bb * 64 + 4 * q);
  };
  auto sub = [&](int ph, int d, int q, const f32x4 (&x)[NT]) __attribute__((always_inline)) {
    const float s = sc[ph][d];
    const i32x4 qv = qw[ph][d];
    const int dw = q == 0 ? qv.x : (q == 1 ? qv.y : (q == 2 ? qv.z : qv.w));
    // dequantised weight = scale * float(w8): the reference's per-element form
    const float w0 = s * (float)(int8_t)(dw & 0xff);
    const float w1 = s * (float)(int8_t)((dw >> 8) & 0xff);
    const float w2 = s * (float)(int8_t)((dw >> 16) & 0xff);
    const float w3 = s * (float)(dw >> 24);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(w0, x[nt].x, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(w1, x[nt].y, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(w2, x[nt].z, acc[nt]);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[nt] = mfma16(w3, x[nt].w, acc[nt]);
  };
  pg_phases<NT, 4>(b0, b1, load_a, load_b, sub);
}

// blockDim.x = NM * ks * 64 (NM = 2 for SWIGLU); blockIdx.x = 16-row tile.
// LDS: [waves][NT][64] float4 partial tiles (skipped when one wave owns the whole K range).
template <bool QUANT, int NT, int EPI>
__global__ __launch_bounds__(KH_PG_WG_MAX) void k_pg_gemm(const KhPgGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int NM = EPI == KH_PG_SWIGLU ? 2 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = (int)(blockDim.x >> 6);
  const int ks = nw / NM;
  const int mat = wave / ks, kpart = wave - mat * ks;
  const int i = lane & 15, h = lane >> 4;
  const int K = a.K;
  int row0 = (int)blockIdx.x * 16;  // first row of the tile in the stacked output
  // weight matrix of this tile
  KhLin W = a.w[mat];
  int wr0 = row0;                   // first row inside that matrix
  if (EPI == KH_PG_QKV) {
    if (row0 >= a.rows0 + a.rows1) {
      W = a.w[2];
      wr0 = row0 - a.rows0 - a.rows1;
    } else if (row0 >= a.rows0) {
      W = a.w[1];
      wr0 = row0 - a.rows0;
    }
  }
  const size_t tile_stride = (size_t)16 * K;
  f32x4 acc[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (the K range depends on the wave index only: keep it in scalar registers so the block guards
  // of the K loop are scalar branches)
  if (!QUANT) {
    const int nb = K >> 4;
    const int b0 = __builtin_amdgcn_readfirstlane((int)((long)kpart * nb / ks));
    const int b1 = __builtin_amdgcn_readfirstlane((int)((long)(kpart + 1) * nb / ks));
    const float* wrow = (const float*)W.w + (size_t)(wr0 + i) * K + 4 * h;
    const float* brow = a.B + (size_t)i * K + 4 * h;
    if (b1 > b0) pg_kloop_f32<NT>(wrow, brow, tile_stride, b0, b1, acc);
  } else {
    const int nb = K >> 6;
    const int b0 = __builtin_amdgcn_readfirstlane((int)((long)kpart * nb / ks));
    const int b1 = __builtin_amdgcn_readfirstlane((int)((long)(kpart + 1) * nb / ks));
    const int8_t* wrow = (const int8_t*)W.w + (size_t)(wr0 + i) * K + 16 * h;
    const float* srow = W.scales + (size_t)(wr0 + i) * nb;
    const float* brow = a.B + (size_t)i * K + 16 * h;
    if (b1 > b0) pg_kloop_q8<NT>(wrow, srow, brow, tile_stride, b0, b1, acc);
  }
  // ---- combine the K slices (fixed order) and run the epilogue on float4 = 4 rows x 1 token --
  f32x4* red = (f32x4*)smem_raw;
  const bool direct = nw == 1;
  if (!direct) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) red[((size_t)wave * NT + nt) * 64 + lane] = acc[nt];
    __syncthreads();
  }
  for (int e = threadIdx.x; e < NT * 64; e += (int)blockDim.x) {
    const int nt = e >> 6, ln = e & 63;
    f32x4 v0, v1 = f32x4{0.f, 0.f, 0.f, 0.f};
    if (direct) {
      v0 = acc[0];
#pragma unroll
      for (int t = 1; t < NT; ++t)
        if (t == nt) v0 = acc[t];
    } else {
      v0 = red[((size_t)0 * NT + nt) * 64 + ln];
      for (int kp = 1; kp < ks; ++kp) {
        const f32x4 p = red[((size_t)kp * NT + nt) * 64 + ln];
        v0.x += p.x; v0.y += p.y; v0.z += p.z; v0.w += p.w;
      }
      if (NM == 2) {
        v1 = red[((size_t)ks * NT + nt) * 64 + ln];
        for (int kp = 1; kp < ks; ++kp) {
          const f32x4 p = red[((size_t)(ks + kp) * NT + nt) * 64 + ln];
          v1.x += p.x; v1.y += p.y; v1.z += p.z; v1.w += p.w;
        }
      }
    }
    const int tok = 16 * nt + (ln & 15);
    if (tok >= a.T) continue;
    const int r = 4 * (ln >> 4);  // row offset inside the tile
    if (EPI == KH_PG_QKV) {
      // bias after the matmul, before RoPE (matmul.cpp:74-77); RoPE itself: k_pg_rope
      int which = 0, rr = row0 + r;
      if (rr >= a.rows0 + a.rows1) {
        which = 2;
        rr -= a.rows0 + a.rows1;
      } else if (rr >= a.rows0) {
        which = 1;
        rr -= a.rows0;
      }
      const float* bias = a.w[which].bias;
      if (bias) {
        const f32x4 bv = *(const f32x4*)(bias + rr);
        v0.x += bv.x; v0.y += bv.y; v0.z += bv.z; v0.w += bv.w;
      }
      float* dst = which == 0 ? a.out + (size_t)tok * a.ldo
                              : (which == 1 ? a.kc : a.vc) + (size_t)(a.pos0 + tok) * a.rows1;
      *(f32x4*)(dst + rr) = v0;
    } else if (EPI == KH_PG_RESID) {
      f32x4* dst = (f32x4*)(a.out + (size_t)tok * a.ldo + row0 + r);
      f32x4 x = *dst;
      x.x += v0.x; x.y += v0.y; x.z += v0.z; x.w += v0.w;  // residual add (llama3.cpp:686,719)
      *dst = x;
    } else {
      f32x4 o;
      o.x = swiglu1(v0.x, v1.x);
      o.y = swiglu1(v0.y, v1.y);
      o.z = swiglu1(v0.z, v1.z);
      o.w = swiglu1(v0.w, v1.w);
      *(f32x4*)(a.out + (size_t)tok * a.ldo + row0 + r) = o;
    }
  }
}
static inline size_t pg_lds_bytes(int waves, int nt) {
  return waves <= 1 ? 0 : (size_t)waves * nt * 64 * sizeof(f32x4);
}

// ---- the small per-token kernels between the GEMMs ---------------------------------------------
// Xn[t] = w * (x[t] / sqrt(mean(x[t]^2) + eps))   (cpu/rmsnorm_kernel.cpp:24-32), one workgroup/token
__global__ __launch_bounds__(KH_WG) void k_pg_rmsnorm(const float* __restrict__ X,
                                                      const float* __restrict__ w,
                                                      float* __restrict__ Xn, int dim, float eps) {
  __shared__ float red[KH_WAVES_MAX];
  const f32x4* x4 = (const f32x4*)(X + (size_t)blockIdx.x * dim);
  const f32x4* w4 = (const f32x4*)w;
  f32x4* o4 = (f32x4*)(Xn + (size_t)blockIdx.x * dim);
  const int n4 = dim >> 2;
  float ss = 0.f;
  for (int k = threadIdx.x; k < n4; k += KH_WG) {
    const f32x4 v = x4[k];
    ss = fma4(v, v, ss);
  }
  ss = block_sum(ss, red);
  const float rs = 1.0f / sqrtf(ss / (float)dim + eps);
  for (int k = threadIdx.x; k < n4; k += KH_WG) {
    const f32x4 v = x4[k], g = w4[k];
    f32x4 o;
    o.x = g.x * (rs * v.x);
    o.y = g.y * (rs * v.y);
    o.z = g.z * (rs * v.z);
    o.w = g.w * (rs * v.w);
    o4[k] = o;
  }
}

// RoPE of the T query rows and of the T fresh key rows, in place (cpu/rope_kernel.cpp:18-42 half,
// :98-121 interleaved); token t sits at position pos0 + t.  blockIdx.x = token.
__global__ __launch_bounds__(KH_WG) void k_pg_rope(float* __restrict__ Q, float* __restrict__ kc,
                                                   const float* __restrict__ sin_cache,
                                                   const float* __restrict__ cos_cache, int dim,
                                                   int kv_dim, int hs, int pos0, int mode) {
  const int t = blockIdx.x, pos = pos0 + t;
  float* q = Q + (size_t)t * dim;
  float* k = kc + (size_t)pos * kv_dim;
  const float* sn = sin_cache + (size_t)pos * hs;
  const float* cs = cos_cache + (size_t)pos * hs;
  const int npq = dim >> 1, npk = kv_dim >> 1, half = hs >> 1;
  for (int p = threadIdx.x; p < npq + npk; p += KH_WG) {
    float* v = p < npq ? q : k;
    const int pp = p < npq ? p : p - npq;
    int r0, r1, cidx;
    if (mode == KH_ROPE_HALF) {
      const int head = pp / half, j = pp - head * half;
      r0 = head * hs + j;
      r1 = r0 + half;
      cidx = 2 * j;
    } else {
      r0 = 2 * pp;
      r1 = r0 + 1;
      cidx = r0 % hs;
    }
    const float fci = sn[cidx], fcr = cs[cidx];
    const float v0 = v[r0], v1 = v[r1];
    v[r0] = v0 * fcr - v1 * fci;
    v[r1] = v0 * fci + v1 * fcr;
  }
}
