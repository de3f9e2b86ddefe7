// kh_gemm.h — prompt prefill as real GEMMs on the matrix cores.
//
// The reference feeds the prompt one token per forward pass (demo/main.cpp:20-22): every prompt
// token streams all weights.  kh_prefill.h already shares one weight pass between 8 (fp32) / 4
// (int8) tokens on the VALU; here up to KH_PG_TMAX = 512 prompt tokens share ONE pass and the
// contraction runs on MFMA:  C[rows, T] = W[rows, K] . Xn[T, K]^T  with
//     v_mfma_f32_16x16x4_f32   (f32 in, f32 accumulate: bit-for-bit a k-ordered fmaf chain,
//                               MI355X guide §3 — exact fp32, no reduced-precision path)
// Tokens sit on the MFMA N dimension, weight rows on M.  Arithmetic intensity at T = 128 is
// 64 flop per weight byte (fp32) against a machine balance of ~22, so the GEMMs are MFMA-bound
// (157 TF f32 peak) and the weights cross HBM once per 128 tokens.
//
// Mapping (one wave = R 16-row weight tiles x NT 16-token tiles x a K slice):
//   * A fragment of 16x16x4: lane (i = l&15, h = l>>4) supplies W[row i][k].  The lane loads ONE
//     float4 = W[i][16b + 4h .. +4] per 16-column block b straight from HBM into VGPRs (non-temporal)
//     and feeds its four components to four MFMAs; MFMA #s therefore contracts the k-quad
//     {16b+s, 16b+4+s, 16b+8+s, 16b+12+s} - any k order is as good as any other for a dot product,
//     and this one makes both operands plain 16-byte loads.
//   * B fragment: lane (j = l&15, h) supplies Xn[token j][same k], one float4 per token tile per
//     block.  The activation slabs the GEMMs read are written by their producers (k_pg_rmsnorm, the
//     SwiGLU epilogue) in a TILED layout [K/16][128 tokens][16] so that such a fragment load is one
//     contiguous KiB (8 full cache lines) instead of 16 half lines of a row-major slab; the first
//     version read row-major slabs and sat at 33 % MFMA utilisation with the waves stalled on
//     issue - the 16x16 tile re-reads every activation element once per 16 weight rows, i.e.
//     32 B/clk/CU through the vector L1 at the full MFMA rate.  Two measures cut that: the tiled
//     layout, and register blocking R = 2 (32 weight rows per wave reuse each B fragment twice).
//   * C/D: lane (j, hq = l>>4) holds rows 4hq..4hq+3 of a tile for token j: a float4 of four
//     consecutive output rows of one token, which is exactly what the epilogues store.
//   * a workgroup = ks waves that split K (fixed-order LDS reduction, deterministic), times two
//     for the (w1, w3) SwiGLU pair; blockIdx.y selects a slice of 16*NT tokens.  (R, NT, ks, kz) are
//     chosen per GEMM by the cost model of kh_model_prefill.hip::pg_shape (one wave per SIMD for
//     fp32, two or three for int8; launches above 256 workgroups one workgroup per CU at a time).
//   * operand feeding: the fp32 loops request weights AND activations of block b + D together when
//     block b has been consumed (pg_kloop_f32_ring, D = 8 for the small tiles, 2 for the (2,8) tile);
//     the int8 loops - and fp32 K ranges a ring does not divide - take the weights in phases through
//     two register rings with the activation operand ping-ponging one step ahead (pg_phases).
//   * blockIdx.z (residual GEMMs only) = K slice ACROSS workgroups: partial rows, added by the
//     RMSNorm kernel that follows (KhPgGemmArgs::part).
// int8 (group 64): the lane's 16-byte load is 16 consecutive weights of one 64-group; they are
// converted once to scale * float(w) - the reference's per-element dequant (cuda/matmul_kernel.cu:73)
// - and feed 16 MFMAs per token tile.  Its tiled activation layout is [K/64][4][128][4][4] (the
// quarter index outermost inside a 64-column block) for the same contiguity.
//
// Numerics: not bit-identical to the decode GEMVs (different summation order); parity is held to the
// fp32 tolerance against the oracle (tests/test_model_gpu.py::test_gemm_prefill_*).  The bit-exact
// B-token path stays available (kh_model_prefill / KH_PREFILL=gemv).
#pragma once
#include "kh_fused.h"

// Prompt tokens per weight pass.  128 tokens (8 MFMA token tiles) already put every GEMM above the
// ridge; the reason to go further is the SMALL-M GEMMs (wo, w2, QKV): with 2048 rows x 128 tokens a
// chip-filling launch is left with (1,4) / (2,2) register tiles, 512 tokens give them the (2,8)
// tile of the SwiGLU GEMM in 256 workgroups (pg_shape).  The tiled slabs of a chunk use a token
// stride `tcap` = its token count rounded up to KH_PG_TSTEP, so a 128-token chunk has exactly the
// layout it always had.
#define KH_PG_TMAX 512
#define KH_PG_TSTEP 128
#define KH_PG_KZ_MAX 4         // K slices across workgroups of a residual GEMM (partial rows, see KhPgGemmArgs)
// Uniform-ring depth for the (2,8) register tile (64 MFMAs per block; pg_kloop_f32_ring; 0 = the phase scheme).
// Measured (profiles/r3_prefill_ring28.txt): depth 2 against the phases 0 ... +3 % over four fp32 models
// (TinyLlama 36.3 -> 37.3 k at 128 tokens, Llama-2-7B fp32 6.97 -> 7.27 k, Llama-3.2-1B unchanged), depth 4
// about half of that (256 VGPRs), depth 3 does not divide the block counts.
#ifndef KH_PG_RING_D16
#define KH_PG_RING_D16 2
#endif
// Workgroup width: <= 8 waves.  The fp32 shapes use 4 (ONE wave per SIMD, see pg_shape); int8 up to 8
// (its dequant VALU work wants a partner wave on the SIMD to keep the matrix pipe busy).
#define KH_PG_WG_MAX_F32 512
#define KH_PG_WG_MAX_Q8 512
#define KH_PG_WG_MAX(QUANT) ((QUANT) ? KH_PG_WG_MAX_Q8 : KH_PG_WG_MAX_F32)

enum { KH_PG_QKV = 0, KH_PG_RESID = 1, KH_PG_SWIGLU = 2 };

// position (in floats) of activation element (token t, column k) in a tiled slab
__host__ __device__ __forceinline__ size_t pg_tiled_index(bool quant, int k, int t, int tcap) {
  if (!quant) return ((size_t)(k >> 4) * tcap + t) * 16 + (k & 15);
  const int b = k >> 6, r = k & 63, h = r >> 4, q = (r >> 2) & 3, e = r & 3;
  return ((((size_t)b * 4 + q) * tcap + t) * 4 + h) * 4 + e;
}
static inline int pg_tcap(int T) { return (T + KH_PG_TSTEP - 1) / KH_PG_TSTEP * KH_PG_TSTEP; }

struct KhPgGemmArgs {
  KhLin w[3];        // QKV: wq, wk, wv ; RESID: w[0] ; SWIGLU: w1, w3
  const float* B;    // activation slab of tcap token rows (rows >= T hold finite garbage)
  float* out;        // QKV: Q [T][ldo] ; RESID: X [T][ldo] (+=) ; SWIGLU: H (tiled slab)
  float* kc;         // QKV: K cache rows of this layer, row (pos0 + t) * kv_dim
  float* vc;
  int rows0, rows1;  // QKV: rows of wq, rows of wk (= rows of wv); else rows0 = rows
  int ldo, K, T, pos0, gshift;
  int b_tiled;       // B is a tiled slab (else row-major [T][K])
  int tcap;          // token stride of the tiled slabs (B when b_tiled, the SwiGLU output): pg_tcap(T)
  // RESID only: K split ACROSS workgroups (gridDim.z = kz > 1).  Slice z stores its partial rows to
  // part[z][tok][ldo] (plain stores, no read-modify-write); the RMSNorm kernel that follows adds the
  // kz partials to X in fixed order (k_pg_rmsnorm) - no ticket, no fence, deterministic.
  float* part;
  // QKV only: RoPE fused into the epilogue (0 = off: k_pg_rope runs afterwards).  KH_PG_ROPE_PAIRS:
  // the pair (2i, 2i+1) sits in one lane's float4 (interleaved mode, cpu/rope_kernel.cpp:98-121).
  // KH_PG_ROPE_TILES: half mode (rope_kernel.cpp:18-42) pairs row j with j + hs/2 - a wave's R = 2
  // tiles are then tile t and tile t + hs/32 of the same head instead of two neighbours, so both
  // partners sit in the same lane and register index.
  int rope, head_size;
  const float* sin_cache;
  const float* cos_cache;
};
enum { KH_PG_ROPE_OFF = 0, KH_PG_ROPE_PAIRS = 1, KH_PG_ROPE_TILES = 2 };

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float pg_comp(const f32x4& v, int c) {  // c is a compile-time constant
  return c == 0 ? v.x : (c == 1 ? v.y : (c == 2 ? v.z : v.w));
}

// ---- K loop ------------------------------------------------------------------------------------------
// A wave has ONE in-order counter for its vector-memory loads (s_waitcnt vmcnt): a wait for a young
// L2-latency load (the activation operand of the next step) also waits for every OLDER load, in
// particular for HBM-latency weight loads issued before it.  A ring that refills one weight block
// per iteration therefore stalls every iteration for an HBM round trip.  Here the weights arrive in
// PHASES: two register rings of RING blocks; while ring `ph` is consumed, the other ring is
// requested in ONE batch right behind the first activation prefetch of the phase, so its latency is
// exposed at most once per phase instead of once per block.  The activation operand ping-pongs
// between two register sets (no copies: a copy makes the compiler wait for a prefetch as soon as it
// is issued).
template <int RING, int SUB /* operand sub-steps per block: 1 fp32, 4 int8 */, class LoadA,
          class LoadB, class Mfma>
__device__ __forceinline__ void pg_phases(int b0, int b1, LoadA&& load_a, LoadB&& load_b,
                                          Mfma&& mfma_sub) {
  static_assert((RING * SUB) % 2 == 0, "ping-pong parity must restart with every phase");
  const int last = b1 - 1;
  load_a(0, b0);  // ring 0 <- blocks b0 .. b0+RING-1
  load_b(0, b0, 0);
  for (int b = b0; b < b1; b += 2 * RING) {
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int base = b + ph * RING;
      if (base < b1) {  // wave-uniform (b0, b1 are scalars)
#pragma unroll
        for (int d = 0; d < RING; ++d) {
          const int bb = base + d;
          if (bb < b1) {
#pragma unroll
            for (int q = 0; q < SUB; ++q) {
              const int cur = (d * SUB + q) & 1;
              // next sub-step's activations first (young, L2 latency) ...
              const int nb_ = q + 1 < SUB ? bb : (bb + 1 < last ? bb + 1 : last);
              load_b(cur ^ 1, nb_, q + 1 < SUB ? q + 1 : 0);
              // ... then, once per phase, the whole next weight ring (HBM latency)
              if (d == 0 && q == 0) load_a(1 - ph, base + RING < last ? base + RING : last);
              __builtin_amdgcn_sched_barrier(0);  // the prefetches are issued before the MFMAs
              mfma_sub(ph, d, q, cur);
            }
          }
        }
      }
    }
  }
}

struct PgBAddr {           // activation operand addressing: base + nt*s_nt + block*s_b + sub*s_q
  const float* base;
  size_t s_nt, s_b, s_q;   // in floats
};

// fp32 weights: blocks of 16 columns; lane (i, h) owns W[i][16b + 4h .. +4] of each of its R tiles
template <int R, int NT>
__device__ __forceinline__ void pg_kloop_f32(const float* const (&wrow)[R], const PgBAddr& B, int b0,
                                             int b1, f32x4 (&acc)[R][NT]) {
  // blocks per weight ring (two rings per wave).  Deeper rings were tried for the one-wave-per-SIMD
  // shapes - 2 x 32 / 16 blocks (the excess lands in AGPRs: 4.04 ms per 128-token prefill) and
  // 2 x 16 / 8 within the 256 architectural VGPRs (3.76 ms) - against 3.71 ms with these.
  constexpr int RING = (R == 2 && NT >= 4) ? 4 : 8;
  f32x4 a[2][RING][R];
  f32x4 xb[2][NT];
  const int last = b1 - 1;
  auto load_a = [&](int ring, int base) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < RING; ++d) {
      const int bb = base + d < last ? base + d : last;
#pragma unroll
      for (int r = 0; r < R; ++r) a[ring][d][r] = ld_nt((const f32x4*)(wrow[r] + (size_t)bb * 16));
    }
  };
  auto load_b = [&](int set, int bb, int) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xb[set][nt] = *(const f32x4*)(B.base + nt * B.s_nt + (size_t)bb * B.s_b);
  };
  auto sub = [&](int ph, int d, int, int set) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[r][nt] = mfma16(pg_comp(a[ph][d][r], c), pg_comp(xb[set][nt], c), acc[r][nt]);
  };
  pg_phases<RING, 1>(b0, b1, load_a, load_b, sub);
}

// Uniform ring (small register tiles): weights AND activations of block b + D are requested together
// when block b has been consumed, D blocks ahead of their use.  With both operand streams at the same
// distance the in-order vmcnt couples nothing: the wait for block b leaves (D - 1) x (R + NT) loads in
// flight, and D x (MFMA time of a block) covers the HBM latency of the weights as well as the L2
// latency of the activations.  The phase scheme above prefetches the activations ONE step ahead -
// enough when a step holds >= 64 MFMAs (R x NT = 16), far too little for the 16-MFMA steps of the
// (1,4) / (2,2) tiles the small-M GEMMs (wo, w2, QKV) need to fill the chip (0.24 us per step against
// ~0.5 us of L2 latency: MfmaUtil 28-33 % in round 2).  Needs (b1 - b0) % D == 0 and >= 2 D blocks;
// D * (R + NT) float4 of ring registers.
template <int R, int NT, int D>
__device__ __forceinline__ void pg_kloop_f32_ring(const float* const (&wrow)[R], const PgBAddr& B, int b0,
                                                  int b1, f32x4 (&acc)[R][NT]) {
  f32x4 a[D][R];
  f32x4 xb[D][NT];
  auto load = [&](int slot, int bb) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < R; ++r) a[slot][r] = ld_nt((const f32x4*)(wrow[r] + (size_t)bb * 16));
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) xb[slot][nt] = *(const f32x4*)(B.base + nt * B.s_nt + (size_t)bb * B.s_b);
    __builtin_amdgcn_sched_barrier(0);  // slots stay in issue order (vmcnt retires in order)
  };
  auto mfma = [&](int slot) __attribute__((always_inline)) {
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[r][nt] = mfma16(pg_comp(a[slot][r], c), pg_comp(xb[slot][nt], c), acc[r][nt]);
  };
#pragma unroll
  for (int d = 0; d < D; ++d) load(d, b0 + d);
  for (int b = b0; b < b1 - D; b += D) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
      mfma(d);
      load(d, b + d + D);
    }
  }
#pragma unroll
  for (int d = 0; d < D; ++d) mfma(d);  // drain: the last D blocks, nothing left to request
}

// int8 group-64 weights: blocks of 64 columns.  Lane (i, h) owns the 16 weights
// W8[i][64b + 16h .. +16] (one dwordx4, one group -> one scale); quarter q of them pairs with
// Xn[token][64b + 16h + 4q .. +4].
template <int R, int NT>
__device__ __forceinline__ void pg_kloop_q8(const int8_t* const (&wrow)[R],
                                            const float* const (&srow)[R], const PgBAddr& B, int b0,
                                            int b1, f32x4 (&acc)[R][NT]) {
  constexpr int RING = 2;
  i32x4 qw[2][RING][R];
  float sc[2][RING][R];
  f32x4 xb[2][NT];
  const int last = b1 - 1;
  auto load_a = [&](int ring, int base) __attribute__((always_inline)) {
#pragma unroll
    for (int d = 0; d < RING; ++d) {
      const int bb = base + d < last ? base + d : last;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        qw[ring][d][r] = ld_nt((const i32x4*)(wrow[r] + (size_t)bb * 64));
        sc[ring][d][r] = srow[r][bb];
      }
    }
  };
  auto load_b = [&](int set, int bb, int q) __attribute__((always_inline)) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
      xb[set][nt] = *(const f32x4*)(B.base + nt * B.s_nt + (size_t)bb * B.s_b + (size_t)q * B.s_q);
  };
  auto sub = [&](int ph, int d, int q, int set) __attribute__((always_inline)) {
    float wf[R][4];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const float s = sc[ph][d][r];
      const i32x4 qv = qw[ph][d][r];
      const int dw = q == 0 ? qv.x : (q == 1 ? qv.y : (q == 2 ? qv.z : qv.w));
      // dequantised weight = scale * float(w8): the reference's per-element form
      wf[r][0] = s * (float)(int8_t)(dw & 0xff);
      wf[r][1] = s * (float)(int8_t)((dw >> 8) & 0xff);
      wf[r][2] = s * (float)(int8_t)((dw >> 16) & 0xff);
      wf[r][3] = s * (float)(dw >> 24);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
          acc[r][nt] = mfma16(wf[r][c], pg_comp(xb[set][nt], c), acc[r][nt]);
  };
  pg_phases<RING, 4>(b0, b1, load_a, load_b, sub);
}

// blockDim.x = NM * ks * 64 (NM = 2 for SWIGLU); blockIdx.x = (16*R)-row tile; blockIdx.y = slice of
// 16*NT tokens.  LDS: [waves][NT][64] float4 partial tiles (skipped when one wave owns the whole K).
template <bool QUANT, int R, int NT, int EPI>
__global__ __launch_bounds__(KH_PG_WG_MAX(QUANT)) void k_pg_gemm(const KhPgGemmArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  constexpr int NM = EPI == KH_PG_SWIGLU ? 2 : 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int nw = kh_nwaves();
  const int ks = nw / NM;
  const int mat = wave / ks, kpart = wave - mat * ks;
  const int i = lane & 15, h = lane >> 4;
  const int K = a.K;
  // first row of the wave's tile r in the stacked output: neighbours, or (half-mode fused RoPE) the
  // tile and its rotation partner hs/2 rows further down the same head
  int row0 = (int)blockIdx.x * (16 * R), rstep = 16;
  if (EPI == KH_PG_QKV && R == 2 && NT <= 4 && a.rope == KH_PG_ROPE_TILES) {
    const int nb = a.head_size >> 5;  // workgroups per head
    const int g = (int)blockIdx.x / nb;
    row0 = g * a.head_size + ((int)blockIdx.x - g * nb) * 16;
    rstep = a.head_size >> 1;
  }
  const int tok0 = (int)blockIdx.y * (16 * NT);
  // weight matrix of this workgroup (its tiles never straddle two matrices: row counts are multiples
  // of 16 * R - of the head size with paired tiles -, checked by the host)
  KhLin W = a.w[mat];
  int wr0 = row0;
  if (EPI == KH_PG_QKV) {
    if (row0 >= a.rows0 + a.rows1) {
      W = a.w[2];
      wr0 = row0 - a.rows0 - a.rows1;
    } else if (row0 >= a.rows0) {
      W = a.w[1];
      wr0 = row0 - a.rows0;
    }
  }
  f32x4 acc[R][NT];
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) acc[r][nt] = f32x4{0.f, 0.f, 0.f, 0.f};
  // (the K range depends on the wave index only: keep it in scalar registers so the block guards
  // of the K loop are scalar branches)
  PgBAddr B;
  if (!QUANT) {
    const int nb = K >> 4;
    const int z0 = (int)((long)blockIdx.z * nb / gridDim.z), z1 = (int)((long)(blockIdx.z + 1) * nb / gridDim.z);
    const int b0 = __builtin_amdgcn_readfirstlane(z0 + (int)((long)kpart * (z1 - z0) / ks));
    const int b1 = __builtin_amdgcn_readfirstlane(z0 + (int)((long)(kpart + 1) * (z1 - z0) / ks));
    const float* wrow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) wrow[r] = (const float*)W.w + (size_t)(wr0 + rstep * r + i) * K + 4 * h;
    if (a.b_tiled) {
      B = PgBAddr{a.B + (size_t)(tok0 + i) * 16 + 4 * h, 256, (size_t)a.tcap * 16, 0};
    } else {
      B = PgBAddr{a.B + (size_t)(tok0 + i) * K + 4 * h, (size_t)16 * K, 16, 0};
    }
    // the uniform ring (both operands D blocks ahead) when the wave's block count allows it: 8 blocks
    // deep for the small register tiles, KH_PG_RING_D16 for the big (2, 8) tile; else the phase scheme
    constexpr int RING_D = R * NT <= 8 ? 8 : KH_PG_RING_D16;
    if constexpr (RING_D > 0) {
      if (b1 - b0 >= 2 * RING_D && (b1 - b0) % RING_D == 0)
        pg_kloop_f32_ring<R, NT, RING_D>(wrow, B, b0, b1, acc);
      else if (b1 > b0)
        pg_kloop_f32<R, NT>(wrow, B, b0, b1, acc);
    } else if (b1 > b0) {
      pg_kloop_f32<R, NT>(wrow, B, b0, b1, acc);
    }
  } else {
    const int nb = K >> 6;
    const int z0 = (int)((long)blockIdx.z * nb / gridDim.z), z1 = (int)((long)(blockIdx.z + 1) * nb / gridDim.z);
    const int b0 = __builtin_amdgcn_readfirstlane(z0 + (int)((long)kpart * (z1 - z0) / ks));
    const int b1 = __builtin_amdgcn_readfirstlane(z0 + (int)((long)(kpart + 1) * (z1 - z0) / ks));
    const int8_t* wrow[R];
    const float* srow[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
      wrow[r] = (const int8_t*)W.w + (size_t)(wr0 + rstep * r + i) * K + 16 * h;
      srow[r] = W.scales + (size_t)(wr0 + rstep * r + i) * nb;
    }
    if (a.b_tiled) {
      B = PgBAddr{a.B + (size_t)(tok0 + i) * 16 + 4 * h, 256, (size_t)4 * a.tcap * 16,
                  (size_t)a.tcap * 16};
    } else {
      B = PgBAddr{a.B + (size_t)(tok0 + i) * K + 16 * h, (size_t)16 * K, 64, 4};
    }
    if (b1 > b0) pg_kloop_q8<R, NT>(wrow, srow, B, b0, b1, acc);
  }
  // ---- combine the K slices (fixed order) and run the epilogue on float4 = 4 rows x 1 token,
  //      one 16-row tile at a time (the LDS area holds one tile row of partials) -------------------
  f32x4* red = (f32x4*)smem_raw;
  const bool direct = nw == 1;
  // fused half-mode RoPE: tile 0's rows wait here for their partners in tile 1 (NT <= 4 only: with
  // eight token tiles the 32 extra registers push the epilogue into scratch; the host then keeps k_pg_rope)
  constexpr bool TILE_ROPE = EPI == KH_PG_QKV && R == 2 && NT <= 4;
  f32x4 keep[TILE_ROPE ? NT : 1];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    if (!direct) {
      if (r > 0) __syncthreads();
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) red[((size_t)wave * NT + nt) * 64 + lane] = acc[r][nt];
      __syncthreads();
    }
#pragma unroll
    for (int it = 0; it < NT; ++it) {  // NT * 64 elements over >= 64 threads (static trip count: keep[])
      const int e = (int)threadIdx.x + it * kh_wg();
      if (e >= NT * 64) break;
      const int nt = e >> 6, ln = e & 63;
      f32x4 v0, v1 = f32x4{0.f, 0.f, 0.f, 0.f};
      if (direct) {
        v0 = acc[r][0];
#pragma unroll
        for (int t = 1; t < NT; ++t)
          if (t == nt) v0 = acc[r][t];
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
      const int tok = tok0 + 16 * nt + (ln & 15);
      if (tok >= a.T) continue;  // (padding tokens: nothing kept, nothing stored)
      const int orow = row0 + rstep * r + 4 * (ln >> 4);  // first of the 4 output rows
      if (EPI == KH_PG_QKV) {
        // bias after the matmul, before RoPE (matmul.cpp:74-77)
        int which = 0, rr = orow;
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
        const int hs = a.head_size;
        const float* sn = a.sin_cache + (size_t)(a.pos0 + tok) * hs;
        const float* cs = a.cos_cache + (size_t)(a.pos0 + tok) * hs;
        if (which < 2 && a.rope == KH_PG_ROPE_PAIRS) {
          // interleaved: the float4 holds pairs (rr, rr+1), (rr+2, rr+3); cache column = row in head
          const int c = rr % hs;
          const float s0 = sn[c], c0 = cs[c], s1 = sn[c + 2], c1 = cs[c + 2];
          f32x4 o;
          o.x = v0.x * c0 - v0.y * s0;
          o.y = v0.x * s0 + v0.y * c0;
          o.z = v0.z * c1 - v0.w * s1;
          o.w = v0.z * s1 + v0.w * c1;
          *(f32x4*)(dst + rr) = o;
        } else if (TILE_ROPE && which < 2 && a.rope == KH_PG_ROPE_TILES) {
          if (r == 0) {
            keep[TILE_ROPE ? it : 0] = v0;  // rows head*hs + j .. + 3 (j < hs/2): rotated when tile 1 arrives
          } else {
            // this tile holds the partners, rows head*hs + hs/2 + j .. + 3; cache column 2 * j
            const int half = hs >> 1, j = (rr - half) % hs;
            const f32x4 lo = keep[TILE_ROPE ? it : 0];
            f32x4 olo, ohi;
            const float s0 = sn[2 * j], c0 = cs[2 * j], s1 = sn[2 * j + 2], c1 = cs[2 * j + 2];
            const float s2 = sn[2 * j + 4], c2 = cs[2 * j + 4], s3 = sn[2 * j + 6], c3 = cs[2 * j + 6];
            olo.x = lo.x * c0 - v0.x * s0; ohi.x = lo.x * s0 + v0.x * c0;
            olo.y = lo.y * c1 - v0.y * s1; ohi.y = lo.y * s1 + v0.y * c1;
            olo.z = lo.z * c2 - v0.z * s2; ohi.z = lo.z * s2 + v0.z * c2;
            olo.w = lo.w * c3 - v0.w * s3; ohi.w = lo.w * s3 + v0.w * c3;
            *(f32x4*)(dst + rr - half) = olo;
            *(f32x4*)(dst + rr) = ohi;
          }
        } else {
          *(f32x4*)(dst + rr) = v0;
        }
      } else if (EPI == KH_PG_RESID) {
        if (gridDim.z > 1) {  // K slice of a cross-workgroup split: the following k_pg_rmsnorm adds it
          *(f32x4*)(a.part + ((size_t)blockIdx.z * a.tcap + tok) * a.ldo + orow) = v0;
        } else {
          f32x4* dst = (f32x4*)(a.out + (size_t)tok * a.ldo + orow);
          f32x4 x = *dst;
          x.x += v0.x; x.y += v0.y; x.z += v0.z; x.w += v0.w;  // residual add (llama3.cpp:686,719)
          *dst = x;
        }
      } else {
        f32x4 o;
        o.x = swiglu1(v0.x, v1.x);
        o.y = swiglu1(v0.y, v1.y);
        o.z = swiglu1(v0.z, v1.z);
        o.w = swiglu1(v0.w, v1.w);
        *(f32x4*)(a.out + pg_tiled_index(QUANT, orow, tok, a.tcap)) = o;  // H feeds the w2 GEMM: tiled slab
      }
    }
  }
}
// dynamic-LDS request that admits ONE workgroup per CU (more than half of the 160 KiB)
#define KH_PG_SOLO_LDS ((size_t)81 * 1024)
static inline size_t pg_lds_bytes(int waves, int nt) {
  return waves <= 1 ? 0 : (size_t)waves * nt * 64 * sizeof(f32x4);
}

// ---- the small per-token kernels between the GEMMs ---------------------------------------------
// Xn[t] = w * (x[t] / sqrt(mean(x[t]^2) + eps))   (cpu/rmsnorm_kernel.cpp:24-32), one workgroup per
// token; Xn is written as the tiled slab the QKV / FFN GEMMs read.  kz > 0: the residual GEMM in front
// of it split K across workgroups - its kz partial rows part[z][t][dim] are added to X first, in fixed
// order (the residual add of llama3.cpp:686,719, deferred).
template <bool QUANT>
__global__ __launch_bounds__(KH_WG) void k_pg_rmsnorm(float* __restrict__ X,
                                                      const float* __restrict__ w,
                                                      float* __restrict__ Xn, int dim, float eps,
                                                      int tcap, const float* __restrict__ part, int kz) {
  __shared__ float red[KH_WAVES_MAX];
  const int t = blockIdx.x;
  f32x4* x4 = (f32x4*)(X + (size_t)t * dim);
  const f32x4* w4 = (const f32x4*)w;
  const int n4 = dim >> 2;
  // the row (after the deferred residual add) stays in registers between the two passes for dim <= 4096
  // (KEEP float4 per thread); longer rows re-read what this thread wrote
  constexpr int KEEP = 4;
  f32x4 keep[KEEP];
  float ss = 0.f;
#pragma unroll
  for (int i = 0; i < KEEP; ++i) {
    const int k = (int)threadIdx.x + i * KH_WG;
    if (k < n4) {
      f32x4 v = x4[k];
      if (kz > 0) {
        for (int z = 0; z < kz; ++z) {
          const f32x4 p = *(const f32x4*)(part + ((size_t)z * tcap + t) * dim + 4 * k);
          v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
        }
        x4[k] = v;
      }
      keep[i] = v;
      ss = fma4(v, v, ss);
    }
  }
  for (int k = (int)threadIdx.x + KEEP * KH_WG; k < n4; k += KH_WG) {
    f32x4 v = x4[k];
    if (kz > 0) {
      for (int z = 0; z < kz; ++z) {
        const f32x4 p = *(const f32x4*)(part + ((size_t)z * tcap + t) * dim + 4 * k);
        v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w;
      }
      x4[k] = v;  // (re-read by this same thread below)
    }
    ss = fma4(v, v, ss);
  }
  ss = block_sum(ss, red);
  const float rs = 1.0f / sqrtf(ss / (float)dim + eps);
  auto emit = [&](int k, const f32x4& v) __attribute__((always_inline)) {
    const f32x4 g = w4[k];
    f32x4 o;
    o.x = g.x * (rs * v.x);
    o.y = g.y * (rs * v.y);
    o.z = g.z * (rs * v.z);
    o.w = g.w * (rs * v.w);
    *(f32x4*)(Xn + pg_tiled_index(QUANT, 4 * k, t, tcap)) = o;
  };
#pragma unroll
  for (int i = 0; i < KEEP; ++i) {
    const int k = (int)threadIdx.x + i * KH_WG;
    if (k < n4) emit(k, keep[i]);
  }
  for (int k = (int)threadIdx.x + KEEP * KH_WG; k < n4; k += KH_WG) emit(k, x4[k]);
}

// RoPE of the T query rows and of the T fresh key rows, in place (cpu/rope_kernel.cpp:18-42 half,
// :98-121 interleaved); token t sits at position pos0 + t.  blockIdx.x = token.
static __global__ __launch_bounds__(KH_WG) void k_pg_rope(float* __restrict__ Q, float* __restrict__ kc,
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
