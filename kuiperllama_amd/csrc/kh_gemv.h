// kh_gemv.h — batch-1 GEMV cores for gfx950.
//
// Shape of the problem: y[K] = W[K,M] . x[M], W row-major as exported in the
// .bin (tools/export.py:79-131), streamed once per token => HBM-bound, 0.5 flop/B (fp32).
//
// Mapping: one WAVE owns a pair of rows at a time and streams them with 16-byte-per-lane
// non-temporal loads (64 lanes x 16 B = 1 KiB contiguous per instruction), U instructions per
// row in flight before the first FMA (2*U KiB per wave).  The activation vector lives in LDS
// (staged once per workgroup, optionally RMS-normalised on the way in); LDS read bandwidth is
// ~25x the per-CU HBM rate so re-reading x per row is free.  No LDS round trip for the
// weights (MI355X guide: "GEMV / M <= 16 decode weights ... load straight to VGPRs, deep
// unroll, late vmcnt").  The only cross-lane step is one DPP wave reduction per row pair.
//
// Row PAIRS are the unit because every fused epilogue consumes two outputs together:
// RoPE rotates (v0,v1), SwiGLU combines (w1.x, w3.x), and plain rows just take (2i, 2i+1).
//
// Latency structure (these kernels are 4-25 MB = a few HBM round trips long, so fixed costs
// dominate): a wave issues the loads of its FIRST row pair before the activation vector is
// staged (the weights do not depend on it: the x round trip through L2, the RMS reduction and
// the barrier overlap with the first HBM round trip), and issues the loads of its NEXT pair
// before reducing/storing the current one, so it always has 2*U KiB in flight.
#pragma once
#include "kh_common.h"

// ---------------------------------------------------------------------------------------------
// Register tiles: one chunk (U x 1 KiB) of two rows.
template <int U>
struct RegsF32 {
  f32x4 v0[U], v1[U];
};
template <int U>
struct RegsQ8 {
  i32x4 q0[U], q1[U];
  float g0[U], g1[U];
};

struct RowsF32 {
  const f32x4* w0;
  const f32x4* w1;
};
struct RowsQ8 {
  const i32x4* w0;
  const i32x4* w1;
  const float* sc0;  // scale of the row's first group (M % group == 0 on this path)
  const float* sc1;
};

// One 16-byte load per row per lane (u-th of the chunk) and its FMAs: the units of the rolling
// refill in gemv_pairs (registers of slot u are re-requested right after slot u was consumed).
template <int U>
__device__ __forceinline__ void load_u(RegsF32<U>& r, const RowsF32& rows, int c0, int M4, int lane,
                                       int u) {
  const int idx = c0 + u * KH_WAVE + lane;
  const int cidx = idx < M4 ? idx : 0;  // clamped address; masked in fma_u
  r.v0[u] = ld_nt(rows.w0 + cidx);
  r.v1[u] = ld_nt(rows.w1 + cidx);
  // the loads of a slot stay together and slots stay in order: the machine scheduler otherwise
  // regroups them (all scales first, then all weights, ...), and since vmcnt retires in order the
  // wait for slot 0 then covers most of the tile.  The SAME issue order in the prologue and in
  // the rolling loop also keeps the compiler's merged waitcnt state at the loop header exact.
  __builtin_amdgcn_sched_barrier(0);
}
template <int U>
__device__ __forceinline__ void fma_u(const RegsF32<U>& r, const f32x4* xs, int c0, int M4, int lane,
                                      int u, float& a0, float& a1) {
  // Lanes past the end of the column range read a clamped address (load_u: the row's FIRST float4) and
  // multiply by a zeroed x instead of branching around the FMAs: straight-line code keeps the compiler's
  // s_waitcnt placement exact (every exec-mask branch is a merge point where it turns conservative
  // and waits for most of the tile), and "+ w * 0" leaves the sum bit-identical - PROVIDED the first four
  // weights of every row (int8: the first group scale) are finite, which every exported .bin satisfies
  // (an Inf / NaN there would poison the row through 0 * Inf; the B-token prefill kernels of kh_prefill.h
  // branch instead; tests/test_model_gpu.py::test_prefill_is_bit_identical_to_token_by_token[qwen-bias] compares the two
  // bit for bit at dim 448 = 112 float4, i.e. with masked lanes in every row, and test_matmul_f32_vs_oracle runs M =
  // 896 / 288 / 36 / 20000 against the oracle).
  const int idx = c0 + u * KH_WAVE + lane;
  const bool in = idx < M4;
  f32x4 xv = xs[in ? idx : 0];
  xv.x = in ? xv.x : 0.f;
  xv.y = in ? xv.y : 0.f;
  xv.z = in ? xv.z : 0.f;
  xv.w = in ? xv.w : 0.f;
  a0 = fma4(r.v0[u], xv, a0);
  a1 = fma4(r.v1[u], xv, a1);
}
template <int U>
__device__ __forceinline__ void load_chunk(RegsF32<U>& r, const RowsF32& rows, int c0, int M4,
                                           int lane) {
#pragma unroll
  for (int u = 0; u < U; ++u) load_u<U>(r, rows, c0, M4, lane, u);
}
template <int U>
__device__ __forceinline__ void fma_chunk(const RegsF32<U>& r, const f32x4* xs, int c0, int M4,
                                          int lane, float& a0, float& a1) {
#pragma unroll
  for (int u = 0; u < U; ++u) fma_u<U>(r, xs, c0, M4, lane, u, a0, a1);
}

// int8 group-quantised rows (tools/export.py:134-210: int8[K*M] then fp32 scales[K*M/g]).
// A lane's dwordx4 = 16 weights of ONE group (group is a power of two >= 16 on this path), so
// one scale per load; dequant factored per 16-weight run:
//   sum_i x_i * s_g * w_i  ==  s_g * sum_i x_i * w_i      (reference: cuda/matmul_kernel.cu:73)
// Issue order per slot u: q0[u], q1[u], s0[u], s1[u].  vmcnt retires in order, so slot u can be
// consumed as soon as ITS four loads are back; with all weight loads first and all scale loads
// behind them (the round-1/2 order) the first FMA waited for the whole tile.
// (Scale handling measured in r2, profiles/r2_int8_convert_ab.txt + tools/mb_scale.hip: the
// per-load scale fetch costs exactly its bytes; one coalesced scale load + ds_bpermute was slower.)
template <int U>
__device__ __forceinline__ void load_u(RegsQ8<U>& r, const RowsQ8& rows, int gshift, int c0, int M16,
                                       int lane, int u) {
  const int idx = c0 + u * KH_WAVE + lane;
  const int cidx = idx < M16 ? idx : 0;
  r.q0[u] = ld_nt(rows.w0 + cidx);
  r.q1[u] = ld_nt(rows.w1 + cidx);
  const int gi = (cidx << 4) >> gshift;
  r.g0[u] = rows.sc0[gi];
  r.g1[u] = rows.sc1[gi];
  __builtin_amdgcn_sched_barrier(0);  // see the fp32 load_u
}
template <int U>
__device__ __forceinline__ void fma_u(const RegsQ8<U>& r, const f32x4* xs, int c0, int M16, int plane,
                                      int lane, int u, float& a0, float& a1) {
  const int idx = c0 + u * KH_WAVE + lane;
  {
    // out-of-range lanes: clamped addresses and a zeroed group scale (see the fp32 fma_u);
    // the int8 dot itself is finite, so scale 0 contributes exactly +0
    const bool in = idx < M16;
    const int ci = in ? idx : 0;
    const f32x4 x0 = xs[ci], x1 = xs[plane + ci], x2 = xs[2 * plane + ci], x3 = xs[3 * plane + ci];
    float t0 = 0.f, t1 = 0.f;
    t0 = dot4_i8(r.q0[u].x, x0, t0);
    t0 = dot4_i8(r.q0[u].y, x1, t0);
    t0 = dot4_i8(r.q0[u].z, x2, t0);
    t0 = dot4_i8(r.q0[u].w, x3, t0);
    t1 = dot4_i8(r.q1[u].x, x0, t1);
    t1 = dot4_i8(r.q1[u].y, x1, t1);
    t1 = dot4_i8(r.q1[u].z, x2, t1);
    t1 = dot4_i8(r.q1[u].w, x3, t1);
    a0 = __builtin_fmaf(in ? r.g0[u] : 0.f, t0, a0);
    a1 = __builtin_fmaf(in ? r.g1[u] : 0.f, t1, a1);
  }
}
template <int U>
__device__ __forceinline__ void load_chunk(RegsQ8<U>& r, const RowsQ8& rows, int gshift, int c0,
                                           int M16, int lane) {
#pragma unroll
  for (int u = 0; u < U; ++u) load_u<U>(r, rows, gshift, c0, M16, lane, u);
}
template <int U>
__device__ __forceinline__ void fma_chunk(const RegsQ8<U>& r, const f32x4* xs, int c0, int M16,
                                          int plane, int lane, float& a0, float& a1) {
#pragma unroll
  for (int u = 0; u < U; ++u) fma_u<U>(r, xs, c0, M16, plane, lane, u, a0, a1);
}

// ---------------------------------------------------------------------------------------------
// Uniform view over fp32 / int8 weight matrices so the kernels are written once.
template <bool QUANT, int U>
struct Gemv;

template <int U>
struct Gemv<false, U> {
  static constexpr int kU = U;
  using Regs = RegsF32<U>;
  using Rows = RowsF32;
  int Mc;  // chunks per row in 16-byte units: M/4
  __device__ __forceinline__ Gemv(int M, int /*gshift*/) : Mc(M >> 2) {}
  __device__ __forceinline__ Rows rows(const void* w0_base, int r0, const void* w1_base, int r1,
                                       const float*, const float*, int M) const {
    return Rows{(const f32x4*)((const float*)w0_base + (size_t)r0 * M),
                (const f32x4*)((const float*)w1_base + (size_t)r1 * M)};
  }
  // lim: end of this wave's column range in 16-byte units (== Mc unless the row is split)
  __device__ __forceinline__ void load(Regs& r, const Rows& rw, int c0, int lim, int lane) const {
    load_chunk<U>(r, rw, c0, lim, lane);
  }
  __device__ __forceinline__ void fma(const Regs& r, const f32x4* xs, int c0, int lim, int lane,
                                      float& a0, float& a1) const {
    fma_chunk<U>(r, xs, c0, lim, lane, a0, a1);
  }
  __device__ __forceinline__ void load1(Regs& r, const Rows& rw, int c0, int lim, int lane, int u) const {
    load_u<U>(r, rw, c0, lim, lane, u);
  }
  __device__ __forceinline__ void fma1(const Regs& r, const f32x4* xs, int c0, int lim, int lane, int u,
                                       float& a0, float& a1) const {
    fma_u<U>(r, xs, c0, lim, lane, u, a0, a1);
  }
};

template <int U>
struct Gemv<true, U> {
  static constexpr int kU = U;
  using Regs = RegsQ8<U>;
  using Rows = RowsQ8;
  int Mc;  // M/16
  int gshift;
  __device__ __forceinline__ Gemv(int M, int gs) : Mc(M >> 4), gshift(gs) {}
  __device__ __forceinline__ Rows rows(const void* w0_base, int r0, const void* w1_base, int r1,
                                       const float* s0_base, const float* s1_base, int M) const {
    const int gpr = M >> gshift;  // groups per row
    return Rows{(const i32x4*)((const int8_t*)w0_base + (size_t)r0 * M),
                (const i32x4*)((const int8_t*)w1_base + (size_t)r1 * M),
                s0_base + (size_t)r0 * gpr, s1_base + (size_t)r1 * gpr};
  }
  __device__ __forceinline__ void load(Regs& r, const Rows& rw, int c0, int lim, int lane) const {
    load_chunk<U>(r, rw, gshift, c0, lim, lane);
  }
  __device__ __forceinline__ void fma(const Regs& r, const f32x4* xs, int c0, int lim, int lane,
                                      float& a0, float& a1) const {
    fma_chunk<U>(r, xs, c0, lim, Mc + 1, lane, a0, a1);
  }
  __device__ __forceinline__ void load1(Regs& r, const Rows& rw, int c0, int lim, int lane, int u) const {
    load_u<U>(r, rw, gshift, c0, lim, lane, u);
  }
  __device__ __forceinline__ void fma1(const Regs& r, const f32x4* xs, int c0, int lim, int lane, int u,
                                       float& a0, float& a1) const {
    fma_u<U>(r, xs, c0, lim, Mc + 1, lane, u, a0, a1);
  }
};

// Pipelined row-pair loop shared by every GEMV kernel.
//   PAIR(p)   -> Rows for work item p                     (pure address arithmetic)
//   ISSUE()   -> issue the loads of the activation vector  (registers, no wait)
//   FINISH()  -> norm + write to LDS                        (contains the barriers)
//   EPI(p, s0, s1) -> epilogue with the two dot products   (called by every lane; lane 0 stores)
// Order of VMEM issue: x first, then the first weight chunk.  vmcnt retires in order, so the
// x values can be consumed (s_waitcnt vmcnt(#weight loads)) while the weights are still in
// flight; the reverse order would make the staging wait for the whole first chunk.
//   PRE(p)    -> small struct of epilogue operands (bias, sin/cos, residual) fetched EARLY, right
//                behind the pair's weight loads, so the epilogue has no dependent load left
//
// SPLIT (1, 2 or 4): waves of a workgroup that share one row pair, each streaming 1/SPLIT of the
// columns; partial sums are combined through LDS in fixed order (deterministic).  Used when a
// matrix has too few rows to put >= ~4096 waves in flight (w2: 1024 pairs of 32 KiB rows), where
// one wave per pair leaves 4 waves per CU and no load/compute overlap.  comb = LDS float[8].
// G: the matrix view - Gemv<QUANT, U>, or any class with its interface (kU, Regs, Rows, Mc, load,
// load1, fma, fma1; tools/mb_gemv_ladder.hip plugs in stream-only views to price each ingredient).
// ROLL: refill the tile slot by slot (slot u of the next tile requested right after slot u was consumed)
// or in one burst after the tile's FMAs; chosen per kernel from same-box A/Bs (kh_fused.h).
// Measured and not kept (profiles/r3_gemv_ladder_depth_roll.txt, commit 0f6b0a3^): a second register
// tile requested in the prologue as well (two work items in flight per wave) - no gain on any
// Llama-2-7B int8 shape, +70 VGPRs.
template <int SPLIT, bool ROLL, class G, class PairFn, class PreFn, class IssueFn, class FinishFn, class EpiFn>
__device__ __forceinline__ void gemv_pairs(const G& g, const f32x4* xs, int total, int lane,
                                           float* comb, PairFn&& PAIR, PreFn&& PRE, IssueFn&& ISSUE,
                                           FinishFn&& FINISH, EpiFn&& EPI) {
  constexpr int U = G::kU;
  const int vb = (int)blockIdx.x, vgrid = (int)gridDim.x;
  static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "SPLIT must be 1, 2 or 4");
  const int PPW = kh_nwaves() / SPLIT;  // pairs per workgroup per iteration
  // The wave index is uniform by construction.  Telling the compiler (readfirstlane) moves the
  // work-item arithmetic - pair index, row addresses, loop control - to the scalar unit: s_cbranch_scc
  // instead of exec masking.  (Round 2 measured it fp32-only, profiles/r2_scalar_wave_ab.txt; since the
  // single-loop form of round 3 every loop control below depends on it, so it is unconditional.)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int part = wave & (SPLIT - 1);
  const int gp = vb * PPW + wave / SPLIT;
  const int np = vgrid * PPW;
  const int step = KH_WAVE * U;
  // column quantum per part, multiple of 4 chunks so an int8 part starts on a group boundary
  const int Q = (((g.Mc + SPLIT - 1) / SPLIT) + 3) & ~3;
  const int cb = part * Q;
  const int ce = cb + Q < g.Mc ? cb + Q : g.Mc;
  const int p0 = gp < total ? gp : 0;
  typename G::Regs regs;
  // the x loads leave FIRST - before the work-item decode of PAIR (k_qkv: an integer division and
  // three-way selects, ~80 instructions the compiler otherwise hoists above them)
  ISSUE();
  __builtin_amdgcn_sched_barrier(0);  // keep the x loads ahead of everything else, the weight loads included
  typename G::Rows cur = PAIR(p0);
  // unconditional (an idle wave re-reads pair 0): a branch here would make the compiler merge
  // the vmcnt state of both paths and wait vmcnt(0) — i.e. for the weights — before using x.
  // The loads stay in flight across the staging barriers.
  g.load(regs, cur, cb, ce, lane);
  auto aux = PRE(p0);
  FINISH();
  KH_STAMP(1);
  const int iters = (total + np - 1) / np;  // uniform trip count: the SPLIT path has barriers
  // The tile registers ROLL: slot u of the next tile is requested right after slot u of the current
  // tile was consumed, so (U-1)/U of a tile stays in flight while the wave computes - with "consume
  // the whole tile, then request the next" (rounds 1-2) a wave had nothing in flight during its
  // FMAs, 20 % of the int8 kernels' cycle (profiles/r3_rolling_refill_ab.txt).  One chunk covers a
  // wave's column range in every BASELINE shape (pick_shape sizes U for it), so "next tile" is
  // normally the wave's next row pair.
  // reduction + epilogue of one work item (shared by the two loops below)
  auto finish_item = [&](int p, bool valid, float a0, float a1, const decltype(aux)& ax)
                         __attribute__((always_inline)) {
    float s0 = wave_sum(a0), s1 = wave_sum(a1);
    if constexpr (SPLIT == 1) {
      if (valid) EPI(p, s0, s1, ax);
    } else {
      if (lane == 0) {
        comb[2 * wave] = s0;
        comb[2 * wave + 1] = s1;
      }
      __syncthreads();
      if (valid && part == 0) {
        s0 = comb[2 * wave];
        s1 = comb[2 * wave + 1];
#pragma unroll
        for (int k = 1; k < SPLIT; ++k) {
          s0 += comb[2 * (wave + k)];
          s1 += comb[2 * (wave + k) + 1];
        }
        EPI(p, s0, s1, ax);
      }
      __syncthreads();
    }
  };
  // ONE loop over the wave's tiles (pair, chunk) in order, a single body that consumes slot u and
  // re-requests it for the following tile - the next chunk of the same rows or the first chunk of
  // the wave's next pair.  (A second loop body for multi-chunk rows next to a rolling one doubled
  // the VGPR count - two tile register sets plus phi copies that drain vmcnt at every latch.)
  // All control below is scalar (wave index is readfirstlane'd): s_cbranch_scc, no exec masking.
  int p = gp, c0 = cb, it = 0;
  float a0 = 0.f, a1 = 0.f;
  for (;;) {
    const bool valid = p < total;  // uniform per wave
    const int c1 = c0 + step;
    const bool last = c1 >= ce;    // last chunk of this wave's column range
    const int pn = last ? p + np : p;
    const int cn = last ? cb : c1;
    const bool more = pn < total;  // the wave has a following tile (implies valid)
    auto aux_next = aux;
    if (more) {
      typename G::Rows nxt = cur;
      if (last) {
        nxt = PAIR(pn);
        // epilogue operands of the next pair (bias, sin/cos, residual): requested AHEAD of its tile.
        // They are vector loads (the kernel has stored by now, so no scalar-cache path), and the
        // loop-carried copy of them at the latch waits for them: behind the tile they were the
        // youngest loads in the in-order queue and that wait was a vmcnt(0) drain of the whole next
        // tile in every iteration of k_qkv / k_gemv_res (rounds 1-2 and the first rolling version).
        aux_next = PRE(pn);
      }
      if constexpr (ROLL) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
          g.fma1(regs, xs, c0, ce, lane, u, a0, a1);
          g.load1(regs, nxt, cn, ce, lane, u);
        }
      } else {
        g.fma(regs, xs, c0, ce, lane, a0, a1);
        g.load(regs, nxt, cn, ce, lane);
      }
      cur = nxt;
    } else if (valid) {
      g.fma(regs, xs, c0, ce, lane, a0, a1);
    }
    if (last) {
      finish_item(p, valid, a0, a1, aux);
      if (it == 0) KH_STAMP(2);
      if (!more) KH_STAMP_W();  // the wave's last item (waves without one stamp their first pass)
      aux = aux_next;
      a0 = a1 = 0.f;
      if (++it == iters) break;
    }
    p = pn;
    c0 = cn;
  }
  KH_STAMP(3);
}
struct NoAux {};

// ---------------------------------------------------------------------------------------------
// Stage a vector into LDS (all threads of the workgroup).  With NORM the vector is the input of an RMSNorm
//   xn = w_norm * (x * rs),  rs = 1/sqrt(mean(x^2)+eps)          (cpu/rmsnorm_kernel.cpp:24-32)
// whose SCALE is applied behind the dot product: LDS receives g = w_norm * x as soon as x is there, the
// per-wave sums of squares travel through `red` under the SAME barrier that publishes g, every thread
// forms rs behind that barrier, and the epilogue of the GEMV multiplies:  W . xn = rs * (W . g).
// One barrier between the vector's arrival and the first FMA instead of three (the block sum's two + the
// LDS write's), nothing but a multiply in front of the LDS write.  The decode kernels, the B-token
// prefill (kh_prefill.h) and the int8 ring kernels (kh_q8ring.h) all stage this way, so they stay
// bit-identical to each other; against the oracle's w * (rs * x) the result differs by fp32 round-off
// (tests: tokens equal, logits <= 4e-5).
// Every workgroup recomputes the norm redundantly from the L2-resident x (8-16 KiB): cheaper
// than a separate single-block launch + kernel boundary (reference: row_rmsnorm_f32, 1 block).
// LAYOUT_Q8 selects the q8_slot() arrangement.  red = LDS float[KH_WAVES_MAX].

// The barrier of a NORM staging and the scale behind it: per-thread partial `ss` -> wave sum -> red[wave]
// -> __syncthreads (also publishes the staged vector) -> red[0] + red[1] + ... in wave order.
__device__ __forceinline__ float stage_rs(float ss, int M, float eps, float* red) {
  static_assert(KH_WAVES_MAX == 8, "red[] is read as two float4");
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  // all eight words in two reads, words of absent waves (never written) dropped by a select: no branch, no
  // dependent LDS round trips (the clamped-index form of block_sum compiles to a ladder of scalar branches)
  const f32x4 ra = ((const f32x4*)red)[0], rb = ((const f32x4*)red)[1];
  const int n = kh_nwaves();
  float r = 0.f;
  r += ra.x;
  r += n > 1 ? ra.y : 0.f;
  r += n > 2 ? ra.z : 0.f;
  r += n > 3 ? ra.w : 0.f;
  r += n > 4 ? rb.x : 0.f;
  r += n > 5 ? rb.y : 0.f;
  r += n > 6 ? rb.z : 0.f;
  r += n > 7 ? rb.w : 0.f;
  return 1.0f / sqrtf(r / (float)M + eps);
}
template <bool NORM, bool LAYOUT_Q8>
__device__ __forceinline__ float stage_vec(const float* __restrict__ x,
                                           const float* __restrict__ wnorm, f32x4* xs, int M,
                                           float eps, float* red) {
  const int M4 = M >> 2;
  const f32x4* x4 = (const f32x4*)x;
  const f32x4* w4 = (const f32x4*)wnorm;
  const int M16 = M >> 4;
  float ss = 0.f;
  for (int i = threadIdx.x; i < M4; i += kh_wg()) {
    f32x4 v = x4[i];
    if (NORM) {
      ss = fma4(v, v, ss);
      const f32x4 w = w4[i];
      v.x = w.x * v.x;
      v.y = w.y * v.y;
      v.z = w.z * v.z;
      v.w = w.w * v.w;
    }
    xs[LAYOUT_Q8 ? q8_slot(i, M16) : i] = v;
  }
  if constexpr (NORM) return stage_rs(ss, M, eps, red);
  __syncthreads();
  return 1.f;
}

// Two-phase staging used by gemv_pairs: issue() puts MAXV float4 per thread of x (and of the
// norm weight) in flight, finish() writes LDS and returns the RMS scale.  MAXV is a COMPILE-TIME
// choice (straight-line code between the x loads and their use, otherwise the compiler's
// waitcnt merge degrades to vmcnt(0) = "wait for the weights too"), the smallest that covers the
// vector (kh_stage_maxv: a slot beyond the vector is still a load instruction - of element 0 - on the
// CU's address path; Llama-3.2-1B's 2048-float vector needs 2 per thread at 256 threads, 1 at 512):
//   MAXV = 1 / 2 / 4  vectors up to 4 / 8 / 16 floats per thread (4096 at 256 threads: dim of every
//             BASELINE config)
//   MAXV = 6  up to 24 floats per thread (12288 at 512 threads: the 11008-float hidden vector of
//             Llama-2-7B's w2; k_gemv_res only)
//   MAXV = 0  any length: single-phase stage_vec AFTER the first weight loads were issued - its x
//             loads sit behind the weight loads in the in-order vmcnt queue, so the staging (and
//             everything after it) waits for the whole first tile
template <bool NORM, bool LAYOUT_Q8, int MAXV>
struct Stager {
  f32x4 xv[MAXV > 0 ? MAXV : 1];
  f32x4 wv[(NORM && MAXV > 0) ? MAXV : 1];
  const float* x;
  const float* wnorm;
  int M;
  __device__ __forceinline__ Stager(const float* x_, const float* wnorm_, int M_)
      : x(x_), wnorm(wnorm_), M(M_) {}
  __device__ __forceinline__ void issue() {
    if constexpr (MAXV > 0) {
      const int M4 = M >> 2;
      const f32x4* x4 = (const f32x4*)x;
      const f32x4* w4 = (const f32x4*)wnorm;
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int i = threadIdx.x + v * kh_wg();
        const int ci = i < M4 ? i : 0;
        xv[v] = x4[ci];
        if (NORM) wv[v] = w4[ci];
      }
    }
  }
  // returns rs (1 without NORM): the caller's epilogue multiplies its dot products by it
  __device__ __forceinline__ float finish(f32x4* xs, float eps, float* red) {
    if constexpr (MAXV == 0) {
      return stage_vec<NORM, LAYOUT_Q8>(x, wnorm, xs, M, eps, red);
    } else {
      const int M4 = M >> 2, M16 = M >> 4;
      KH_STAMP(5);
      float ss = 0.f;
      if (NORM) {
#pragma unroll
        for (int v = 0; v < MAXV; ++v) {
          const float t = fma4(xv[v], xv[v], 0.f);
          ss += (threadIdx.x + v * kh_wg() < M4) ? t : 0.f;
        }
#ifdef KH_TRACE
        asm volatile("" :: "v"(ss));  // the stamp below is taken once the vector has arrived
#endif
        KH_STAMP(6);
      }
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const int i = threadIdx.x + v * kh_wg();
        if (i < M4) {
          f32x4 t = xv[v];
          if (NORM) {
            t.x = wv[v].x * t.x;
            t.y = wv[v].y * t.y;
            t.z = wv[v].z * t.z;
            t.w = wv[v].w * t.w;
          }
          xs[LAYOUT_Q8 ? q8_slot(i, M16) : i] = t;
        }
      }
      if (NORM) {
        const float rs = stage_rs(ss, M, eps, red);
        KH_STAMP(7);
        return rs;
      }
      __syncthreads();
      return 1.f;
    }
  }
};
// float4 of the vector per staging thread: the smallest of 1 / 2 / 4 / 6 that covers M (0: single-phase staging)
// does the 4-deep in-register staging cover M?  (the op-level kernels, the B-token prefill and the int8 ring
// kernels use that depth whatever the length; a slot beyond the vector adds an exact 0 to the sum of
// squares, so every depth that covers the vector stages the same bits)
static inline bool kh_stage_fits4(int M, int wg = KH_WG) { return M <= 4 * 4 * wg; }
#ifndef KH_STAGE_MAXV_MIN  // experiment hook (profiles/r6_staging_ab.txt): 4 = the depth of rounds 1-5
#define KH_STAGE_MAXV_MIN 1
#endif
static inline int kh_stage_maxv(int M, int wg = KH_WG) {
  if (KH_STAGE_MAXV_MIN >= 4 && M <= 4 * 4 * wg) return 4;
  return M <= 1 * 4 * wg ? 1 : (M <= 2 * 4 * wg ? 2 : (M <= 4 * 4 * wg ? 4 : (M <= 6 * 4 * wg ? 6 : 0)));
}
