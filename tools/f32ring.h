// f32ring.h - the fp32 decode GEMVs (ffn13, w2) on the LDS-DMA ring core: VERDICT r5 item 3, measured in
// tools/mb_f32ring.hip (profiles/r6_fp32_ring_ab.txt).  Not launched by the product.
//
// Same idea as kh_q8ring.h (reference kernel: kuiper/source/op/kernels/cuda/matmul_kernel.cu:7-54): every wave owns R
// ring slots in LDS, a slot holds one PIECE PAIR (1 KiB of each of the pair's two fp32 rows = 256 columns), requested
// by two global_load_lds_dwordx4 operations; the wave keeps R - 1 slots in flight, waits for the oldest with an exact
// s_waitcnt vmcnt(2 (R - 1)), reads it with ds_read_b128, re-requests the slot and runs the eight FMAs.  No flag, no
// barrier, the ring is private to the wave.
// Arithmetic: per lane exactly kh_gemv.h::fma_u<fp32> - lane l owns the float4 l, l + 64, ... of the wave's column
// range in ascending order, one fma4 chain per row from zero - then wave_sum and the fixed-order SPLIT combination: the
// outputs are bit-identical to gemv_pairs (checked word for word by the microbenchmark).
#pragma once
#include <type_traits>

#include "kh_fused_ring.h"

#define KH_F32RING_SLOT 2048  // two 1-KiB row pieces

// StagerAsm (kh_q8ring.h) with the plain fp32 vector layout in LDS: xs[i] = float4 i
template <bool NORM, int MAXV>
struct StagerAsmF32 {
  f32x4 xv[MAXV];
  f32x4 wv[NORM ? MAXV : 1];
  const float* x;
  const float* wnorm;
  f32x4* xs;
  int M;
  __device__ __forceinline__ StagerAsmF32(const float* x_, const float* wnorm_, f32x4* xs_, int M_)
      : x(x_), wnorm(wnorm_), xs(xs_), M(M_) {}
  __device__ __forceinline__ void issue() {
    const int M4 = M >> 2;
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      const int i = threadIdx.x + v * kh_wg();
      const unsigned off = (unsigned)(i < M4 ? i : 0) * 16u;
      asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(xv[v]) : "v"(off), "s"(x) : "memory");
      if (NORM) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(wv[v]) : "v"(off), "s"(wnorm) : "memory");
    }
  }
  template <int YOUNGER>
  __device__ __forceinline__ float finish(float eps, float* red, bool exact) {
    if (exact)
      wait_vm<YOUNGER>();
    else
      wait_vm<0>();
#pragma unroll
    for (int v = 0; v < MAXV; ++v) {
      asm volatile("" : "+v"(xv[v]));
      if (NORM) asm volatile("" : "+v"(wv[v]));
    }
    const int M4 = M >> 2;
    float ss = 0.f;
    if (NORM) {
#pragma unroll
      for (int v = 0; v < MAXV; ++v) {
        const float t = fma4(xv[v], xv[v], 0.f);
        ss += (threadIdx.x + v * kh_wg() < M4) ? t : 0.f;
      }
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
        xs[i] = t;
      }
    }
    if (NORM) return stage_rs(ss, M, eps, red);
    __syncthreads();
    return 1.f;
  }
};

// xs (M floats) | red[KH_WAVES_MAX] | comb[2 * KH_WAVES_MAX] | pad to 256 | rings
__host__ __device__ static inline size_t f32ring_lds_off(int M) {
  return ((size_t)M * 4 + 3 * KH_WAVES_MAX * sizeof(float) + 255) & ~(size_t)255;
}
static inline size_t f32ring_lds_bytes(int M, int waves, int R) {
  return f32ring_lds_off(M) + (size_t)waves * R * KH_F32RING_SLOT;
}

// The row-pair loop: kh_q8ring.h::ring_pairs for fp32 rows (no scales: two DMA operations per piece pair).
template <int SPLIT, int R, class PairFn, class AuxFn, class IssueFn, class FinishFn, class EpiFn>
__device__ __forceinline__ void ring_pairs_f32(int M, const f32x4* xs, int total, int lane, float* comb,
                                               char* ring_base, PairFn&& PAIR, AuxFn&& AUX, IssueFn&& ISSUE,
                                               FinishFn&& FINISH, EpiFn&& EPI) {
  constexpr int OPS = 2;
  static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "SPLIT must be 1, 2 or 4");
  static_assert(R >= 1 && R * OPS <= 60, "ring depth exceeds the vmcnt range");
  const int Mc = M >> 2;
  const int vb = (int)blockIdx.x, vgrid = (int)gridDim.x;
  const int PPW = kh_nwaves() / SPLIT;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int part = wave & (SPLIT - 1);
  const int gp = vb * PPW + wave / SPLIT;
  const int np = vgrid * PPW;
  const int Q = (((Mc + SPLIT - 1) / SPLIT) + 3) & ~3;  // gemv_pairs' column quantum
  const int cb = part * Q;
  const int ce = cb + Q < Mc ? cb + Q : Mc;
  char* const ring = ring_base + (size_t)wave * (R * KH_F32RING_SLOT);
  const unsigned ring0 = (unsigned)__builtin_amdgcn_readfirstlane((int)kh_lds_addr(ring));

  ISSUE();
  __builtin_amdgcn_sched_barrier(0);
  const int ppi = (ce - cb + KH_WAVE - 1) >> 6;  // pieces per work item
  const int my_items = gp < total ? (total - gp + np - 1) / np : 0;
  const int N = my_items * ppi;

  // ---- the issue side: runs R pieces ahead of the consume side
  int pi = gp, ci = cb;
  unsigned si = 0;
  RowsF32 rw = PAIR(gp < total ? gp : 0);
  auto issue1 = [&]() __attribute__((always_inline)) {
    const int idx = ci + lane;
    const int cidx = idx < ce ? idx : 0;  // clamped like load_u; masked at the FMA
    const unsigned d = ring0 + si;
    dma_x4(rw.w0, (unsigned)cidx << 4, d);
    dma_x4(rw.w1, (unsigned)cidx << 4, d + 1024);
    ci += KH_WAVE;
    if (ci >= ce) {
      ci = cb;
      pi += np;
      if (pi < total) rw = PAIR(pi);
    }
    si += KH_F32RING_SLOT;
    if (si == (unsigned)(R * KH_F32RING_SLOT)) si = 0;
  };
  const int n0 = N < R ? N : R;
  for (int k = 0; k < n0; ++k) issue1();
  FINISH(n0 == R);

  // ---- the consume side
  int pc = gp, cc = cb;
  unsigned sc = 0;
  float a0 = 0.f, a1 = 0.f;
  auto aux = AUX(gp < total ? gp : 0);
  auto finish_item = [&]() __attribute__((always_inline)) {
    float s0 = wave_sum(a0), s1 = wave_sum(a1);
    if constexpr (SPLIT == 1) {
      EPI(pc, s0, s1, aux);
    } else {
      if (lane == 0) {
        comb[2 * wave] = s0;
        comb[2 * wave + 1] = s1;
      }
      __syncthreads();
      if (part == 0) {
        s0 = comb[2 * wave];
        s1 = comb[2 * wave + 1];
#pragma unroll
        for (int k = 1; k < SPLIT; ++k) {
          s0 += comb[2 * (wave + k)];
          s1 += comb[2 * (wave + k) + 1];
        }
        EPI(pc, s0, s1, aux);
      }
      __syncthreads();
    }
  };
  auto consume = [&](auto wtag, auto refill) __attribute__((always_inline)) {
    wait_vm<decltype(wtag)::value>();
    const char* s = ring + sc;
    const f32x4 v0 = ((const f32x4*)s)[lane];
    const f32x4 v1 = ((const f32x4*)(s + 1024))[lane];
    const int idx = cc + lane;
    const bool in = idx < ce;
    f32x4 xv = xs[in ? idx : 0];
    // the slot's bytes are in registers before the slot is handed back to the DMA engine
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (decltype(refill)::value) issue1();
    if (cc == cb) aux = AUX(pc);  // uniform; scalar loads, consumed at the end of the item
    xv.x = in ? xv.x : 0.f;
    xv.y = in ? xv.y : 0.f;
    xv.z = in ? xv.z : 0.f;
    xv.w = in ? xv.w : 0.f;
    a0 = fma4(v0, xv, a0);
    a1 = fma4(v1, xv, a1);
    cc += KH_WAVE;
    sc += KH_F32RING_SLOT;
    if (sc == (unsigned)(R * KH_F32RING_SLOT)) sc = 0;
    if (cc >= ce) {
      finish_item();
      pc += np;
      cc = cb;
      a0 = a1 = 0.f;
    }
  };
  int k = 0;
  for (; k + R < N; ++k) consume(std::integral_constant<int, OPS*(R - 1)>{}, std::true_type{});
  ring_tail<R - 1>(k, N, [&](auto jt) __attribute__((always_inline)) {
    consume(std::integral_constant<int, OPS * decltype(jt)::value>{}, std::false_type{});
  });
  if constexpr (SPLIT > 1) {
    const int iters = total > 0 ? (total + np - 1) / np : 0;
    for (int e = my_items; e < iters; ++e) {
      __syncthreads();
      __syncthreads();
    }
  }
}

// k_ffn13<false, ...> on the ring
template <int R, int MAXV>
__global__ __launch_bounds__(1024) void k_ffn13_ring_f32(const KhFfn13Args a) {
  extern __shared__ __attribute__((aligned(256))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<false>(xs, a.dim);
  const int lane = threadIdx.x & 63;
  const int dim = a.dim;
  const void *w1 = a.w1.w, *w3 = a.w3.w;
  float* const h = a.h;
  const float eps = a.eps;
  const Gemv<false, 1> g(dim, 0);
  StagerAsmF32<true, MAXV> st(a.x, a.ffn_norm, xs, dim);
  auto pair = [&](int r) __attribute__((always_inline)) { return g.rows(w1, r, w3, r, nullptr, nullptr, dim); };
  float rs = 1.f;
  auto epi = [&](int r, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (lane == 0) h[r] = swiglu1(rs * s0, rs * s1);
  };
  ring_pairs_f32<1, R>(
      dim, xs, a.hidden, lane, nullptr, smem_raw + f32ring_lds_off(dim), pair,
      [](int) __attribute__((always_inline)) { return NoAux{}; },
      [&]() __attribute__((always_inline)) { st.issue(); },
      [&](bool exact) __attribute__((always_inline)) { rs = st.template finish<R * 2>(eps, red, exact); }, epi);
}

// k_gemv_res<false, ...> (wo / w2) on the ring; the residual words come through the scalar cache
template <int R, int MAXV, int SPLIT>
__global__ __launch_bounds__(1024) void k_gemv_res_ring_f32(const KhGemvResArgs a) {
  extern __shared__ __attribute__((aligned(256))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  float* red = lds_red_ptr<false>(xs, a.M);
  const int lane = threadIdx.x & 63;
  const int M = a.M;
  const void* const w = a.w.w;
  float* const x = a.x;
  const Gemv<false, 1> g(M, 0);
  StagerAsmF32<false, MAXV> st(a.vec, nullptr, xs, M);
  auto pair = [&](int p) __attribute__((always_inline)) { return g.rows(w, 2 * p, w, 2 * p + 1, nullptr, nullptr, M); };
  struct Aux {
    float x0, x1;
  };
  auto auxf = [&](int p) __attribute__((always_inline)) { return Aux{ld_uniform(x + 2 * p), ld_uniform(x + 2 * p + 1)}; };
  auto epi = [&](int p, float s0, float s1, const Aux& r) __attribute__((always_inline)) {
    if (lane != 0) return;
    x[2 * p] = r.x0 + s0;
    x[2 * p + 1] = r.x1 + s1;
  };
  ring_pairs_f32<SPLIT, R>(
      M, xs, a.K >> 1, lane, red + KH_WAVES_MAX, smem_raw + f32ring_lds_off(M), pair, auxf,
      [&]() __attribute__((always_inline)) { st.issue(); },
      [&](bool exact) __attribute__((always_inline)) { (void)st.template finish<R * 2>(0.f, red, exact); }, epi);
}
