// kh_q8ring.h — int8 group-dequant GEMV whose weight stream runs through a per-wave LDS ring filled by the
// DMA path of the vector memory unit (global_load_lds_dwordx4, gfx950).
//
// Replaces, for the int8 decode GEMVs, the load-to-VGPR core of kh_gemv.h (reference kernel:
// kuiper/source/op/kernels/cuda/matmul_kernel.cu:56-87).  Why: with register tiles a wave's bytes in flight
// are bounded by its VGPRs (8 KiB per wave, 64 KiB per CU on the shipped shapes) and it has nothing in flight
// while it dequantises; the pure-stream floor of the same bytes is 15-25 % below the shipped kernels
// (profiles/r3_int8_floors_stream.txt).  Here every wave owns R ring slots in LDS; a slot holds one PIECE PAIR
// (1 KiB of each of the pair's two rows + the group scale each lane needs).
// The wave keeps R-1 slots in flight at all times - requested by LDS-DMA, which needs no registers - waits for
// the oldest with an exact s_waitcnt vmcnt(OPS * (R-1)), reads it with ds_read_b128, re-requests the slot for
// the piece R positions ahead and only then runs the converts and FMAs.  No wave waits on another wave: the only
// landing signal LDS-DMA has is the issuing wave's own vmcnt, so the ring is private to the wave and there is no
// flag, no barrier and no loader/consumer hand-off.
//
// Arithmetic: per lane exactly the order of kh_gemv.h::fma_u - lane l owns the 16-byte chunks l, l + 64, ... of
// the wave's column range in ascending order, 16 sequential FMAs from zero per chunk, one scale FMA per chunk,
// then the DPP butterfly of wave_sum and the fixed-order combination of SPLIT parts - so results are
// bit-identical to gemv_pairs (and to the B-token prefill, kh_prefill.h) for the same SPLIT.
//
// The compiler does not know that the asm DMA operations occupy vmcnt slots; its own s_waitcnt for a
// compiler-visible global LOAD would therefore drain the ring.  Hence: no vector load the compiler knows about
// while the ring is live.  The activation vector is fetched by asm register loads with a hand-placed wait
// (StagerAsm; tools/ring_variants.h::StagerDma is the measured-equal alternative through the DMA path), epilogue operands (residual,
// sin / cos, bias) through the scalar cache (ld_uniform: lgkmcnt), pointers and sizes are kernel arguments.
// Stores are fine (nothing waits for them; vmcnt(N) with extra younger stores only over-waits).
#pragma once
#include <type_traits>

#include "kh_gemv.h"

#define KH_RING_W 2048                 // two 1-KiB row pieces
#define KH_RING_SC (KH_RING_W)         // two 256-B scale vectors (one float per lane)
#define KH_RING_SLOT (KH_RING_W + 512)

__device__ __forceinline__ unsigned kh_lds_addr(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}
// M0 (the LDS destination of an LDS-DMA operation) is written inside the statements below and named in their
// clobber lists, so the compiler sees a definition of M0 there and cannot carry a value of its own in M0 across
// them.  M0 is a reserved register of the AMDGPU backend (never allocated, written by the compiler only right in
// front of the instruction that consumes it), which is what -Winline-asm remarks on; nothing is expected to be
// "preserved".
#pragma clang diagnostic push
#pragma clang diagnostic ignored "-Winline-asm"
// 64 lanes x 16 B from (base + voff[lane]) to LDS [dst + 16 * lane]; dst is wave-uniform.  Weights: read once, nt.
__device__ __forceinline__ void dma_x4(const void* base, unsigned voff, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt"
               :
               : "v"(voff), "s"(base), "s"(dst)
               : "memory", "m0");
}
// 64 lanes x 4 B to LDS [dst + 4 * lane]
__device__ __forceinline__ void dma_x1(const void* base, unsigned voff, unsigned dst) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1"
               :
               : "v"(voff), "s"(base), "s"(dst)
               : "memory", "m0");
}
#pragma clang diagnostic pop
template <int N>
__device__ __forceinline__ void wait_vm() {
  static_assert(N >= 0 && N <= 63, "vmcnt is a 6-bit counter");
  asm volatile("s_waitcnt vmcnt(%0)" : : "n"(N) : "memory");
}
// A wave-uniform word through the scalar cache (s_load: lgkmcnt, never vmcnt).  For epilogue operands only:
// read-only tables (sin / cos, bias) and residual words that no OTHER wave writes during the kernel - the one
// wave that owns row pair p reads x[2p], x[2p+1] before it stores them, so a scalar-cache line that went stale
// in its neighbours' words still holds the right value in ours; the cache is invalidated at kernel start.
__device__ __forceinline__ float ld_uniform(const float* p) {
  return *(const __attribute__((address_space(4))) float*)(unsigned long long)p;
}

// Input vector staging with register loads issued by inline asm (the compiler must not know them: its own wait
// for a load it knows about counts only the loads it knows about and would drain the ring behind it).  issue()
// requests MAXV float4 of x (and of the norm weight) per thread, exactly like Stager; finish() waits for exactly
// those - the YOUNGER operations of the ring prologue stay in flight (loads and LDS-DMA operations retire in issue
// order through the one vmcnt counter: tools/mb_vmcnt_order.hip) - ties every loaded register to the wait so that no
// use can be scheduled ahead of it, and then does what Stager<NORM, true, MAXV>::finish does, in the same order.
// Against tools/ring_variants.h::StagerDma (the same through LDS-DMA pieces, measured equal): no LDS detour, two
// barriers fewer, no raw norm-weight area in LDS.
template <bool NORM, int MAXV, int VT = 0>
struct StagerAsm {
  static_assert(VT == 0, "the register staging uses the real workgroup width");
  static constexpr bool kRawArea = false;  // no raw norm-weight area between the reduction words and the rings
  f32x4 xv[MAXV];
  f32x4 wv[NORM ? MAXV : 1];
  const float* x;
  const float* wnorm;
  f32x4* xs;
  int M;
  __device__ __forceinline__ StagerAsm(const float* x_, const float* wnorm_, f32x4* xs_, const void* /*wraw*/, int M_)
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
  // returns the RMS scale (1 without NORM), as Stager::finish
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
    const int M4 = M >> 2, M16 = M >> 4;
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
        xs[q8_slot(i, M16)] = t;
      }
    }
    if (NORM) return stage_rs(ss, M, eps, red);
    __syncthreads();
    return 1.f;
  }
};

template <int J, class F>
__device__ __forceinline__ void ring_tail(int& k, int N, F&& f) {
  if (N - k == J + 1) {  // uniform
    f(std::integral_constant<int, J>{});
    ++k;
  }
  if constexpr (J > 0) ring_tail<J - 1>(k, N, f);
}

// xs (q8 layout) | red[KH_WAVES_MAX] | comb[2 * KH_WAVES_MAX] | pad to 256 | (StagerDma + norm only: raw norm weight,
// M floats) | rings
__host__ __device__ static inline size_t ring_lds_wraw_off(int M) {
  const size_t xs_bytes = (size_t)4 * (size_t)(M / 16 + 1) * 16;  // kh_q8_lds_bytes
  return ((xs_bytes + 3 * KH_WAVES_MAX * sizeof(float)) + 255) & ~(size_t)255;
}
__host__ __device__ static inline size_t ring_lds_off(int M, bool norm) {
  return ring_lds_wraw_off(M) + (norm ? (size_t)M * 4 : 0);
}
static inline size_t ring_lds_bytes(int M, bool norm, int waves, int R) {
  return ring_lds_off(M, norm) + (size_t)waves * R * KH_RING_SLOT;
}

// The row-pair loop.  Same work decomposition as gemv_pairs: work item = (pair p, column part), p = gp + it * np.
//   PAIR(p)        -> RowsQ8 of work item p (scalar address arithmetic)
//   AUX(p)         -> small struct of epilogue operands fetched through the scalar cache (ld_uniform) when the
//                     item's first piece is consumed - no vector load may be issued while the ring is live
//   ISSUE()        -> StagerAsm::issue
//   FINISH(exact)  -> StagerAsm::finish<R * 4>(..., exact)
//   EPI(p, s0, s1, aux) -> epilogue with the two dot products
// BLOCKED: workgroup b owns the contiguous items [b * ipw, (b + 1) * ipw), ipw = ceil(total / grid), and its waves
// take them round-robin - with one workgroup per CU every CU streams the same number of bytes whatever the wave
// count (gemv_pairs' mapping p = gp + it * np gives the first workgroups one item more per wave than the last).
template <int SPLIT, int R, bool BLOCKED, class PairFn, class AuxFn, class IssueFn, class FinishFn, class EpiFn>
__device__ __forceinline__ void ring_pairs(int M, int gshift, const f32x4* xs, int total, int lane, float* comb,
                                           char* ring_base, PairFn&& PAIR, AuxFn&& AUX, IssueFn&& ISSUE,
                                           FinishFn&& FINISH, EpiFn&& EPI) {
  constexpr int OPS = 4;
  static_assert(SPLIT == 1 || SPLIT == 2 || SPLIT == 4, "SPLIT must be 1, 2 or 4");
  static_assert(R >= 1 && R * OPS <= 60, "ring depth exceeds the vmcnt range");
  const int Mc = M >> 4, plane = Mc + 1;
  const int vb = (int)blockIdx.x, vgrid = (int)gridDim.x;
  const int PPW = kh_nwaves() / SPLIT;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int part = wave & (SPLIT - 1);
  const int ipw = BLOCKED ? (total + vgrid - 1) / vgrid : 0;
  const int gp = BLOCKED ? vb * ipw + wave / SPLIT : vb * PPW + wave / SPLIT;
  const int np = BLOCKED ? PPW : vgrid * PPW;
  if constexpr (BLOCKED) total = (vb + 1) * ipw < total ? (vb + 1) * ipw : total;  // this workgroup's end
  const int Q = (((Mc + SPLIT - 1) / SPLIT) + 3) & ~3;  // gemv_pairs' column quantum
  const int cb = part * Q;
  const int ce = cb + Q < Mc ? cb + Q : Mc;
  char* const ring = ring_base + (size_t)wave * (R * KH_RING_SLOT);
  const unsigned ring0 = (unsigned)__builtin_amdgcn_readfirstlane((int)kh_lds_addr(ring));

  ISSUE();  // the vector's pieces leave before any of the work-item arithmetic below (an integer division among it)
  __builtin_amdgcn_sched_barrier(0);
  const int ppi = (ce - cb + KH_WAVE - 1) >> 6;  // pieces per work item
  const int my_items = gp < total ? (total - gp + np - 1) / np : 0;
  const int N = my_items * ppi;

  // ---- the issue side: runs R pieces ahead of the consume side
  int pi = gp, ci = cb;
  unsigned si = 0;
  RowsQ8 rw = PAIR(gp < total ? gp : 0);
  auto issue1 = [&]() __attribute__((always_inline)) {
    const int idx = ci + lane;
    const int cidx = idx < ce ? idx : 0;  // clamped like load_u; masked at the FMA
    const unsigned d = ring0 + si;
    dma_x4(rw.w0, (unsigned)cidx << 4, d);
    dma_x4(rw.w1, (unsigned)cidx << 4, d + 1024);
    const unsigned go = (unsigned)((cidx << 4) >> gshift) << 2;
    dma_x1(rw.sc0, go, d + KH_RING_SC);
    dma_x1(rw.sc1, go, d + KH_RING_SC + 256);
    ci += KH_WAVE;
    if (ci >= ce) {
      ci = cb;
      pi += np;
      if (pi < total) rw = PAIR(pi);
    }
    si += KH_RING_SLOT;
    if (si == (unsigned)(R * KH_RING_SLOT)) si = 0;
  };
  const int n0 = N < R ? N : R;
  for (int k = 0; k < n0; ++k) issue1();
  FINISH(n0 == R);
  KH_STAMP(1);

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
    const i32x4 q0 = ((const i32x4*)s)[lane];
    const i32x4 q1 = ((const i32x4*)(s + 1024))[lane];
    const float g0 = ((const float*)(s + KH_RING_SC))[lane];
    const float g1 = ((const float*)(s + KH_RING_SC + 256))[lane];
    const int idx = cc + lane;
    const bool in = idx < ce;
    const int cx = in ? idx : 0;
    const f32x4 x0 = xs[cx], x1 = xs[plane + cx], x2 = xs[2 * plane + cx], x3 = xs[3 * plane + cx];
    // the slot's bytes are in registers before the slot is handed back to the DMA engine
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (decltype(refill)::value) issue1();
    if (cc == cb) aux = AUX(pc);  // uniform; scalar loads, consumed at the end of the item
    float t0 = 0.f, t1 = 0.f;
    t0 = dot4_i8(q0.x, x0, t0);
    t0 = dot4_i8(q0.y, x1, t0);
    t0 = dot4_i8(q0.z, x2, t0);
    t0 = dot4_i8(q0.w, x3, t0);
    t1 = dot4_i8(q1.x, x0, t1);
    t1 = dot4_i8(q1.y, x1, t1);
    t1 = dot4_i8(q1.z, x2, t1);
    t1 = dot4_i8(q1.w, x3, t1);
    a0 = __builtin_fmaf(in ? g0 : 0.f, t0, a0);
    a1 = __builtin_fmaf(in ? g1 : 0.f, t1, a1);
    cc += KH_WAVE;
    sc += KH_RING_SLOT;
    if (sc == (unsigned)(R * KH_RING_SLOT)) sc = 0;
    if (cc >= ce) {
      finish_item();
      if (pc == gp) KH_STAMP(2);
      pc += np;
      cc = cb;
      a0 = a1 = 0.f;
    }
  };
  int k = 0;
  for (; k + R < N; ++k) consume(std::integral_constant<int, OPS*(R - 1)>{}, std::true_type{});
  // the tail: everything is requested, j + 1 pieces are left -> j pieces are younger than the one waited for
  ring_tail<R - 1>(k, N, [&](auto jt) __attribute__((always_inline)) {
    consume(std::integral_constant<int, OPS * decltype(jt)::value>{}, std::false_type{});
  });
  KH_STAMP_W();
  KH_STAMP(3);
  if constexpr (SPLIT > 1) {
    const int span = BLOCKED ? total - vb * ipw : total;  // items the workgroup's waves share
    const int iters = span > 0 ? (span + np - 1) / np : 0;
    for (int e = my_items; e < iters; ++e) {
      __syncthreads();
      __syncthreads();
    }
  }
}
