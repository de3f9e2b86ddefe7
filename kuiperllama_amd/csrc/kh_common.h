// kh_common.h — device helpers shared by the op-level and fused kernels (gfx950 / CDNA4).
// Wave = 64 lanes everywhere; workgroups are 256 threads (4 waves, one per SIMD).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kuiper_hip.h"

#define KH_WAVE 64
#ifndef KH_WG
#define KH_WG 256
#endif
#define KH_WAVES_PER_WG (KH_WG / KH_WAVE)
// The fused decode kernels read their workgroup size from blockDim.x (256 or 512 threads, chosen
// per launch: a wider workgroup halves the per-launch re-staging of a long input vector, see
// pick_shape); KH_WG is the default and the size every op-level kernel uses.
#define KH_WG_MAX 512
#define KH_WAVES_MAX (KH_WG_MAX / KH_WAVE)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define KH_CHECK_HIP(expr)                      \
  do {                                          \
    hipError_t _e = (expr);                     \
    if (_e != hipSuccess) return (int)_e;       \
  } while (0)

static inline int kh_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? KH_OK : (int)e;
}

static inline bool kh_aligned16(const void* p) { return (((uintptr_t)p) & 15u) == 0; }

// Tuning / test hooks (KH_SHAPE_*, KH_PREFILL, KH_PG_*, ...): looked up in the process-wide table of
// kh_debug.cpp (seeded once from the KH_* environment variables, changed through kh_debug_set()).
// nullptr when the hook is not set.  Host code only.
namespace khm {
const char* dbg(const char* key);
inline bool dbg_off(const char* key) {  // hook present and starting with '0'
  const char* e = dbg(key);
  return e && e[0] == '0';
}
}  // namespace khm

// Weight rows are streamed exactly once per token: non-temporal loads keep them from
// displacing the activations in L2 (MI355X guide: nt-weights row).
template <typename T>
__device__ __forceinline__ T ld_nt(const T* p) {
  return __builtin_nontemporal_load(p);
}

// ---- wave64 cross-lane reductions without LDS ----------------------------------------------
// hipcc lowers __shfl_xor to ds_bpermute_b32 (an LDS-pipe instruction, ~50+ cycles each and six
// of them in a dependent chain per reduction).  DPP quad/row permutes + the gfx950
// v_permlane16/32_swap do the same butterfly on the VALU: xor1, xor2 (quad_perm), half-mirror
// (= xor4 once quads are uniform), mirror (= xor8), permlane16_swap (rows 0<->1, 2<->3),
// permlane32_swap (halves).  Pairing is identical to an xor butterfly, so sums are bitwise
// the same as the shuffle version.
#define KH_DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define KH_DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define KH_DPP_HALF_MIRROR 0x141  // lane i <-> 7-i within 8
#define KH_DPP_MIRROR 0x140       // lane i <-> 15-i within 16

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
  return __builtin_bit_cast(
      float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
// v_permlane16_swap(A, B): odd rows of A <-> even rows of B (row = 16 lanes).  With A = B = v
// the two results are {own row's value, partner row's value} in some order in EVERY lane, so a
// commutative combine of r[0], r[1] is the xor-16 butterfly step.  Same for 32-lane halves.
#define KH_SWAP16(v, a, b)                                                     \
  do {                                                                         \
    const int _i = __builtin_bit_cast(int, (v));                               \
    const auto _r = __builtin_amdgcn_permlane16_swap(_i, _i, false, false);    \
    (a) = __builtin_bit_cast(float, (int)_r[0]);                               \
    (b) = __builtin_bit_cast(float, (int)_r[1]);                               \
  } while (0)
#define KH_SWAP32(v, a, b)                                                     \
  do {                                                                         \
    const int _i = __builtin_bit_cast(int, (v));                               \
    const auto _r = __builtin_amdgcn_permlane32_swap(_i, _i, false, false);    \
    (a) = __builtin_bit_cast(float, (int)_r[0]);                               \
    (b) = __builtin_bit_cast(float, (int)_r[1]);                               \
  } while (0)

__device__ __forceinline__ float xor16_sum(float v) {
  float a, b;
  KH_SWAP16(v, a, b);
  return a + b;
}
__device__ __forceinline__ float xor32_sum(float v) {
  float a, b;
  KH_SWAP32(v, a, b);
  return a + b;
}
__device__ __forceinline__ float xor16_max(float v) {
  float a, b;
  KH_SWAP16(v, a, b);
  return fmaxf(a, b);
}
__device__ __forceinline__ float xor32_max(float v) {
  float a, b;
  KH_SWAP32(v, a, b);
  return fmaxf(a, b);
}

// Phase stamps of the decode-step GEMV kernels (tools/mb_q8ring.hip builds with -DKH_TRACE and reads them back:
// where a launch's microseconds go, per workgroup and per wave).  Compiled out of the product.  The stamps go to LDS
// and are flushed at the kernel's end: a global store per stamp would join the in-order vmcnt queue the register
// tiles and the LDS-DMA ring count their loads on.
//   0 entry | 1 input vector staged | 2 first work item finished | 3 last work item finished | 4 kernel end
//   5 vector and first tile requested | 6 vector arrived (sum of squares formed) | 7 block sum done
//   8 + w: wave w's last work item finished
#ifdef KH_TRACE
__device__ unsigned long long* kh_trace_buf;  // [gridDim.x][32] of the 100-MHz constant clock
__device__ __forceinline__ unsigned long long* kh_trace_lds() {
  __shared__ unsigned long long t[32];
  return t;
}
#define KH_STAMP_INIT()                                      \
  do {                                                       \
    if (threadIdx.x < 32) kh_trace_lds()[threadIdx.x] = 0;   \
    if (threadIdx.x == 0) kh_trace_lds()[0] = wall_clock64(); \
  } while (0)
#define KH_STAMP(i)                                              \
  do {                                                           \
    __builtin_amdgcn_sched_barrier(0);                           \
    if (threadIdx.x == 0) kh_trace_lds()[(i)] = wall_clock64();  \
    __builtin_amdgcn_sched_barrier(0);                           \
  } while (0)
#define KH_STAMP_W()                                                                                  \
  do {                                                                                                \
    if ((threadIdx.x & 63) == 0 && threadIdx.x < 512) kh_trace_lds()[8 + (threadIdx.x >> 6)] = wall_clock64(); \
  } while (0)
#define KH_STAMP_FLUSH()                                                                                   \
  do {                                                                                                     \
    KH_STAMP(4);                                                                                           \
    __syncthreads();                                                                                       \
    if (threadIdx.x < 32) kh_trace_buf[(size_t)blockIdx.x * 32 + threadIdx.x] = kh_trace_lds()[threadIdx.x]; \
  } while (0)
#else
#define KH_STAMP_INIT() do { } while (0)
#define KH_STAMP(i) do { } while (0)
#define KH_STAMP_W() do { } while (0)
#define KH_STAMP_FLUSH() do { } while (0)
#endif

// sum over aligned groups of G lanes (G = 1,2,4,...,64); every lane gets its group's sum
template <int G>
__device__ __forceinline__ float group_sum(float v) {
  if (G >= 2) v += dpp_f32<KH_DPP_XOR1>(v);
  if (G >= 4) v += dpp_f32<KH_DPP_XOR2>(v);
  if (G >= 8) v += dpp_f32<KH_DPP_HALF_MIRROR>(v);
  if (G >= 16) v += dpp_f32<KH_DPP_MIRROR>(v);
  if (G >= 32) v = xor16_sum(v);
  if (G >= 64) v = xor32_sum(v);
  return v;
}
template <int G>
__device__ __forceinline__ float group_max(float v) {
  if (G >= 2) v = fmaxf(v, dpp_f32<KH_DPP_XOR1>(v));
  if (G >= 4) v = fmaxf(v, dpp_f32<KH_DPP_XOR2>(v));
  if (G >= 8) v = fmaxf(v, dpp_f32<KH_DPP_HALF_MIRROR>(v));
  if (G >= 16) v = fmaxf(v, dpp_f32<KH_DPP_MIRROR>(v));
  if (G >= 32) v = xor16_max(v);
  if (G >= 64) v = xor32_max(v);
  return v;
}
// sum / max ACROSS the 64/G groups of a wave (lanes with equal lane % G), G >= 16
template <int G>
__device__ __forceinline__ float across_groups_sum(float v) {
  static_assert(G >= 16, "across_groups_* needs G >= 16");
  if (G <= 16) v = xor16_sum(v);
  if (G <= 32) v = xor32_sum(v);
  return v;
}
template <int G>
__device__ __forceinline__ float across_groups_max(float v) {
  static_assert(G >= 16, "across_groups_* needs G >= 16");
  if (G <= 16) v = xor16_max(v);
  if (G <= 32) v = xor32_max(v);
  return v;
}

__device__ __forceinline__ float wave_sum(float v) { return group_sum<64>(v); }
__device__ __forceinline__ float wave_max(float v) { return group_max<64>(v); }

// Workgroup width.  NOT blockDim.x: HIP's blockDim goes through __ockl_get_local_size, which selects
// between the full and the remainder group size; when the optimiser sinks that select into the
// ADDRESS (one load from "offset 12 or 18") the uniform-workgroup fold no longer matches and the
// kernel starts with a vector global_load_ushort + s_waitcnt vmcnt(0) - a memory round trip
// before the first useful load (seen in every decode kernel: +0.1-0.3 us each, 1-2 % of a token).
// Every launch of this library uses full workgroups, so the hidden group-size word (a scalar
// kernarg load) is the answer.
__device__ __forceinline__ int kh_wg() { return (int)__builtin_amdgcn_workgroup_size_x(); }
__device__ __forceinline__ int kh_nwaves() { return (int)(__builtin_amdgcn_workgroup_size_x() >> 6); }

// Sum over the workgroup; every thread gets the result. red = LDS float[KH_WAVES_MAX+].
// Two barriers so `red` can be reused immediately afterwards.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = 0.f;
  const int n = kh_nwaves();
#pragma unroll  // clamped index + select: all KH_WAVES_MAX LDS reads issue back to back
  for (int w = 0; w < KH_WAVES_MAX; ++w) r += w < n ? red[w < n ? w : 0] : 0.f;
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
  const int n = kh_nwaves();
#pragma unroll
  for (int w = 1; w < KH_WAVES_MAX; ++w) r = fmaxf(r, red[w < n ? w : 0]);
  __syncthreads();
  return r;
}

__device__ __forceinline__ float fma4(f32x4 w, f32x4 x, float acc) {
  acc = __builtin_fmaf(w.x, x.x, acc);
  acc = __builtin_fmaf(w.y, x.y, acc);
  acc = __builtin_fmaf(w.z, x.z, acc);
  acc = __builtin_fmaf(w.w, x.w, acc);
  return acc;
}

// 4 packed int8 (one dword) . 4 floats
__device__ __forceinline__ float dot4_i8(int packed, f32x4 x, float acc) {
  acc = __builtin_fmaf((float)(int8_t)(packed & 0xff), x.x, acc);
  acc = __builtin_fmaf((float)(int8_t)((packed >> 8) & 0xff), x.y, acc);
  acc = __builtin_fmaf((float)(int8_t)((packed >> 16) & 0xff), x.z, acc);
  acc = __builtin_fmaf((float)(packed >> 24), x.w, acc);  // arithmetic shift sign-extends
  return acc;
}

// silu(a) * b exactly as cpu/swiglu_kernel.cpp:21-22: a * (1/(1+exp(-a))) * b
__device__ __forceinline__ float swiglu1(float a, float b) {
  const float sg = 1.0f / (1.0f + expf(-a));
  return (a * sg) * b;
}

// argmax candidate merge: larger value wins, ties -> lower index (argmax_sampler.cpp:7,
// cuda/argmax_kernel.cu:13-18)
__device__ __forceinline__ void amax_merge(float& v, int& i, float ov, int oi) {
  if (ov > v || (ov == v && oi < i)) {
    v = ov;
    i = oi;
  }
}
__device__ __forceinline__ void wave_amax(float& v, int& i) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const float ov = __shfl_xor(v, off, KH_WAVE);
    const int oi = __shfl_xor(i, off, KH_WAVE);
    amax_merge(v, i, ov, oi);
  }
}

// weights of one projection, fp32 or int8+scales (file order of the .bin: int8[K*M] then scales)
struct KhLin {
  const void* w;        // fp32 [K,M] or int8 [K,M]
  const float* scales;  // int8 only: [K*M/group]
  const float* bias;    // Qwen2 q/k/v only
};

// ---- LDS layout of the activation vector for the int8 GEMV --------------------------------
// A lane owns 16 consecutive weights (one dwordx4), so it needs 16 consecutive x values =
// four float4 (f = 4j+i, i<4) per 16-chunk j.  Stored as slot(f) = (f&3)*(M16+1) + (f>>2):
// for a fixed i consecutive lanes (consecutive j) read consecutive 16-B slots -> no bank
// conflict on ds_read_b128; the +1 pad staggers the four planes for the staging writes.
__device__ __forceinline__ int q8_slot(int f, int M16) { return (f & 3) * (M16 + 1) + (f >> 2); }
static inline size_t kh_q8_lds_bytes(int M) { return (size_t)4 * (size_t)(M / 16 + 1) * 16; }
