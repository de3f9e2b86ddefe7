// kh_common.h — device helpers shared by the op-level and fused kernels (gfx950 / CDNA4).
// Wave = 64 lanes everywhere; workgroups are 256 threads (4 waves, one per SIMD).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/kuiper_hip.h"

#define KH_WAVE 64
#define KH_WG 256
#define KH_WAVES_PER_WG (KH_WG / KH_WAVE)

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

// Weight rows are streamed exactly once per token: non-temporal loads keep them from
// displacing the activations in L2 (MI355X guide: nt-weights row).
template <typename T>
__device__ __forceinline__ T ld_nt(const T* p) {
  return __builtin_nontemporal_load(p);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, KH_WAVE);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, KH_WAVE));
  return v;
}

// Sum over a 256-thread workgroup; every thread gets the result. red = LDS float[4+].
// Two barriers so `red` can be reused immediately afterwards.
__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int w = 0; w < KH_WAVES_PER_WG; ++w) r += red[w];
  __syncthreads();
  return r;
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) red[wave] = v;
  __syncthreads();
  float r = red[0];
#pragma unroll
  for (int w = 1; w < KH_WAVES_PER_WG; ++w) r = fmaxf(r, red[w]);
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

// ---- LDS layout of the activation vector for the int8 GEMV --------------------------------
// A lane owns 16 consecutive weights (one dwordx4), so it needs 16 consecutive x values =
// four float4 (f = 4j+i, i<4) per 16-chunk j.  Stored as slot(f) = (f&3)*(M16+1) + (f>>2):
// for a fixed i consecutive lanes (consecutive j) read consecutive 16-B slots -> no bank
// conflict on ds_read_b128; the +1 pad staggers the four planes for the staging writes.
__device__ __forceinline__ int q8_slot(int f, int M16) { return (f & 3) * (M16 + 1) + (f >> 2); }
static inline size_t kh_q8_lds_bytes(int M) { return (size_t)4 * (size_t)(M / 16 + 1) * 16; }
