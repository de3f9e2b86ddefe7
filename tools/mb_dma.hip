// mb_dma.hip — how fast can ONE wave per CU pull a weight stream into an LDS ring with global_load_lds_dwordx4?
// (the loader of tools/mb_engine.hip in isolation).  256 workgroups, each streams its own contiguous slab.
//   V=0  one asm statement per 1 KiB piece (M0 saved / restored each time), vaddr = 64-bit pointer
//   V=1  one asm statement per ROW of P pieces: saddr + 32-bit lane offset, M0 bumped in the loop
//   V=2  as V=1 without nt
//   V=3  plain global_load_dwordx4 into registers (16 per batch), no LDS (reference: the product's way)
// DEPTH = batches of 16 pieces allowed in flight (vmcnt counts up to 63 only).
//   hipcc --offload-arch=gfx950 -O3 tools/mb_dma.hip -o kuiperllama_amd/lib/mb_dma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
typedef __attribute__((address_space(3))) char lds_char;
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define RING 96  // pieces

__device__ __forceinline__ void dma_piece(const void* gsrc, unsigned lds_dst) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
// n pieces of one row: base (uniform pointer), voff = lane * 16, ring position dst (bytes), wraps at rend -> rbeg
template <bool NT>
__device__ __forceinline__ void dma_row(const void* base, unsigned voff, unsigned& dst, unsigned rbeg, unsigned rend, int n) {
  unsigned keep;
  if (NT)
    asm volatile(
        "s_mov_b32 %[keep], m0\n"
        ".Lrow%=:\n\t"
        "s_mov_b32 m0, %[dst]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[voff], %[base] nt\n\t"
        "v_add_u32 %[voff], 0x400, %[voff]\n\t"
        "s_add_u32 %[dst], %[dst], 0x400\n\t"
        "s_cmp_eq_u32 %[dst], %[rend]\n\t"
        "s_cselect_b32 %[dst], %[rbeg], %[dst]\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 .Lrow%=\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [voff] "+v"(voff), [dst] "+s"(dst), [n] "+s"(n)
        : [base] "s"(base), [rend] "s"(rend), [rbeg] "s"(rbeg)
        : "memory", "scc");
  else
    asm volatile(
        "s_mov_b32 %[keep], m0\n"
        ".Lrow%=:\n\t"
        "s_mov_b32 m0, %[dst]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[voff], %[base]\n\t"
        "v_add_u32 %[voff], 0x400, %[voff]\n\t"
        "s_add_u32 %[dst], %[dst], 0x400\n\t"
        "s_cmp_eq_u32 %[dst], %[rend]\n\t"
        "s_cselect_b32 %[dst], %[rbeg], %[dst]\n\t"
        "s_sub_u32 %[n], %[n], 1\n\t"
        "s_cmp_lg_u32 %[n], 0\n\t"
        "s_cbranch_scc1 .Lrow%=\n\t"
        "s_mov_b32 m0, %[keep]"
        : [keep] "=&s"(keep), [voff] "+v"(voff), [dst] "+s"(dst), [n] "+s"(n)
        : [base] "s"(base), [rend] "s"(rend), [rbeg] "s"(rbeg)
        : "memory", "scc");
}

template <int V, int DEPTH, int WAVES>
__global__ __launch_bounds__(256) void k_stream(const char* __restrict__ slab, size_t bytes_per_cu, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem_g[];
  lds_char* smem = (lds_char*)smem_g;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int lane = threadIdx.x & 63;
  if (wave >= WAVES) return;
  // WAVES loader waves split the CU's slab
  const size_t per_wave = bytes_per_cu / WAVES;
  const char* src = slab + (size_t)blockIdx.x * bytes_per_cu + (size_t)wave * per_wave;
  const unsigned rbeg = (unsigned)(size_t)smem + (unsigned)wave * (RING / WAVES) * 1024u, rend = rbeg + (RING / WAVES) * 1024u;
  const int npieces = (int)(per_wave >> 10);
  if (V == 3) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < npieces; p += 16) {
      f32x4 v[16];
#pragma unroll
      for (int k = 0; k < 16; ++k) v[k] = __builtin_nontemporal_load((const f32x4*)(src + (size_t)(p + k) * 1024) + lane);
#pragma unroll
      for (int k = 0; k < 16; ++k) acc += v[k];
    }
    if (acc.x == 123.456f) sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
    return;
  }
  unsigned dst = rbeg;
  int outstanding = 0;
  if (V == 0) {
    for (int p = 0; p < npieces; ++p) {
      dma_piece(src + (size_t)p * 1024 + lane * 16, (unsigned)__builtin_amdgcn_readfirstlane((int)dst));
      dst += 1024;
      if (dst == rend) dst = rbeg;
      if ((p & 15) == 15 && ++outstanding > DEPTH) {
        if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if (DEPTH == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        --outstanding;
      }
    }
  } else {
    const int PPR = 8;  // pieces per "row"
    for (int p = 0; p < npieces; p += PPR) {
      if (V == 1) dma_row<true>(src + (size_t)p * 1024, (unsigned)lane * 16u, dst, rbeg, rend, PPR);
      else dma_row<false>(src + (size_t)p * 1024, (unsigned)lane * 16u, dst, rbeg, rend, PPR);
      if ((p & 15) == 8 && ++outstanding > DEPTH) {
        if (DEPTH == 3) asm volatile("s_waitcnt vmcnt(48)" ::: "memory");
        if (DEPTH == 2) asm volatile("s_waitcnt vmcnt(32)" ::: "memory");
        if (DEPTH == 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        if (DEPTH == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        --outstanding;
      }
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (smem[lane * 16] == 77 && sink) sink[0] = 1.f;
}

template <int V, int DEPTH, int WAVES>
static void run(const char* label, const char* slab, size_t bytes_per_cu, float* sink, hipStream_t st) {
  const size_t lds = RING * 1024;
  CK(hipFuncSetAttribute((const void*)k_stream<V, DEPTH, WAVES>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k_stream<V, DEPTH, WAVES>), dim3(256), dim3(256), lds, st, slab, bytes_per_cu, sink);
  CK(hipStreamSynchronize(st));
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, st));
    hipLaunchKernelGGL((k_stream<V, DEPTH, WAVES>), dim3(256), dim3(256), lds, st, slab, bytes_per_cu, sink);
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  const double gb = 256.0 * (double)bytes_per_cu / 1e9;
  printf("%-46s %8.1f us  %6.2f TB/s  %5.1f GB/s per CU\n", label, best * 1e3, gb / best, gb / best / 256.0 * 1e3);
}

int main() {
  hipStream_t st;
  CK(hipStreamCreate(&st));
  const size_t bytes_per_cu = 4u << 20;  // 4 MiB per CU = 1 GiB in total: nothing is served from a cache
  char* slab;
  CK(hipMalloc(&slab, 256 * bytes_per_cu + 65536));
  CK(hipMemset(slab, 1, 256 * bytes_per_cu + 65536));
  float* sink;
  CK(hipMalloc(&sink, 4096));
  run<3, 0, 1>("plain nt loads to VGPRs, 1 wave, 16 KiB batches", slab, bytes_per_cu, sink, st);
  run<3, 0, 4>("plain nt loads to VGPRs, 4 waves", slab, bytes_per_cu, sink, st);
  run<0, 3, 1>("V0 per-piece asm, depth 3, 1 wave", slab, bytes_per_cu, sink, st);
  run<0, 1, 1>("V0 per-piece asm, depth 1, 1 wave", slab, bytes_per_cu, sink, st);
  run<1, 3, 1>("V1 row asm (saddr) nt, depth 3, 1 wave", slab, bytes_per_cu, sink, st);
  run<1, 2, 1>("V1 row asm nt, depth 2, 1 wave", slab, bytes_per_cu, sink, st);
  run<1, 1, 1>("V1 row asm nt, depth 1, 1 wave", slab, bytes_per_cu, sink, st);
  run<1, 0, 1>("V1 row asm nt, depth 0, 1 wave", slab, bytes_per_cu, sink, st);
  run<2, 3, 1>("V2 row asm default policy, depth 3, 1 wave", slab, bytes_per_cu, sink, st);
  run<1, 3, 2>("V1 row asm nt, depth 3, 2 loader waves", slab, bytes_per_cu, sink, st);
  run<1, 3, 4>("V1 row asm nt, depth 3, 4 loader waves", slab, bytes_per_cu, sink, st);
  run<1, 1, 4>("V1 row asm nt, depth 1, 4 loader waves", slab, bytes_per_cu, sink, st);
  return 0;
}
