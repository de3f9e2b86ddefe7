// mb_vmcnt_order.hip — do a VGPR-returning global load and LDS-DMA operations retire in issue order relative to each
// other?  (They share the one vmcnt counter of gfx9.)  Each wave: set a register to a sentinel, request it from a COLD
// address (its own 64-byte line of a 1-GiB buffer nobody touched: an HBM round trip), then request NDMA 1-KiB pieces
// by global_load_lds_dwordx4 from one HOT 1-KiB region (L2 / MALL hits), then s_waitcnt vmcnt(NDMA) - "everything
// older than the NDMA youngest operations has retired" if retirement is in order - and store the register.  A lane
// that still holds the sentinel was read before its load landed.  Control: the same with NDMA further register loads
// from the hot region instead of DMA (loads of one kind do retire in order).
//   hipcc --offload-arch=gfx950 -O3 tools/mb_vmcnt_order.hip -o kuiperllama_amd/lib/mb_vmcnt_order
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)
#define SENTINEL 0x7fc0dead

// MODE 0: as above.  MODE 1: the DMA pieces carry `nt` and come from a COLD region too (the ring's weight stream), the
// register load is a dwordx4 from a HOT address (the staging loads of the first ring version): can a hot register
// load be overtaken... it cannot be "overtaken" in an in-order queue, the question is only whether vmcnt says so.
template <int NDMA, int MODE>
__global__ __launch_bounds__(256) void k_order2(const uint32_t* __restrict__ cold, const uint32_t* __restrict__ hot,
                                                uint32_t* out, size_t cold_stride_words) {
  __shared__ __attribute__((aligned(1024))) char ring[4][32 * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const uint32_t* csrc = cold;  // uniform base; the wave's 32 KiB of cold bytes are addressed through the lane offset
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)&ring[__builtin_amdgcn_readfirstlane(wave)][0]);
  unsigned voff = (unsigned)lane * 16u;
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  u4 v[8];
  for (int i = 0; i < 8; ++i) v[i] = u4{SENTINEL, SENTINEL, SENTINEL, SENTINEL};
  // eight dwordx4 register loads from the hot KiB (every wave of the chip reads the same lines: the hot spot of the
  // vector staging), then NDMA cold nt DMA pieces, then vmcnt(NDMA)
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("global_load_dwordx4 %0, %1, %2" : "+v"(v[i]) : "v"(voff), "s"(hot) : "memory");
  unsigned dst = lds0;
  unsigned coff = (unsigned)(gw * cold_stride_words * 4) + (unsigned)lane * 16u;
  for (int i = 0; i < NDMA; ++i) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" : : "v"(coff), "s"(csrc), "s"(dst) : "memory");
    coff += 1024u;
    dst += 1024u;
  }
  asm volatile("s_waitcnt vmcnt(%c8)"
               : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7])
               : "n"(NDMA)
               : "memory");
  uint32_t bad = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) bad += (v[i].x == SENTINEL) + (v[i].y == SENTINEL) + (v[i].z == SENTINEL) + (v[i].w == SENTINEL);
  out[gw * 64 + lane] = bad ? SENTINEL : 0x22222222u;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  (void)MODE;
}

template <int NDMA, bool DMA>
__global__ __launch_bounds__(256) void k_order(const uint32_t* __restrict__ cold, const uint32_t* __restrict__ hot,
                                               uint32_t* out, size_t cold_stride_words) {
  __shared__ __attribute__((aligned(1024))) char ring[4][16 * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const size_t gw = (size_t)blockIdx.x * 4 + wave;
  const uint32_t* src = cold + gw * cold_stride_words + lane;  // one cold 256-B stretch per wave
  const unsigned lds0 = (unsigned)__builtin_amdgcn_readfirstlane(
      (int)(unsigned)(size_t)(__attribute__((address_space(3))) char*)&ring[__builtin_amdgcn_readfirstlane(wave)][0]);
  const unsigned voff = (unsigned)lane * 16u;
  uint32_t v = SENTINEL, junk[16];
  if (DMA) {
    asm volatile(
        "global_load_dword %[v], %[src], off\n\t"
        ".rept %c[n]\n\t"
        "s_mov_b32 m0, %[lds]\n\t"
        "s_nop 0\n\t"
        "global_load_lds_dwordx4 %[voff], %[hot]\n\t"
        ".endr\n\t"
        "s_waitcnt vmcnt(%c[n])\n\t"
        : [v] "+v"(v)
        : [src] "v"(src), [voff] "v"(voff), [hot] "s"(hot), [lds] "s"(lds0), [n] "n"(NDMA)
        : "memory");
  } else {
    asm volatile("global_load_dword %[v], %[src], off" : [v] "+v"(v) : [src] "v"(src) : "memory");
#pragma unroll
    for (int i = 0; i < NDMA && i < 16; ++i)
      asm volatile("global_load_dword %0, %1, %2" : "=v"(junk[i]) : "v"(voff), "s"(hot) : "memory");
    asm volatile("s_waitcnt vmcnt(%c[n])" : "+v"(v) : [n] "n"(NDMA < 16 ? NDMA : 16) : "memory");
  }
  out[gw * 64 + lane] = v;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!DMA) {
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < NDMA && i < 16; ++i) s += junk[i];
    if (s == 0x12345678u) out[0] = s;
  }
}

template <int NDMA, bool DMA>
static void run(const char* label, const uint32_t* cold, const uint32_t* hot, uint32_t* out, size_t nwaves, size_t stride,
                int rep) {
  const int grid = (int)(nwaves / 4);
  CK(hipMemset(out, 0, nwaves * 64 * 4));
  // a different cold region per repetition: offset the base so that no line was touched before
  hipLaunchKernelGGL((k_order<NDMA, DMA>), dim3(grid), dim3(256), 0, 0, cold + (size_t)rep * 64, hot, out, stride);
  CK(hipDeviceSynchronize());
  uint32_t* h = (uint32_t*)malloc(nwaves * 64 * 4);
  CK(hipMemcpy(h, out, nwaves * 64 * 4, hipMemcpyDeviceToHost));
  size_t stale = 0, waves_stale = 0;
  for (size_t w = 0; w < nwaves; ++w) {
    size_t s = 0;
    for (int l = 0; l < 64; ++l) s += h[w * 64 + l] == SENTINEL;
    stale += s;
    waves_stale += s != 0;
  }
  printf("%-62s %8zu of %zu lanes still held the sentinel (%zu of %zu waves)\n", label, stale, nwaves * 64, waves_stale, nwaves);
  free(h);
}

template <int NDMA>
static void run2(const char* label, const uint32_t* cold, const uint32_t* hot, uint32_t* out, size_t nwaves, size_t stride,
                 int rep) {
  CK(hipMemset(out, 0, nwaves * 64 * 4));
  hipLaunchKernelGGL((k_order2<NDMA, 1>), dim3((int)(nwaves / 4)), dim3(256), 0, 0, cold + (size_t)rep * 8192, hot, out, stride);
  CK(hipDeviceSynchronize());
  uint32_t* h = (uint32_t*)malloc(nwaves * 64 * 4);
  CK(hipMemcpy(h, out, nwaves * 64 * 4, hipMemcpyDeviceToHost));
  size_t stale = 0;
  for (size_t i = 0; i < nwaves * 64; ++i) stale += h[i] == SENTINEL;
  printf("%-62s %8zu of %zu lanes read a register before its load had landed\n", label, stale, nwaves * 64);
  free(h);
}

int main() {
  const size_t nwaves = 4096, stride = 64 * 1024;  // words: each wave's cold line sits 256 KiB from the next
  uint32_t *cold, *hot, *out;
  CK(hipMalloc(&cold, nwaves * stride * 4 + 65536));
  CK(hipMemset(cold, 0x11, nwaves * stride * 4 + 65536));  // 0x11111111 != sentinel
  CK(hipMalloc(&hot, 4096));
  CK(hipMemset(hot, 0x22, 4096));
  CK(hipMalloc(&out, nwaves * 64 * 4));
  // evict the cold buffer's lines touched by the memset: stream another GiB through the caches
  uint32_t* flush;
  CK(hipMalloc(&flush, (size_t)1 << 30));
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(flush, rep, (size_t)1 << 30));
    CK(hipDeviceSynchronize());
    printf("-- repetition %d (cold lines never read before; the hot KiB is re-read by every wave)\n", rep);
    run<8, true>("1 register load (cold) + 8 LDS-DMA pieces (hot), vmcnt(8)", cold, hot, out, nwaves, stride, 4 * rep + 0);
    run<28, true>("1 register load (cold) + 28 LDS-DMA pieces (hot), vmcnt(28)", cold, hot, out, nwaves, stride, 4 * rep + 1);
    run<8, false>("control: 1 register load (cold) + 8 register loads (hot), vmcnt(8)", cold, hot, out, nwaves, stride, 4 * rep + 2);
    run<16, false>("control: 1 register load (cold) + 16 register loads (hot), vmcnt(16)", cold, hot, out, nwaves, stride, 4 * rep + 3);
    run2<8>("8 dwordx4 register loads (hot spot) + 8 cold nt DMA pieces, vmcnt(8)", cold, hot, out, 2048, stride, rep);
    run2<28>("8 dwordx4 register loads (hot spot) + 28 cold nt DMA pieces, vmcnt(28)", cold, hot, out, 2048, stride, rep + 3);
  }
  return 0;
}
