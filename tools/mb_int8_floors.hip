// mb_int8_floors.hip — pure-stream floors for the EXACT int8 shapes of Llama-2-7B's decode kernels
// ("mb_scale-style floors for the exact shapes of k_gemv_res<true,4,0,4> (w2),
// k_qkv<true,4,4,1> and k_gemv_res<true,2,4,2> (wo)").  A floor kernel does what the real kernel's
// memory side does and nothing else: one wave per (row pair, column part), 16-byte non-temporal
// loads of the int8 rows + one scale dword per 16 weights, everything of a tile requested before it
// is touched, no LDS, no activation vector, no dequant (the loaded words are xor-folded so the loads
// cannot be dropped).  32 launches over DISTINCT slabs in one hipGraph (weights never hit a cache),
// timed with HIP events, best of 5; swept over grid sizes and U (16-byte loads per row in flight per
// lane).  Output: us per launch and the fraction of 8 TB/s for the shape's algorithmic bytes.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_int8_floors.hip -o kuiperllama_amd/lib/mb_int8_floors
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int i32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// rows x rowbytes int8 matrix + rows*rowbytes/64 fp32 scales; SPLIT waves share a row pair
template <int U, int SPLIT>
__global__ __launch_bounds__(512) void k_stream(const char* __restrict__ w, const float* __restrict__ sc, int pairs,
                                                int rowbytes, float* out) {
  const int nwav = blockDim.x >> 6, ppw = nwav / SPLIT;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, part = wave & (SPLIT - 1);
  const int Mc = rowbytes >> 4;                                  // 16-byte chunks per row
  const int Q = (((Mc + SPLIT - 1) / SPLIT) + 3) & ~3;           // chunks per part (group aligned), as gemv_pairs
  const int cb = part * Q, ce = cb + Q < Mc ? cb + Q : Mc;
  float acc = 0.f;
  for (int p = blockIdx.x * ppw + wave / SPLIT; p < pairs; p += gridDim.x * ppw) {
    const i32x4* r0 = (const i32x4*)(w + (size_t)(2 * p) * rowbytes);
    const i32x4* r1 = (const i32x4*)(w + (size_t)(2 * p + 1) * rowbytes);
    const float* s0 = sc + (size_t)(2 * p) * (rowbytes >> 6);
    const float* s1 = s0 + (rowbytes >> 6);
    for (int c0 = cb; c0 < ce; c0 += 64 * U) {
      i32x4 q0[U], q1[U];
      float g0[U], g1[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int idx = c0 + u * 64 + lane, ci = idx < ce ? idx : cb;
        q0[u] = __builtin_nontemporal_load(r0 + ci);
        q1[u] = __builtin_nontemporal_load(r1 + ci);
        g0[u] = s0[ci >> 2];
        g1[u] = s1[ci >> 2];
      }
#pragma unroll
      for (int u = 0; u < U; ++u)
        acc += g0[u] * (float)(q0[u].x ^ q0[u].y ^ q0[u].z ^ q0[u].w) + g1[u] * (float)(q1[u].x ^ q1[u].y ^ q1[u].z ^ q1[u].w);
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

// pseudo-random fill (xorshift per 16-byte word): data-dependent effects (DRAM toggling, power) are part
// of what the real kernels see; a constant fill flatters the floor
__global__ void k_fill(unsigned* p, size_t n, unsigned seed) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned x = (unsigned)i * 2654435761u ^ seed;
    x ^= x << 13; x ^= x >> 17; x ^= x << 5;
    p[i] = x;
  }
}

__global__ void k_fix_scales(unsigned* p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    p[i] = (p[i] & 0x007fffffu) | 0x3B800000u;
}

struct Shape { const char* name; int rows, rowbytes; double extra_bytes; };

template <int U, int SPLIT>
static float run(hipStream_t S, const char* w, const float* sc, size_t slab, size_t sslab, int NL, int pairs, int rowbytes,
                 int grid, int wg, float* out) {
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < NL; ++l)
    hipLaunchKernelGGL((k_stream<U, SPLIT>), dim3(grid), dim3(wg), 0, S, w + slab * l, (const float*)((const char*)sc + sslab * l),
                       pairs, rowbytes, out);
  CK(hipStreamEndCapture(S, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, S)); CK(hipStreamSynchronize(S));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9f;
  for (int r = 0; r < 5; ++r) {
    CK(hipEventRecord(e0, S)); CK(hipGraphLaunch(ge, S)); CK(hipEventRecord(e1, S)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
  }
  CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1)); CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  return best * 1e3f / NL;
}

int main(int argc, char** argv) {
  const bool random_fill = argc > 1 && argv[1][0] == 'r';
  printf("fill: %s\n", random_fill ? "pseudo-random bytes, scales in [2^-8, 2^-7)" : "constant (weights 0x01, scales 0)");
  hipStream_t S; CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  // Llama-2-7B int8, group 64: dim 4096, hidden 11008, vocab 32000
  const Shape shapes[] = {
      {"qkv   [12288 x 4096]", 12288, 4096, 2.0 * 4096 * 4 + 12288 * 4.0},   // + x, norm weight, q/k/v rows out
      {"wo    [ 4096 x 4096]", 4096, 4096, 4096 * 4.0 + 2.0 * 4096 * 4},      // + att vector, residual rd/wr
      {"ffn13 [22016 x 4096]", 22016, 4096, 2.0 * 4096 * 4 + 11008 * 4.0},
      {"w2    [ 4096 x 11008]", 4096, 11008, 11008 * 4.0 + 2.0 * 4096 * 4},
      {"cls   [32000 x 4096]", 32000, 4096, 2.0 * 4096 * 4 + 32000 * 4.0},
  };
  const int NL = 32;
  float* out; CK(hipMalloc(&out, 64));
  for (const Shape& sh : shapes) {
    const size_t slab = (size_t)sh.rows * sh.rowbytes, sslab = slab / 64 * 4;
    char* w; float* sc;
    CK(hipMalloc(&w, slab * NL)); CK(hipMalloc(&sc, sslab * NL));
    CK(hipMemset(w, 1, slab * NL)); CK(hipMemset(sc, 0, sslab * NL));
    if (random_fill) {
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)w, slab * NL / 4, 12345u);
      hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)sc, sslab * NL / 4, 777u);
      // scales: keep them finite floats: exponent 0x3B (2^-8), random mantissa
      hipLaunchKernelGGL(k_fix_scales, dim3(4096), dim3(256), 0, 0, (unsigned*)sc, sslab * NL / 4);
    }
    CK(hipDeviceSynchronize());
    const int pairs = sh.rows / 2;
    const double bytes = (double)slab + (double)sslab + sh.extra_bytes;
    float best = 1e9f; char bestcfg[96] = "";
    for (int wg : {256, 512})
      for (int grid : {256, 512, 768, 1024, 2048})
        for (int cfg = 0; cfg < 6; ++cfg) {
          float us;
          const char* nm;
          switch (cfg) {
            case 0: us = run<4, 1>(S, w, sc, slab, sslab, NL, pairs, sh.rowbytes, grid, wg, out); nm = "U4 split1"; break;
            case 1: us = run<2, 1>(S, w, sc, slab, sslab, NL, pairs, sh.rowbytes, grid, wg, out); nm = "U2 split1"; break;
            case 2: us = run<4, 2>(S, w, sc, slab, sslab, NL, pairs, sh.rowbytes, grid, wg, out); nm = "U4 split2"; break;
            case 3: us = run<2, 2>(S, w, sc, slab, sslab, NL, pairs, sh.rowbytes, grid, wg, out); nm = "U2 split2"; break;
            case 4: us = run<4, 4>(S, w, sc, slab, sslab, NL, pairs, sh.rowbytes, grid, wg, out); nm = "U4 split4"; break;
            default: us = run<8, 1>(S, w, sc, slab, sslab, NL, pairs, sh.rowbytes, grid, wg, out); nm = "U8 split1"; break;
          }
          if (us < best) { best = us; snprintf(bestcfg, sizeof bestcfg, "wg%d grid%d %s", wg, grid, nm); }
        }
    printf("%s  %7.2f MB algorithmic  floor %6.2f us  (%s)  = %.3f of 8 TB/s\n", sh.name, bytes / 1e6, best, bestcfg,
           bytes / (best * 1e-6) / 8e12);
    fflush(stdout);
    CK(hipFree(w)); CK(hipFree(sc));
  }
  return 0;
}
