// mb_gemv_ladder.hip — where does an int8 decode GEMV lose time against a pure stream of its bytes?
// The product's own row-pair loop (kh_gemv.h::gemv_pairs, rolling tile refill) with ingredients
// switched on rung by rung, on the shapes of Llama-2-7B int8 (group 64):
//   rung 0  stream only: the loop requests and retires its tiles (weights + scales, xor-folded);
//           nothing is staged, no LDS, no arithmetic, nothing stored                 (the floor)
//   rung 1  + the staging prologue: the input vector (and norm weight) loaded FIRST, RMS block
//             reduction, normalise, LDS write, barrier - the FMAs still replaced by the fold
//   rung 2  + the four ds_read_b128 of x per slot (folded, no arithmetic)
//   rung 3  + int8 -> f32 converts and FMAs (the full inner loop)
//   rung 4  + wave reduction, epilogue (SwiGLU / residual add), store              (= the kernel)
// Shapes: ffn13 (11008 (w1,w3) pairs x 4096, norm), qkv-like (6144 pairs x 4096, norm; the RoPE
// epilogue is replaced by a plain store), wo (2048 pairs x 4096, SPLIT 2, no norm), w2 (2048 pairs
// x 11008, SPLIT 4, 512 threads, no norm).  32 launches over distinct slabs per hipGraph, best of 5.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I kuiperllama_amd/csrc tools/mb_gemv_ladder.hip -o kuiperllama_amd/lib/mb_gemv_ladder
#include <cstdio>
#include <cstdlib>

#include "kh_gemv.h"
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// stream-only views of an int8 matrix with the interface gemv_pairs expects
template <int U, int LDSREADS>
struct GemvFold : Gemv<true, U> {
  using Base = Gemv<true, U>;
  __device__ __forceinline__ GemvFold(int M, int gs) : Base(M, gs) {}
  __device__ __forceinline__ void fma1(const typename Base::Regs& r, const f32x4* xs, int c0, int lim, int lane,
                                       int u, float& a0, float& a1) const {
    const int idx = c0 + u * KH_WAVE + lane;
    if (LDSREADS) {
      const int ci = idx < lim ? idx : 0, plane = this->Mc + 1;
      const f32x4 x0 = xs[ci], x1 = xs[plane + ci], x2 = xs[2 * plane + ci], x3 = xs[3 * plane + ci];
      a0 += x0.x + x1.y + x2.z + x3.w;
    }
    a0 += r.g0[u] * (float)(r.q0[u].x ^ r.q0[u].y ^ r.q0[u].z ^ r.q0[u].w);
    a1 += r.g1[u] * (float)(r.q1[u].x ^ r.q1[u].y ^ r.q1[u].z ^ r.q1[u].w);
  }
  __device__ __forceinline__ void fma(const typename Base::Regs& r, const f32x4* xs, int c0, int lim, int lane,
                                      float& a0, float& a1) const {
#pragma unroll
    for (int u = 0; u < U; ++u) fma1(r, xs, c0, lim, lane, u, a0, a1);
  }
};

struct Args {
  const int8_t* wa; const int8_t* wb; const float* sa; const float* sb;  // two row sets (w1/w3) or the same matrix twice
  const float* x; const float* wn; float* out; int M, pairs, paired;    // paired: rows (r, r) of (wa, wb); else rows (2p, 2p+1) of wa
};

template <int RUNG, int U, int SPLIT, bool NORM, int MAXV>
__global__ __launch_bounds__(KH_WG_MAX) void k_ladder(const Args a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  f32x4* xs = (f32x4*)smem_raw;
  float* red = (float*)(xs + 4 * ((a.M >> 4) + 1));
  const int lane = threadIdx.x & 63;
  const int8_t *wa = a.wa, *wb = a.wb;
  const float *sa = a.sa, *sb = a.sb;
  float* const out = a.out;
  const int M = a.M, paired = a.paired;
  using G = typename std::conditional<(RUNG >= 3), Gemv<true, U>, GemvFold<U, (RUNG >= 2)>>::type;
  const G g(M, 6);
  Stager<NORM, true, MAXV> st(a.x, a.wn, M);
  auto pair = [&](int p) __attribute__((always_inline)) {
    return paired ? g.rows(wa, p, wb, p, sa, sb, M) : g.rows(wa, 2 * p, wa, 2 * p + 1, sa, sa, M);
  };
  float fold = 0.f;
  auto epi = [&](int p, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (RUNG >= 4) {
      if (lane == 0) {
        if (paired) out[p] = swiglu1(s0, s1);
        else { out[2 * p] += s0; out[2 * p + 1] += s1; }
      }
    } else {
      fold += s0 + s1;
    }
  };
  gemv_pairs<SPLIT, true>(g, xs, a.pairs, lane, red + KH_WAVES_MAX, pair, [](int) __attribute__((always_inline)) { return NoAux{}; },
                    [&]() __attribute__((always_inline)) { if (RUNG >= 1) st.issue(); },
                    [&]() __attribute__((always_inline)) { if (RUNG >= 1) st.finish(xs, 1e-5f, red); else __syncthreads(); }, epi);
  if (RUNG < 4 && fold == 123.456f) out[0] = fold;
}

template <int RUNG, int U, int SPLIT, bool NORM, int MAXV>
static float run(hipStream_t S, const Args* args, int NL, int grid, int wg, size_t lds) {
  if (lds > 64 * 1024)
    CK(hipFuncSetAttribute((const void*)k_ladder<RUNG, U, SPLIT, NORM, MAXV>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
  for (int l = 0; l < NL; ++l) hipLaunchKernelGGL((k_ladder<RUNG, U, SPLIT, NORM, MAXV>), dim3(grid), dim3(wg), lds, S, args[l]);
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

template <int U, int SPLIT, bool NORM, int MAXV>
static void ladder(hipStream_t S, const char* name, const Args* args, int NL, int grid, int wg, size_t lds, double bytes) {
  const float t[5] = {run<0, U, SPLIT, NORM, MAXV>(S, args, NL, grid, wg, lds), run<1, U, SPLIT, NORM, MAXV>(S, args, NL, grid, wg, lds),
                      run<2, U, SPLIT, NORM, MAXV>(S, args, NL, grid, wg, lds), run<3, U, SPLIT, NORM, MAXV>(S, args, NL, grid, wg, lds),
                      run<4, U, SPLIT, NORM, MAXV>(S, args, NL, grid, wg, lds)};
  printf("%-6s wg%d grid%-4d U%d split%d |", name, wg, grid, U, SPLIT);
  for (int r = 0; r < 5; ++r) printf(" %6.2f (%.3f)", t[r], bytes / (t[r] * 1e-6) / 8e12);
  printf("\n");
  fflush(stdout);
}

int main() {
  hipStream_t S; CK(hipStreamCreateWithFlags(&S, hipStreamNonBlocking));
  const int dim = 4096, hidden = 11008, NL = 32;
  const size_t big = (size_t)2 * hidden * dim;  // largest matrix set per layer (w1 + w3)
  char* w; char* sc; float *x, *wn, *out;
  CK(hipMalloc(&w, big * NL)); CK(hipMalloc(&sc, big / 64 * 4 * NL));
  CK(hipMalloc(&x, hidden * 4)); CK(hipMalloc(&wn, hidden * 4)); CK(hipMalloc(&out, 3 * hidden * 4));
  CK(hipMemset(w, 1, big * NL)); CK(hipMemset(sc, 0, big / 64 * 4 * NL)); CK(hipMemset(x, 0, hidden * 4)); CK(hipMemset(wn, 0, hidden * 4));
  CK(hipMemset(out, 0, 3 * hidden * 4));
  CK(hipDeviceSynchronize());
  printf("Llama-2-7B int8 decode GEMV shapes; us per launch (fraction of 8 TB/s of the shape's algorithmic bytes)\n");
  printf("%-34s     rung0: stream   rung1: +stage   rung2: +LDS rd  rung3: +dequant rung4: +epilogue\n", "shape / launch");
  Args args[NL];
  auto fill = [&](size_t rows, int M, int paired) {
    const size_t wb = rows * (size_t)M, sb = wb / 64 * 4;  // bytes per layer slab
    for (int l = 0; l < NL; ++l) {
      const int8_t* w0 = (const int8_t*)(w + wb * l);
      const float* s0 = (const float*)(sc + sb * l);
      args[l] = Args{w0, paired ? w0 + wb / 2 : w0, s0, paired ? (const float*)((const char*)s0 + sb / 2) : s0, x, wn, out, M,
                     (int)(rows / 2), paired};
    }
    return (double)wb + (double)sb;
  };
  const size_t lds_dim = kh_q8_lds_bytes(dim) + 3 * KH_WAVES_MAX * sizeof(float);
  const size_t lds_hid = kh_q8_lds_bytes(hidden) + 3 * KH_WAVES_MAX * sizeof(float);
#define VARIANTS(NAME, UU, SP, NORM, MV, GRID, WG, LDS, B) ladder<UU, SP, NORM, MV>(S, NAME, args, NL, GRID, WG, LDS, B);
  {
    const double b = fill(2 * (size_t)hidden, dim, 1) + 2.0 * dim * 4 + hidden * 4.0;
    for (int grid : {512, 1024}) { VARIANTS("ffn13", 4, 1, true, 4, grid, 256, lds_dim, b) }
    for (int grid : {512, 1024}) { VARIANTS("ffn13", 2, 1, true, 4, grid, 256, lds_dim, b) }
  }
  {
    const double b = fill(3 * (size_t)dim, dim, 0) + 2.0 * dim * 4 + 3.0 * dim * 4;
    for (int grid : {512, 768}) { VARIANTS("qkv", 4, 1, true, 4, grid, 256, lds_dim, b) }
    for (int grid : {768, 1024}) { VARIANTS("qkv", 2, 1, true, 4, grid, 256, lds_dim, b) }
  }
  {
    const double b = fill((size_t)dim, dim, 0) + dim * 4.0 + 2.0 * dim * 4;
    for (int grid : {256, 512}) { VARIANTS("wo", 2, 2, false, 4, grid, 256, lds_dim, b) }
    for (int grid : {256, 512}) { VARIANTS("wo", 4, 1, false, 4, grid, 256, lds_dim, b) }
    for (int grid : {256, 512}) { VARIANTS("wo", 2, 1, false, 4, grid, 256, lds_dim, b) }
  }
  {
    const double b = fill((size_t)dim, hidden, 0) + hidden * 4.0 + 2.0 * dim * 4;
    for (int grid : {256, 512}) { VARIANTS("w2", 4, 4, false, 6, grid, 512, lds_hid, b) }
    for (int grid : {256, 512}) { VARIANTS("w2-mv0", 4, 4, false, 0, grid, 512, lds_hid, b) }
    for (int grid : {256, 512}) { VARIANTS("w2", 4, 2, false, 6, grid, 512, lds_hid, b) }
  }
  return 0;
}
