// mb_overlap.hip — does STARTING THE NEXT KERNEL EARLY recover the kernel boundaries of the decode chain?
//
// The decode step is a chain of short GEMV launches (5 per layer); every boundary costs the tail of one kernel, the
// dispatch of the next and the first HBM round trip of its weight stream, i.e. ~2 us of a 5-20 us kernel
// (DESIGN.md §5).  Two persistent-kernel structures lost to the chain (profiles/r4_engine_layer_ab.txt).  This
// prototype keeps the product's GEMV loop (kh_gemv.h::gemv_pairs, same register tiles and shapes) and changes only
// HOW consecutive launches are ordered:
//
//   serial      one stream, kernel N+1 starts when N has ended (the product: 5L+2 graph nodes in a line)
//   overlapped  kernels go round-robin onto S streams (S = 2, 3, 4), so N+1 ... N+S-1 are dispatched while N runs.
//               A kernel requests its first weight tile (weights do not depend on activations), then waits for its
//               producer's PER-WORKGROUP completion flags, then reads its input vector with agent-scope loads.
//               Producers store their outputs with agent-scope stores (global_store sc1, write-through), drain
//               vmcnt, barrier, one lane stores flag[workgroup] = epoch (guide G16 R1/R2: no fence).  The epoch is
//               the workgroup's own previous flag + 1, so graph replay needs no memset and no host-side counter.
//               In-stream order (N+S after N) bounds the kernels in flight to S; no kernel waits for a LATER one
//               and S consecutive grids fit the chip together (checked on the host), so it cannot deadlock; polls
//               are bounded anyway (give-up code in err[]).
//
// Workload: the five GEMV shapes of one Llama-3.2-1B fp32 layer (qkv 3072x2048, an attention stand-in of the
// latency-bound kind 2048x256, wo 2048x2048, ffn13 2x8192x2048 with a SwiGLU pair epilogue, w2 2048x8192) with the
// product's launch shapes, 16 layers with their own weights (3.9 GB per pass), RMS-normalised inputs so the
// recurrence stays bounded; the output of layer 15 feeds layer 0 of the next replay, so a stale read anywhere
// changes the final bits: every variant must end with the SAME 2048 floats as the serial chain.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/mb_overlap.hip -o kuiperllama_amd/lib/mb_overlap
//   mb_overlap [reps]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "../kuiperllama_amd/csrc/kh_gemv.h"

namespace khm {
const char* dbg(const char*) { return nullptr; }
}  // namespace khm

#define CK(x)                                                                       \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) {                                                         \
      printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
      exit(1);                                                                      \
    }                                                                               \
  } while (0)

#define OV_POLL_LIMIT (1 << 17)  // ~0.1-0.2 s of polling, then give up loudly
#define OV_MAX_FLAGS 1024

// the launch that FOLLOWS this one (serial chain only): its first weight tile is touched line by line early in this
// kernel so that it sits in L2 / the memory-side cache when that launch asks for it.  W == nullptr: nothing.
struct PfDesc {
  const float* W;
  int K, M, U, SPLIT, wg, grid, swiglu;
};
struct OvArgs {
  const float* W;       // [K, M] row-major
  const float* xin;     // [M]
  const float* wnorm;   // [M] (ones)
  float* yout;          // [K] or [K/2] (SwiGLU)
  int K, M;
  unsigned* my_flags;          // [grid]
  const unsigned* prod_flags;  // [n_prod]
  int n_prod;
  int delta;  // producer's epoch relative to mine: 0 same pass, -1 the producer is the last kernel of the previous pass
  unsigned* err;
  unsigned long long* trace;  // [4]: start / end of the first and of the last workgroup (100 MHz wall clock)
  PfDesc pf;
};

// One 4-byte load per 128-byte line of the first tile (2 rows x U KiB) of every wave of the next launch; the waves of
// this launch share them out (workgroup b takes the next launch's workgroups b, b + grid, ...: same XCD under the
// round-robin placement, so the lines land in the L2 that will be asked for them).  Plain loads (they must allocate),
// tracked by the compiler's vmcnt like any other; the xor of the values is "used" once at the very end.
__device__ __forceinline__ int prefetch_next(const PfDesc& d, int lane) {
  int acc = 0;
  if (!d.W) return acc;
  const int nw_c = d.wg >> 6, ppw = nw_c / d.SPLIT;
  const int Mc = d.M >> 2;
  const int Q = (((Mc + d.SPLIT - 1) / d.SPLIT) + 3) & ~3;
  const int half = d.K >> 1;
  const int wave = threadIdx.x >> 6, nw_p = kh_nwaves();
  const int lines = d.U * 8;  // 128-byte lines per row of the tile
  for (int vbc = blockIdx.x; vbc < d.grid; vbc += gridDim.x)
    for (int wc = wave; wc < nw_c; wc += nw_p) {
      const int part = wc & (d.SPLIT - 1);
      int p0 = vbc * ppw + wc / d.SPLIT;
      if (p0 >= half) p0 = 0;
      const int r0 = d.swiglu ? p0 : 2 * p0, r1 = d.swiglu ? p0 + half : 2 * p0 + 1;
      const int col0 = part * Q * 4;  // floats
      for (int l = lane; l < 2 * lines; l += KH_WAVE) {
        const int row = l < lines ? r0 : r1;
        int col = col0 + (l < lines ? l : l - lines) * 32;
        if (col >= d.M) col = col0;
        acc ^= *(const int*)(d.W + (size_t)row * d.M + col);
      }
    }
  return acc;
}

__device__ __forceinline__ unsigned ld_agent_u32(const unsigned* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void st_agent_f32(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// Serial form: the product's staging (x loads first, RMS norm, LDS), plain stores.
// Overlapped form: own epoch -> first weight tile -> poll the producer's flags -> x with agent loads -> norm -> LDS;
// agent stores; drain; barrier; flag.
template <int U, int SPLIT, bool SWIGLU, bool HAND>
__global__ __launch_bounds__(KH_WG_MAX, 4) void k_stage(const OvArgs a) {
  extern __shared__ __attribute__((aligned(16))) char smem_raw[];
  const int tslot = blockIdx.x == 0 ? 0 : (blockIdx.x == gridDim.x - 1 ? 2 : -1);
  unsigned long long* const trace = a.trace;
  if (tslot >= 0 && threadIdx.x == 0) trace[tslot] = wall_clock64();
  f32x4* xs = (f32x4*)smem_raw;
  const int M = a.M, K = a.K;
  float* red = (float*)(xs + (M >> 2));
  const float* const W = a.W;
  float* const yout = a.yout;
  const int lane = threadIdx.x & 63;
  const Gemv<false, U> g(M, 0);
  const int half = K >> 1;
  auto pair = [&](int p) __attribute__((always_inline)) {
    return SWIGLU ? g.rows(W, p, W, p + half, nullptr, nullptr, M) : g.rows(W, 2 * p, W, 2 * p + 1, nullptr, nullptr, M);
  };
  auto pre = [&](int) __attribute__((always_inline)) { return NoAux{}; };
  auto epi = [&](int p, float s0, float s1, const NoAux&) __attribute__((always_inline)) {
    if (lane != 0) return;
    if (SWIGLU) {
      const float v = swiglu1(s0, s1);
      if (HAND) st_agent_f32(&yout[p], v); else yout[p] = v;
    } else {
      if (HAND) { st_agent_f32(&yout[2 * p], s0); st_agent_f32(&yout[2 * p + 1], s1); }
      else { yout[2 * p] = s0; yout[2 * p + 1] = s1; }
    }
  };
  if constexpr (!HAND) {
    Stager<true, false, 4> st(a.xin, a.wnorm, M);
    const PfDesc pfd = a.pf;
    int pf_acc = 0;
    gemv_pairs<SPLIT, false>(g, xs, half, lane, red + KH_WAVES_MAX, pair, pre,
                             [&]() __attribute__((always_inline)) {
                               st.issue();
                               pf_acc = prefetch_next(pfd, lane);  // behind the x loads, ahead of the weight tile
                             },
                             [&]() __attribute__((always_inline)) { st.finish(xs, 1e-5f, red); }, epi);
    if (pf_acc == 0x7fc0ffee && a.err) atomicOr(a.err, 2u);  // keeps the loads; never true in practice
    if (tslot >= 0 && threadIdx.x == 0) trace[tslot + 1] = wall_clock64();
  } else {
    unsigned* const my_flag = a.my_flags + blockIdx.x;
    const unsigned* const pf = a.prod_flags;
    const int n_prod = a.n_prod, delta = a.delta;
    unsigned* const err = a.err;
    const f32x4* const x4 = (const f32x4*)a.xin;
    const f32x4* const w4 = (const f32x4*)a.wnorm;
    unsigned e_prev = 0;
    auto issue = [&]() __attribute__((always_inline)) { e_prev = ld_agent_u32(my_flag); };
    auto finish = [&]() __attribute__((always_inline)) {
      const unsigned e = e_prev + 1;
      const unsigned want = e + (unsigned)delta;
      if (threadIdx.x < KH_WAVE) {  // wave 0 polls; the others sleep at the barrier
        int spins = 0;
        for (;;) {
          bool ok = true;
          for (int i = lane; i < n_prod; i += KH_WAVE) ok &= ld_agent_u32(pf + i) == want;
          if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
          if (ld_agent_u32(err) != 0) break;  // somebody gave up: everybody leaves
          if (++spins > OV_POLL_LIMIT) {
            if (lane == 0) atomicOr(err, 1u);
            break;
          }
          __builtin_amdgcn_s_sleep(4);
        }
      }
      __syncthreads();
      const int M4 = M >> 2;
      f32x4 xv[4], wv[4];
      {
        const f32x4* p[4];
#pragma unroll
        for (int v = 0; v < 4; ++v) {
          const int i = threadIdx.x + v * kh_wg();
          p[v] = x4 + (i < M4 ? i : 0);
          wv[v] = w4[i < M4 ? i : 0];
        }
        asm volatile(
            "global_load_dwordx4 %0, %4, off sc1\n\t"
            "global_load_dwordx4 %1, %5, off sc1\n\t"
            "global_load_dwordx4 %2, %6, off sc1\n\t"
            "global_load_dwordx4 %3, %7, off sc1\n\t"
            "s_waitcnt vmcnt(0)"
            : "=&v"(xv[0]), "=&v"(xv[1]), "=&v"(xv[2]), "=&v"(xv[3])
            : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3])
            : "memory");
      }
      float ss = 0.f;
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const float t = fma4(xv[v], xv[v], 0.f);
        ss += (threadIdx.x + v * kh_wg() < M4) ? t : 0.f;
      }
      ss = block_sum(ss, red);
      const float rs = 1.0f / sqrtf(ss / (float)M + 1e-5f);
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        const int i = threadIdx.x + v * kh_wg();
        if (i < M4) {
          f32x4 t = xv[v];
          t.x = wv[v].x * (rs * t.x);
          t.y = wv[v].y * (rs * t.y);
          t.z = wv[v].z * (rs * t.z);
          t.w = wv[v].w * (rs * t.w);
          xs[i] = t;
        }
      }
      __syncthreads();
    };
    gemv_pairs<SPLIT, false>(g, xs, half, lane, red + KH_WAVES_MAX, pair, pre, issue, finish, epi);
    // completion: every wave's stores acknowledged (write-through), then the workgroup's flag
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_store(my_flag, e_prev + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tslot >= 0 && threadIdx.x == 0) trace[tslot + 1] = wall_clock64();
  }
}

struct StageDef {
  const char* name;
  int K, M, U, SPLIT, wg, grid;
  bool swiglu;
  int in_len;  // floats of the producer's output this stage reads (== M)
};
// launch shapes of the product for Llama-3.2-1B fp32 (profiles/r4_kernel_timeline.txt)
static const StageDef STAGES[5] = {
    {"qkv", 3072, 2048, 4, 2, 256, 768, false, 2048},
    {"attn*", 2048, 256, 1, 1, 256, 256, false, 256},
    {"wo", 2048, 2048, 4, 2, 256, 512, false, 2048},
    {"ffn13", 16384, 2048, 8, 1, 256, 512, true, 2048},
    {"w2", 2048, 8192, 8, 4, 512, 512, false, 8192},
};

// grids: the product's; one that lets any TWO consecutive kernels be resident together (VGPRs per SIMD: 80-register
// kernels at n waves + 112-register kernels at m waves <= 512); one that lets any THREE
static const int GRIDS[3][5] = {{768, 256, 512, 512, 512}, {768, 256, 512, 512, 256}, {512, 256, 512, 256, 256}};

template <bool HAND>
static void launch_stage(int s, const OvArgs& a, hipStream_t st, int grid) {
  StageDef d = STAGES[s];
  d.grid = grid;
  const size_t lds = (size_t)d.M * 4 + 3 * KH_WAVES_MAX * sizeof(float);
  switch (s) {
    case 0: hipLaunchKernelGGL((k_stage<4, 2, false, HAND>), dim3(d.grid), dim3(d.wg), lds, st, a); break;
    case 1: hipLaunchKernelGGL((k_stage<1, 1, false, HAND>), dim3(d.grid), dim3(d.wg), lds, st, a); break;
    case 2: hipLaunchKernelGGL((k_stage<4, 2, false, HAND>), dim3(d.grid), dim3(d.wg), lds, st, a); break;
    case 3: hipLaunchKernelGGL((k_stage<8, 1, true, HAND>), dim3(d.grid), dim3(d.wg), lds, st, a); break;
    default: hipLaunchKernelGGL((k_stage<8, 4, false, HAND>), dim3(d.grid), dim3(d.wg), lds, st, a); break;
  }
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 50;
  const int L = argc > 2 ? atoi(argv[2]) : 16;
  const int NK = 5 * L;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  printf("device %s, %d CUs; %d layers x 5 kernels, %d replays per timing\n", prop.gcnArchName, prop.multiProcessorCount, L, reps);

  // weights: per layer, per stage
  std::vector<float*> W(NK);
  size_t wbytes = 0;
  {
    std::vector<float> h;
    uint32_t seed = 12345;
    for (int k = 0; k < NK; ++k) {
      const StageDef& d = STAGES[k % 5];
      const size_t n = (size_t)d.K * d.M;
      h.resize(n);
      const float amp = sqrtf(3.0f / (float)d.M);
      for (size_t i = 0; i < n; ++i) {
        seed = seed * 1664525u + 1013904223u;
        h[i] = amp * ((float)(seed >> 8) * (1.0f / 8388608.0f) - 1.0f);
      }
      CK(hipMalloc(&W[k], n * 4));
      CK(hipMemcpy(W[k], h.data(), n * 4, hipMemcpyHostToDevice));
      wbytes += n * 4;
    }
  }
  printf("weights %.1f MB per pass (%.1f MB per layer)\n", wbytes / 1e6, wbytes / 1e6 / L);
  float* ones;
  CK(hipMalloc(&ones, 8192 * 4));
  {
    std::vector<float> h(8192, 1.0f);
    CK(hipMemcpy(ones, h.data(), 8192 * 4, hipMemcpyHostToDevice));
  }
  // activation buffers: one per kernel slot (outputs), the last one feeds slot 0
  std::vector<float*> Y(NK);
  for (int k = 0; k < NK; ++k) CK(hipMalloc(&Y[k], 16384 * 4));
  unsigned* flags;
  CK(hipMalloc(&flags, (size_t)NK * OV_MAX_FLAGS * 4));
  unsigned* err;
  CK(hipMalloc(&err, 4));
  unsigned long long* trace;
  CK(hipMalloc(&trace, (size_t)NK * 4 * 8));

  auto reset = [&]() {
    std::vector<float> h(16384);
    for (int i = 0; i < 16384; ++i) h[i] = sinf(0.37f * (float)i) + 0.25f;
    for (int k = 0; k < NK; ++k) CK(hipMemcpy(Y[k], h.data(), 16384 * 4, hipMemcpyHostToDevice));
    CK(hipMemset(flags, 0, (size_t)NK * OV_MAX_FLAGS * 4));
    CK(hipMemset(err, 0, 4));
    CK(hipDeviceSynchronize());
  };
  int gset = 0;
  bool prefetch = false;
  auto args_of = [&](int k) {
    const StageDef& d = STAGES[k % 5];
    const int kp = (k + NK - 1) % NK;
    OvArgs a;
    a.W = W[k];
    a.xin = Y[kp];
    a.wnorm = ones;
    a.yout = Y[k];
    a.K = d.K;
    a.M = d.M;
    a.my_flags = flags + (size_t)k * OV_MAX_FLAGS;
    a.prod_flags = flags + (size_t)kp * OV_MAX_FLAGS;
    a.n_prod = GRIDS[gset][kp % 5];
    a.delta = k == 0 ? -1 : 0;
    a.err = err;
    a.trace = trace + (size_t)k * 4;
    a.pf = PfDesc{nullptr, 0, 0, 0, 0, 0, 0, 0};
    if (prefetch) {
      const int kn = (k + 1) % NK;
      const StageDef& n = STAGES[kn % 5];
      a.pf = PfDesc{W[kn], n.K, n.M, n.U, n.SPLIT, n.wg, GRIDS[gset][kn % 5], n.swiglu ? 1 : 0};
    }
    return a;
  };

  // residency check: S consecutive grids must fit the chip together (threads per CU <= 2048, i.e. <= 64 VGPRs)
  {
    hipFuncAttributes fa;
    const void* fns[5] = {(const void*)k_stage<4, 2, false, true>, (const void*)k_stage<1, 1, false, true>,
                          (const void*)k_stage<4, 2, false, true>, (const void*)k_stage<8, 1, true, true>,
                          (const void*)k_stage<8, 4, false, true>};
    for (int s = 0; s < 5; ++s) {
      CK(hipFuncGetAttributes(&fa, fns[s]));
      printf("  %-6s grid %4d x %3d  VGPRs %3d  scratch %zu B  LDS %zu B\n", STAGES[s].name, STAGES[s].grid, STAGES[s].wg,
             fa.numRegs, (size_t)fa.localSizeBytes, (size_t)STAGES[s].M * 4 + 96);
    }
  }

  hipStream_t st[4];
  for (int i = 0; i < 4; ++i) CK(hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking));
  hipEvent_t e0, e1, evf, evj[4];
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  CK(hipEventCreateWithFlags(&evf, hipEventDisableTiming));
  for (int i = 0; i < 4; ++i) CK(hipEventCreateWithFlags(&evj[i], hipEventDisableTiming));

  std::vector<float> ref(2048), got(2048);
  auto run_variant = [&](const char* name, bool hand, int S, int gs, bool graph = true, bool pf = false) {
    gset = gs;
    prefetch = pf;
    reset();
    hipGraph_t gr = nullptr;
    hipGraphExec_t ge = nullptr;
    auto enqueue_pass = [&]() {
      for (int k = 0; k < NK; ++k) {
        const OvArgs a = args_of(k);
        if (hand) launch_stage<true>(k % 5, a, st[k % S], GRIDS[gs][k % 5]);
        else launch_stage<false>(k % 5, a, st[k % S], GRIDS[gs][k % 5]);
      }
    };
    auto fork = [&]() {
      if (S > 1) {
        CK(hipEventRecord(evf, st[0]));
        for (int i = 1; i < S; ++i) CK(hipStreamWaitEvent(st[i], evf, 0));
      }
    };
    auto join = [&]() {
      for (int i = 1; i < S; ++i) {
        CK(hipEventRecord(evj[i], st[i]));
        CK(hipStreamWaitEvent(st[0], evj[i], 0));
      }
    };
    float ms = 0.f;
    if (graph) {
      CK(hipStreamBeginCapture(st[0], hipStreamCaptureModeGlobal));
      fork();
      enqueue_pass();
      join();
      CK(hipStreamEndCapture(st[0], &gr));
      CK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
      // 3 warm passes, then reps timed; the recurrence runs 3 + reps passes in every variant
      for (int r = 0; r < 3; ++r) CK(hipGraphLaunch(ge, st[0]));
      CK(hipStreamSynchronize(st[0]));
      CK(hipEventRecord(e0, st[0]));
      for (int r = 0; r < reps; ++r) CK(hipGraphLaunch(ge, st[0]));
      CK(hipEventRecord(e1, st[0]));
      CK(hipEventSynchronize(e1));
    } else {
      // direct launches: the host runs ahead of the GPU (3-4 us per launch against ~10 us per kernel); the streams
      // are forked once and joined once, passes follow each other without a join (the flags carry the order)
      for (int r = 0; r < 3; ++r) enqueue_pass();
      for (int i = 0; i < S; ++i) CK(hipStreamSynchronize(st[i]));
      CK(hipEventRecord(e0, st[0]));
      fork();
      for (int r = 0; r < reps; ++r) enqueue_pass();
      join();
      CK(hipEventRecord(e1, st[0]));
      CK(hipEventSynchronize(e1));
    }
    CK(hipEventElapsedTime(&ms, e0, e1));

    unsigned herr = 0;
    CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(got.data(), Y[NK - 1], 2048 * 4, hipMemcpyDeviceToHost));
    const double us_pass = ms * 1e3 / reps;
    bool same = true;
    if (!hand && S == 1 && gs == 0) ref = got;
    else same = memcmp(ref.data(), got.data(), 2048 * 4) == 0;
    bool finite = true;
    for (float v : got) finite &= std::isfinite(v);
    printf("%-28s %8.1f us per pass  %6.2f us per layer  %5.2f TB/s  %s%s%s\n", name, us_pass, us_pass / L,
           wbytes / us_pass / 1e6, same ? "bits==serial" : "BITS DIFFER", herr ? "  POLL GAVE UP" : "",
           finite ? "" : "  NON-FINITE");
    {  // timeline of one middle layer of the last pass: when did each kernel's first / last workgroup start and end
      std::vector<unsigned long long> t((size_t)NK * 4);
      CK(hipMemcpy(t.data(), trace, t.size() * 8, hipMemcpyDeviceToHost));
      const int k0 = 5 * (L / 2);
      const double base = (double)t[(size_t)k0 * 4];
      printf("    timeline (us, layer %d):", L / 2);
      for (int k = k0; k < k0 + 6 && k < NK; ++k) {
        const unsigned long long* q = &t[(size_t)k * 4];
        const double s0 = ((double)q[0] - base) / 100.0, e0_ = ((double)q[1] - base) / 100.0;
        const double s1 = ((double)q[2] - base) / 100.0, e1_ = ((double)q[3] - base) / 100.0;
        printf("  %s[%.1f/%.1f -> %.1f/%.1f]", STAGES[k % 5].name, s0, s1, e0_, e1_);
      }
      printf("\n");
    }
    fflush(stdout);
    if (ge) CK(hipGraphExecDestroy(ge));
    if (gr) CK(hipGraphDestroy(gr));
    return us_pass;
  };

  if (argc > 3 && !strcmp(argv[3], "pf")) {  // only the prefetch question
    for (int round = 0; round < 4; ++round) {
      const double a = run_variant("serial, product grids", false, 1, 0);
      const double ap = run_variant("serial + prefetch of next tile", false, 1, 0, true, true);
      printf("  -> prefetch of the next launch's first tile: %.3fx of the serial chain\n", ap / a);
    }
    return 0;
  }
  for (int round = 0; round < 2; ++round) {
    const double a = run_variant("serial, product grids", false, 1, 0);
    const double ap = run_variant("serial + prefetch of next tile", false, 1, 0, true, true);
    printf("  -> prefetch of the next launch's first tile: %.3fx of the serial chain\n", ap / a);
    run_variant("serial, pair-safe grids", false, 1, 1);
    run_variant("serial, triple-safe grids", false, 1, 2);
    const double b = run_variant("serial + flags, product grids", true, 1, 0);
    const double c2 = run_variant("2 streams, pair-safe grids", true, 2, 1);
    const double c2b = run_variant("2 streams, triple-safe grids", true, 2, 2);
    const double c3 = run_variant("3 streams, triple-safe grids", true, 3, 2);
    run_variant("direct launches, serial", false, 1, 0, false);
    run_variant("direct, serial + flags", true, 1, 0, false);
    run_variant("direct, 2 streams, pair-safe", true, 2, 1, false);
    run_variant("direct, 3 streams, triple-safe", true, 3, 2, false);
    printf("  -> of the serial chain: protocol alone %.3fx, 2 streams %.3fx / %.3fx, 3 streams %.3fx\n", b / a, c2 / a,
           c2b / a, c3 / a);
  }
  return 0;
}
