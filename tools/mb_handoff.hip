// mb_handoff.hip — what does a hand-off between two dependent GEMV-shaped stages cost?
//   (A) the shipped structure: one kernel per stage, replayed as a hipGraph chain; a stage reads the
//       N-float vector its predecessor wrote (plain loads after the kernel boundary), stages it into
//       LDS, streams its weight slab, and every wave writes a few vector entries;
//   (B) ONE persistent launch: every workgroup runs all stages; the vector is handed over as
//       (value, step) pairs written write-through (agent-scope relaxed 64-bit atomic stores ->
//       global_store ... sc1); a consumer first issues the loads of its first weight tile, then
//       re-reads its slice of the vector (sc1 loads) until every tag carries the current step.
//       No counter, no fence, no barrier across workgroups.  Spins are bounded (error flag).
// Same slabs, same tile shape (kh_gemv.h: a wave owns 2*U KiB), same vector length.
//   hipcc --offload-arch=gfx950 -O3 tools/mb_handoff.hip -o kuiperllama_amd/lib/mb_handoff
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

#define U 4          // KiB per row per wave tile
#define NMAX 8192    // vector length limit (LDS)
struct Stage {
  const f32x4* slab;
  unsigned long long n4;
};

__device__ __forceinline__ void load_tile(f32x4 (&v)[2 * U], const f32x4* p, size_t n4, size_t t, int lane) {
#pragma unroll
  for (int u = 0; u < 2 * U; ++u) {
    size_t idx = t * (size_t)(2 * U * 64) + (size_t)u * 64 + lane;
    if (idx >= n4) idx = 0;
    v[u] = __builtin_nontemporal_load(p + idx);
  }
}
__device__ __forceinline__ float dot_tile(const f32x4 (&v)[2 * U], const float* xs, int lane, int N) {
  float a = 0.f;
#pragma unroll
  for (int u = 0; u < 2 * U; ++u) {
    const f32x4 x = *(const f32x4*)(xs + ((u * 64 + lane) * 4) % N);
    a = __builtin_fmaf(v[u].x, x.x, a);
    a = __builtin_fmaf(v[u].y, x.y, a);
    a = __builtin_fmaf(v[u].z, x.z, a);
    a = __builtin_fmaf(v[u].w, x.w, a);
  }
  return a;
}

// (A) one stage per kernel
__global__ __launch_bounds__(256) void k_stage(Stage st, const float* __restrict__ xin, float* __restrict__ yout, int N) {
  __shared__ __attribute__((aligned(16))) float xs[NMAX];
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
  const size_t ntiles = (st.n4 + 2 * U * 64 - 1) / (2 * U * 64);
  f32x4 v[2 * U];
  size_t t = wave;
  if (t < ntiles) load_tile(v, st.slab, st.n4, t, lane);  // weights first in flight
  for (int i = threadIdx.x; i < N; i += 256) xs[i] = xin[i];
  __syncthreads();
  float acc = 0.f;
  while (t < ntiles) {
    acc += dot_tile(v, xs, lane, N);
    t += nw;
    if (t < ntiles) load_tile(v, st.slab, st.n4, t, lane);
  }
  for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
  if (lane == 0)
    for (size_t j = wave; j < (size_t)N; j += nw) yout[j] = acc * 1e-9f + 1.0f;
}

// (B) all stages in one persistent launch, tagged hand-off
template <int K /* vector entries per thread = N / 256 */, int SLEEP /* s_sleep units between poll rounds */>
__global__ __launch_bounds__(256) void k_persist(const Stage* __restrict__ stages, int nstages, unsigned long long* xbuf,
                                                 int N, unsigned epoch0, int* err) {
  __shared__ __attribute__((aligned(16))) float xs[NMAX];
  const int lane = threadIdx.x & 63;
  const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6), nw = (size_t)gridDim.x * 4;
  for (int s = 0; s < nstages; ++s) {
    const Stage st = stages[s];
    const unsigned long long* in = xbuf + (size_t)(s & 1) * N;
    unsigned long long* out = xbuf + (size_t)((s + 1) & 1) * N;
    const unsigned tag = epoch0 + (unsigned)s;
    const size_t ntiles = (st.n4 + 2 * U * 64 - 1) / (2 * U * 64);
    f32x4 v[2 * U];
    size_t t = wave;
    if (t < ntiles) load_tile(v, st.slab, st.n4, t, lane);  // the weight stream does not wait for the vector
    {  // all K loads of a poll round are issued before any tag is looked at; stale entries are re-read
      unsigned long long p[K];
      bool ok = false;
      for (int spins = 0; !ok && spins < (1 << 16); ++spins) {
#pragma unroll
        for (int k = 0; k < K; ++k)
          p[k] = __hip_atomic_load(in + threadIdx.x + k * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = true;
#pragma unroll
        for (int k = 0; k < K; ++k) ok = ok && (unsigned)(p[k] >> 32) == tag;
        if (!ok) __builtin_amdgcn_s_sleep(SLEEP);
      }
      if (!ok) *err = 1;  // bounded: never hang the box
#pragma unroll
      for (int k = 0; k < K; ++k) xs[threadIdx.x + k * 256] = __builtin_bit_cast(float, (unsigned)(p[k] & 0xffffffffull));
    }
    __syncthreads();
    float acc = 0.f;
    while (t < ntiles) {
      acc += dot_tile(v, xs, lane, N);
      t += nw;
      if (t < ntiles) load_tile(v, st.slab, st.n4, t, lane);
    }
    for (int o = 32; o; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) {
      const float val = acc * 1e-9f + 1.0f;
      const unsigned long long pk = (unsigned long long)__builtin_bit_cast(unsigned, val) | ((unsigned long long)(tag + 1) << 32);
      for (size_t j = wave; j < (size_t)N; j += nw) __hip_atomic_store(out + j, pk, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();  // xs is rewritten by the next stage
  }
}
__global__ void k_tag(unsigned long long* xbuf, int N, unsigned tag) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < N) xbuf[i] = (unsigned long long)__builtin_bit_cast(unsigned, 1.0f) | ((unsigned long long)tag << 32);
}

int main(int argc, char** argv) {
  const int N = 2048;
  hipStream_t S; CK(hipStreamCreate(&S));
  hipEvent_t E0, E1; CK(hipEventCreate(&E0)); CK(hipEventCreate(&E1));
  const size_t MB = 1 << 20;
  // a Llama-3.2-1B-like layer: qkv 25 MB, (attention stand-in: 1 MB), wo 16.8 MB, ffn13 134 MB, w2 67 MB
  const std::vector<std::vector<size_t>> sets = {
      {25 * MB, 1 * MB, 17 * MB, 134 * MB, 67 * MB}, {4 * MB, 3 * MB, 3 * MB, 35 * MB, 17 * MB}, {4 * MB}, {16 * MB}, {64 * MB}};
  char* pool; const size_t pool_bytes = 1024 * MB; CK(hipMalloc(&pool, pool_bytes)); CK(hipMemset(pool, 0, pool_bytes));
  float *xa, *xb; CK(hipMalloc(&xa, NMAX * 4)); CK(hipMalloc(&xb, NMAX * 4));
  CK(hipMemset(xa, 0, NMAX * 4)); CK(hipMemset(xb, 0, NMAX * 4));
  unsigned long long* xbuf; CK(hipMalloc(&xbuf, 2 * NMAX * 8));
  int* err; CK(hipMalloc(&err, 4)); CK(hipMemset(err, 0, 4));
  Stage* dstages; CK(hipMalloc(&dstages, 64 * sizeof(Stage)));
  printf("# vector length %d floats; tile 2 x %d KiB per wave; us per stage\n", N, U);
  for (int grid : {256, 512}) {  // 1024 workgroups with 32 KB of LDS each are not co-resident (the bounded spin times out)
    for (const auto& set : sets) {
      const int layers = set.size() > 1 ? 3 : 15;
      std::vector<Stage> st;
      size_t off = 0;
      for (int l = 0; l < layers; ++l)
        for (size_t b : set) {
          if (off + b > pool_bytes) off = 0;
          st.push_back(Stage{(const f32x4*)(pool + off), (unsigned long long)(b / 16)});
          off += b;  // distinct slabs: no cache reuse between stages
        }
      const int ns = (int)st.size();
      CK(hipMemcpy(dstages, st.data(), ns * sizeof(Stage), hipMemcpyHostToDevice));
      // (A) graph chain
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(S, hipStreamCaptureModeThreadLocal));
      for (int s = 0; s < ns; ++s)
        hipLaunchKernelGGL(k_stage, dim3(grid), dim3(256), 0, S, st[s], (s & 1) ? xb : xa, (s & 1) ? xa : xb, N);
      CK(hipStreamEndCapture(S, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, S)); CK(hipStreamSynchronize(S));
      float best_a = 1e30f;
      for (int r = 0; r < 5; ++r) {
        CK(hipEventRecord(E0, S)); CK(hipGraphLaunch(ge, S)); CK(hipEventRecord(E1, S)); CK(hipEventSynchronize(E1));
        float ms; CK(hipEventElapsedTime(&ms, E0, E1)); if (ms < best_a) best_a = ms;
      }
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
      // (B) persistent, tagged hand-off (grid must be co-resident: <= 8 workgroups of 256 threads per CU)
      float best_b[3] = {1e30f, 1e30f, 1e30f};
      unsigned epoch = 1;
      for (int variant = 0; variant < 3; ++variant)
        for (int r = 0; r < 5; ++r) {
          hipLaunchKernelGGL(k_tag, dim3((N + 255) / 256), dim3(256), 0, S, xbuf, N, epoch);
          CK(hipEventRecord(E0, S));
          if (variant == 0) hipLaunchKernelGGL((k_persist<8, 2>), dim3(grid), dim3(256), 0, S, dstages, ns, xbuf, N, epoch, err);
          else if (variant == 1) hipLaunchKernelGGL((k_persist<8, 16>), dim3(grid), dim3(256), 0, S, dstages, ns, xbuf, N, epoch, err);
          else hipLaunchKernelGGL((k_persist<8, 64>), dim3(grid), dim3(256), 0, S, dstages, ns, xbuf, N, epoch, err);
          CK(hipEventRecord(E1, S)); CK(hipEventSynchronize(E1));
          float ms; CK(hipEventElapsedTime(&ms, E0, E1)); if (r > 0 && ms < best_b[variant]) best_b[variant] = ms;
          epoch += (unsigned)ns + 1;
        }
      int herr = 0; CK(hipMemcpy(&herr, err, 4, hipMemcpyDeviceToHost));
      size_t tot = 0; for (size_t b : set) tot += b;
      printf("grid %4d  stages %2d  %-22s  kernel-chain %7.2f  persistent-tagged (sleep 2/16/64) %7.2f %7.2f %7.2f  us/stage%s\n", grid, ns,
             set.size() > 1 ? (set[0] == 25 * MB ? "layer 25|1|17|134|67 MB" : "layer 4|3|3|35|17 MB") : (std::to_string(set[0] / MB) + " MB").c_str(),
             best_a * 1e3f / ns, best_b[0] * 1e3f / ns, best_b[1] * 1e3f / ns, best_b[2] * 1e3f / ns, herr ? "   (SPIN TIMEOUT)" : "");
      (void)tot;
      fflush(stdout);
    }
  }
  return 0;
}
